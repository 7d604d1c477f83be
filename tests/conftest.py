import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
GOLDEN = os.path.join(REPO, "tests", "golden")
# the tests drive single kernels (jmid_dbg_*) and kernel-variant knobs: they load the DIAGNOSTICS flavour of the library - the
# same kernels as the production libjmid_hip.so, which smoke(), bench.py and tests/test_gpu_production_lib.py run on
PROD_LIB = os.path.join(REPO, "safe-interactive-crowdnav_amd", "csrc", "libjmid_hip.so")
DIAG_LIB = os.path.join(REPO, "safe-interactive-crowdnav_amd", "csrc", "libjmid_hip_diag.so")
os.environ.setdefault("JMID_LIB", DIAG_LIB)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "expects_erange: the test provokes JMID_ERANGE on purpose")
    config.addinivalue_line("markers", "expects_timeout: the test provokes (or tolerates) JMID_ETIMEOUT on purpose")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(autouse=True)
def no_fp16_range_fallback(request):
    """No fixture, golden or synthetic, may ever drive an activation out of the fp16 range: a JMID_ERANGE turns a call into an
    exact-fp32 rerun at ~5x the latency (forecaster.py), so every -m gpu test asserts that none happened - unless it provokes
    one on purpose (marker ``expects_erange``)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from safe_interactive_crowdnav_amd import engine as EN, forecaster as FC
    n0, f0, t0 = len(EN.ERANGE_EVENTS), FC.ERANGE_FALLBACKS, len(EN.TIMEOUT_EVENTS)
    yield
    if request.node.get_closest_marker("expects_erange") is None:
        assert len(EN.ERANGE_EVENTS) == n0, EN.ERANGE_EVENTS[n0:]
        assert FC.ERANGE_FALLBACKS == f0
    # ... and no workgroup of a one-launch GEMM + LayerNorm may ever give up waiting for its partners (JMID_ETIMEOUT: the engine retries
    # and records it) unless the test withholds a partner or crowds the chip on purpose (marker ``expects_timeout``)
    if request.node.get_closest_marker("expects_timeout") is None:
        assert len(EN.TIMEOUT_EVENTS) == t0, EN.TIMEOUT_EVENTS[t0:]
