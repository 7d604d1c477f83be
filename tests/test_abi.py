"""CPU-side checks of the drop-in boundary: the C-ABI library builds, loads and exports every symbol that
include/jmid_hip.h declares; without a GPU the compute path fails loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest
import torch

import __graft_entry__ as graft
from safe_interactive_crowdnav_amd import _lib

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    graft.build()
    return _lib.load_library()


def declared_functions():
    src = open(os.path.join(REPO, "include", "jmid_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(jmid_[a-z_0-9]+)\s*\(", src)))


def test_header_and_binding_agree(lib):
    names = declared_functions()
    assert len(names) >= 20
    assert sorted(_lib.SIGNATURES) == names
    for n in names:
        assert hasattr(lib, n), f"libjmid_hip.so does not export {n}"


def test_version_and_class_names(lib):
    assert b"gfx950" in lib.jmid_version()
    n = lib.jmid_kernel_class_count()
    names = [lib.jmid_kernel_class_name(i).decode() for i in range(n)]
    assert "attention" in names and "gemm_qkv" in names and len(set(names)) == n


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful on a box without a GPU")
def test_create_fails_loudly_without_gpu(lib):
    assert lib.jmid_device_count() == 0
    h = _lib.Handle()
    rc = lib.jmid_create(C.byref(h), 0, _lib.NET_JMID, 256, 3, 4, 6)
    assert rc == -3 and not h.value
    assert b"no HIP device" in lib.jmid_last_error(None)
    from safe_interactive_crowdnav_amd.engine import JmidEngine, JmidError
    from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

    with pytest.raises(JmidError):
        JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=32), 0), joint=True)


def test_product_never_imports_oracle():
    pkg = os.path.join(REPO, "safe-interactive-crowdnav_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                txt = open(os.path.join(root, f)).read()
                assert "oracle" not in txt.replace("# oracle-free", ""), f"{f} mentions the oracle"
