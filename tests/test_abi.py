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
    """Every function include/jmid_hip.h declares is bound, and exported by the flavour the header says: the production
    library the product loads exports the ABI proper and NOT the jmid_dbg_* diagnostics; the diagnostics flavour (what the tests
    load: tests/conftest.py) exports both."""
    from safe_interactive_crowdnav_amd.build import LIB, LIB_DIAG
    names = declared_functions()
    assert len(names) >= 20
    assert sorted(list(_lib.SIGNATURES) + list(_lib.DIAG_SIGNATURES)) == names
    assert all(n.startswith("jmid_dbg_") for n in _lib.DIAG_SIGNATURES) and not any(n.startswith("jmid_dbg_") for n in _lib.SIGNATURES)
    prod, diag = C.CDLL(LIB), C.CDLL(LIB_DIAG)
    prod.jmid_version.restype = diag.jmid_version.restype = C.c_char_p
    assert b"diagnostics" in diag.jmid_version() and b"diagnostics" not in prod.jmid_version()
    for n in _lib.SIGNATURES:
        assert hasattr(prod, n) and hasattr(diag, n), f"the library does not export {n}"
    for n in _lib.DIAG_SIGNATURES:
        assert hasattr(diag, n), f"libjmid_hip_diag.so does not export {n}"
        assert not hasattr(prod, n), f"the production library exports the diagnostics entry point {n}"
    assert lib.has_diagnostics          # the flavour this test session runs on


def test_libraries_export_the_c_abi_and_nothing_else():
    """-fvisibility=hidden + the header's visibility block: the dynamic symbol table of either flavour holds the declared
    jmid_* entry points only (no C++ template instances or their static guards that a second copy of the library - tests load
    both flavours in one process - or the host application could interpose)."""
    import subprocess
    from safe_interactive_crowdnav_amd.build import LIB, LIB_DIAG
    for path, want in ((LIB, set(_lib.SIGNATURES)), (LIB_DIAG, set(_lib.SIGNATURES) | set(_lib.DIAG_SIGNATURES))):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        syms = {l.split()[-1] for l in out.splitlines() if l.strip()}
        # (HIP gives every __global__ function's host-side handle default visibility: those stay, nothing else does)
        host = {s for s in syms if not (s.startswith("_Z") and "_kernel" in s)}
        assert host == want, sorted(host ^ want)[:20]


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


def test_device_code_holds_no_packed_fp32_instructions(tmp_path):
    """build.py compiles with -packed-fp32-ops: v_pk_{add,mul,fma}_f32 with crossed operand selects go wrong in lanes 48-63
    when their wave shares a CU with the LDS-DMA attention kernel (docs/NOTEBOOK.md section 3, tools/concurrency_probe8.hip), and
    hipcc forms exactly those for float4 arithmetic.  The shipped library must not contain any."""
    import shutil
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    lib = os.path.join(REPO, "safe-interactive-crowdnav_amd", "csrc", "libjmid_hip.so")
    if not (os.path.exists(objdump) and os.path.exists(lib)):
        pytest.skip("needs llvm-objdump and the built library")
    work = tmp_path / "libjmid_hip.so"
    shutil.copy(lib, work)
    subprocess.run([objdump, "--offloading", str(work)], check=True, capture_output=True, cwd=tmp_path)
    images = [p for p in os.listdir(tmp_path) if "amdgcn" in p]
    assert images, "no device code object in the library"
    n_kernels = 0
    for img in images:
        dis = subprocess.run([objdump, "-d", str(tmp_path / img)], check=True, capture_output=True, text=True).stdout
        assert "v_mfma_f32_32x32x16_f16" in dis                  # it is the gfx950 code of the kernels
        n_kernels += dis.count("<_ZN4jmid")
        for op in ("v_pk_add_f32", "v_pk_mul_f32", "v_pk_fma_f32"):
            assert op not in dis, f"{op} found in the device code: build with -Xclang -target-feature -Xclang -packed-fp32-ops"
        # ... and no SCALED fp8 MFMA: it is an instruction pair (v_mfma_ld_scale_b32 + MFMA) that computed with a wrong scale next to
        # waves of another kernel on the same SIMD (tools/concurrency_probe9.hip); the F16MX kernels use the unscaled instruction
        assert "v_mfma_f32_32x32x64_f8f6f4" in dis
        for op in ("v_mfma_scale", "v_mfma_ld_scale"):
            assert op not in dis, f"{op} found in the device code"
    assert n_kernels > 50
