"""Builds csrc/libjmid_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
from __future__ import annotations

import os
import shutil
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libjmid_hip.so")
# the diagnostics flavour: the same kernels + the jmid_dbg_* single-kernel entry points and the jmid_set_tuning experiment knobs
# (-DJMID_DIAGNOSTICS).  tests/ and tools/ load it (tests/conftest.py sets JMID_LIB); the product never does.
LIB_DIAG = os.path.join(CSRC, "libjmid_hip_diag.so")
# (The third flavour of rounds 4-5, -DJMID_EXPERIMENTS - seven attention / tail kernels that measured slower - was retired in round 6:
# docs/NOTEBOOK.md and profiles/r05_*_check.log are the record; `git revert` of that one commit brings the kernels back.)
# translation units of the host side (csrc/jmid_ctx.hpp says what each holds); every unit instantiates the kernels it launches
SOURCES = ["jmid_abi.hip", "jmid_weights.hip", "jmid_planner.hip", "jmid_profile.hip", "jmid_diag.hip"]


# what the last build_library() call of this process did, per flavour: "compiled" or "reused" (__graft_entry__.build() prints it)
LAST_BUILD = {}


def _source_digest(flags) -> str:
    """sha256 over every source the library is made of (csrc/*.hip|hpp|map, include/*.h), this file (the command line lives
    here) and the flags: a built .so is reused only when the stamp written next to it holds exactly this digest - file times do
    not survive a snapshot / checkout, contents do."""
    import hashlib
    h = hashlib.sha256(" ".join(flags).encode())
    for root in (CSRC, os.path.join(os.path.dirname(CSRC), "..", "include")):
        for f in sorted(os.listdir(root)):
            if f.endswith((".hip", ".hpp", ".h", ".map")):
                h.update(f.encode() + b"\0" + open(os.path.join(root, f), "rb").read())
    h.update(open(os.path.abspath(__file__), "rb").read())
    return h.hexdigest()


def library_path() -> str:
    """The in-tree library; JMID_LIB overrides it with an experimental build (tools/ only: A/B of kernel variants)."""
    return os.environ.get("JMID_LIB") or LIB


def build_library(force: bool = False, verbose: bool = False, diagnostics: bool = False) -> str:
    """Compile the HIP library in-tree (``diagnostics``: the -DJMID_DIAGNOSTICS flavour).
    Returns the path of the .so."""
    LIB = LIB_DIAG if diagnostics else globals()["LIB"]
    flavour = "diagnostics" if diagnostics else "production"
    # -ffp-contract=off: no implicit FMA contraction, so a value never depends on which template instance /
    # code path computed it (results are bit-identical across tile variants and chunkings); fmaf() is explicit.
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found: cannot build libjmid_hip.so")
    # -packed-fp32-ops: no v_pk_{add,mul,fma}_f32 in the device code.  hipcc forms them with crossed operand selects
    # (op_sel:[0,1] op_sel_hi:[1,0]) for float4 arithmetic, and on MI355X such an instruction returns wrong values in
    # lanes 48-63 when its wave shares a CU with waves of the LDS-DMA attention kernel (tools/concurrency_probe8.hip: 397 of
    # 400 overlaps, 0 with straight selects or other co-runners) - the cause of the run-to-run variation with several
    # chunks in flight (docs/NOTEBOOK.md section 3).  Same IEEE arithmetic without them (bit-identical results), and 2 % faster.
    # (The host pass of the same command line warns that the feature is unknown to x86: harmless.)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-value", "-ffp-contract=off",
             "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"] + (["-DJMID_DIAGNOSTICS"] if diagnostics else [])
    stamp, digest = LIB + ".stamp", _source_digest(flags)
    force = force or os.environ.get("JMID_FORCE_BUILD", "0") not in ("", "0")
    if not force and os.path.exists(LIB) and os.path.exists(stamp) and open(stamp).read().strip() == digest:
        LAST_BUILD[flavour] = "reused"
        if verbose:
            print(f"[build] {flavour}: reused {os.path.basename(LIB)} (sources + flags digest {digest[:12]} unchanged; "
                  "JMID_FORCE_BUILD=1 recompiles)")
        return LIB
    objdir = os.path.join(CSRC, "obj_diag" if diagnostics else "obj")
    os.makedirs(objdir, exist_ok=True)
    # the units compile side by side (the planner - every GEMM / attention instantiation of the denoise loop - is the long one)
    jobs = []
    for src in SOURCES:
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        jobs.append((src, obj, subprocess.Popen(cmd, cwd=CSRC, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    objs, failed = [], []
    for src, obj, proc in jobs:
        out, _ = proc.communicate()
        if proc.returncode != 0:
            failed.append(f"{src}:\n{out}")
        objs.append(obj)
    if failed:
        raise RuntimeError("hipcc failed:\n" + "\n".join(failed))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=libjmid.map", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd))
    proc = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(f"hipcc (link) failed:\n{proc.stdout}\n{proc.stderr}")
    with open(stamp, "w") as fh:
        fh.write(digest + "\n")
    LAST_BUILD[flavour] = "compiled"
    if verbose:
        print(f"[build] {flavour}: compiled {len(SOURCES)} units with hipcc for gfx950 -> {os.path.basename(LIB)} (digest {digest[:12]})")
    return LIB


if __name__ == "__main__":
    import sys
    print(build_library(force=True, verbose=True))
    print(build_library(force=True, verbose=True, diagnostics=True))
