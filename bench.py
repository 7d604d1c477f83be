#!/usr/bin/env python3
"""Throughput bench of the JMID predictor hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

A "step" is ONE full predictor pass over one batch of synthetic scenes already resident in HBM:
context encoder -> 50-step batched DDIM reverse-denoising loop -> integrator -> per-episode metrics
(+ one gather of the metrics when N > 1: RCCL on the GPU box, gloo with --dist-backend gloo).  Episodes are
independent and shard across ranks (weak scaling: E episodes per GPU).  Metric: sampled trajectories / s =
N_gpus * E * N * K / time.

Default workload = BASELINE.json configs[2] ("256 parallel episodes x N=5 x K=20, 1 MI355X"), the configuration the
throughput target and the roofline are quoted on; the single-scene case (configs[1], E=1) is measured in the same run
and reported under "single_scene".

Precision modes.  `value` is the mode named by --precision (default f16mx: fp16 activation x split-fp16 weight with the
weight-lo correction term as one bf8 x bf8 MFMA per k64; >= the bf16 BASELINE.json names; an OPT-IN of the drop-in predictor
class `HumanTrajectoryForecasterSim`, whose own default is the fp32-class mode f16x3 - the line carries both:
`config.precision` and `config.class_default_precision`, and `modes.f16x3` is the class default's number).  Every mode listed in --modes (default "f16mx,f16x2,f16x3": f16x2 = the
same products with the correction term in fp16, f16x3 = fp32-class three-term products) gets THE SAME measurement - W warm-up steps, one untimed profiling step, K timed steps between
barriers, parity against the oracle on the same episodes - and is reported under `modes[<name>]` with the same keys;
the top-level keys are a copy of modes[--precision].

Output.  Rank 0 prints ONE COMPACT JSON line (< 4 KB: the contract keys, `roofline`, `cpu_baseline`, `parity`, one short row per
mode) as the last line of stdout; everything else (notes, sources, per-kernel tables, forecaster_e2e, worker scaling, the full
per-mode blocks) goes to the file named by `detail` in that line (--detail, default bench_detail.json next to this script).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

from safe_interactive_crowdnav_amd.engine import JmidEngine  # noqa: E402
from safe_interactive_crowdnav_amd.scene import build_scenes_batched, synthetic_episodes  # noqa: E402
from safe_interactive_crowdnav_amd.sweep import gather_metrics  # noqa: E402
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims  # noqa: E402

WORKLOADS = {
    # name: (episodes per GPU, N humans, K samples, H horizon, DDIM steps)
    "cfg2": (1, 5, 20, 12, 50),      # BASELINE.json configs[1]: one scene
    "cfg3": (256, 5, 20, 12, 50),    # configs[2]: 256 parallel episodes on one GPU
    "cfg4": (1, 25, 64, 12, 50),     # configs[3]: dense crowd, one scene
    "cfg5": (512, 5, 20, 12, 50),    # configs[4]: 4096 episodes sharded 512 / GPU
}
PEAK_TFLOPS = {"f32": 157.3, "f16x3": 2500.0, "f16x2": 2500.0, "f16mx": 2500.0}   # dense MFMA peaks, MI355X_MICROARCH.md
# MFMA FLOPs spent per algorithmic FLOP; f16x2: 2 in the GEMMs, (3 + 2) / 2 in attention (logits keep all three terms)
MFMA_PASSES = {"f32": 1, "f16x3": 3, "f16x2": 2, "f16mx": 1.5}   # f16mx: the fp8 correction term costs half an fp16 pass
MFMA_PASSES_ATTN = {"f32": 1, "f16x3": 3, "f16x2": 2.0, "f16mx": 1.5}   # f16x2: logits 3, P.V 1; f16mx: logits 1 + 2 x 0.5, P.V 1
DTYPE_TEXT = {
    "f32": "f32",
    "f16x3": "f32-class: fp16 hi/lo split operands, 3 MFMAs per product, fp32 accumulate",
    "f16x2": "fp16 activation x split-fp16 (hi + lo) weight, 2 MFMAs per product, fp32 accumulate; P.V with one fp16 plane of P; softmax logits, "
             "residual stream, LayerNorm and DDIM state at f32-class precision",
    "f16mx": "fp16 activation x split-fp16 (hi + lo) weight: A_hi x W_hi as fp16 MFMAs plus the correction term as ONE bf8 x bf8 MFMA "
             "per k64 (bf8 = top byte of the fp16 activation / of fp16(W_lo)): 1.5 MFMA passes per GEMM product, fp32 accumulate; "
             "softmax logits with all three terms (the two correction terms as bf8 x bf8 MFMAs too), P.V with one fp16 plane of P, residual stream, LayerNorm and "
             "DDIM state at f32-class precision",
}
PROF_CLASSES = ["gemm_qkv", "gemm_attn_out", "gemm_ff1", "gemm_ff2", "gemm_ff", "gemm_tail", "attention"]
# kernel class -> substring of the kernel names of that class in the per-call PMC summary (tools/pmc_call.sh)
PMC_KERNEL_OF_CLASS = {"attention": "attn_f16x3_dma_kernel", "gemm_qkv": ("gemm_f16x3_dma256x256_kernel<0, 2", "gemm_mx_kernel<0, 2"),
                       "gemm_ff1": ("gemm_f16x3_dma256x256_kernel<1, 1", "gemm_mx_kernel<1, 1"), "gemm_ff": "ff_ln_f16x3_kernel",
                       "gemm_attn_out": "gemm_ln", "gemm_ff2": "gemm_ln"}


def algorithmic_flops(dims: NetDims, joint: bool, E, A, K, T):
    """FLOPs of ONE denoise step for E episodes, per kernel class."""
    d, ff, dm, dl = dims.d_model, dims.d_ff, dims.d_mid, dims.d_low
    M = E * K * A * T
    S = K * A * T if joint else T
    fl = {
        "gemm_qkv": 2.0 * M * d * 3 * d * dims.tf_layer,
        "gemm_attn_out": 2.0 * M * d * d * dims.tf_layer,
        "gemm_ff1": 2.0 * M * d * ff * dims.tf_layer,
        "gemm_ff2": 2.0 * M * d * ff * dims.tf_layer,
        "gemm_tail": 2.0 * M * (d * dm + dm * dl),
        "attention": 4.0 * M * S * d * dims.tf_layer,
    }
    fl["gemm_ff"] = fl["gemm_ff1"] + fl["gemm_ff2"]      # the fused linear1 -> linear2 + LayerNorm kernel, when it runs
    return fl


def git_blob_sha1(path):
    """`git hash-object` of a file: lets a reader check which committed profile a bench field was taken from."""
    data = open(path, "rb").read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def pmc_summary(mode):
    """Newest committed per-call PMC summary of this mode (tools/pmc_call.sh -> profiles/rNN_pmc_call_<mode>.json)."""
    pdir = os.path.join(REPO, "profiles")
    cands = sorted((f for f in os.listdir(pdir) if f.endswith(f"_pmc_call_{mode}.json")), reverse=True) \
        if os.path.isdir(pdir) else []
    for f in cands:
        try:
            p = os.path.join(pdir, f)
            return json.load(open(p)), {"file": f"profiles/{f}", "git_blob_sha1": git_blob_sha1(p)}
        except Exception:
            continue
    return None, None


def sustained_mfma_tflops():
    """What a pure v_mfma_f32_32x32x16_f16 loop sustains on THIS box with fresh random operands (tools/mfma_peak.hip: every CU, two
    waves per SIMD, warm clocks; the chip is power-limited well below the 2.5 PFLOP/s dense peak).  Compiled and run next to the
    bench (hipcc is part of the image); None when that is not possible."""
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = os.path.join(REPO, "tools", "mfma_peak.hip")
    if not (os.path.exists(hipcc) and os.path.exists(src)):
        return None
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as td:
            exe = os.path.join(td, "mfma_peak")
            subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", src, "-o", exe], check=True, capture_output=True, timeout=180)
            out = subprocess.run([exe, "sustained"], check=True, capture_output=True, text=True, timeout=120).stdout
        tf = float(out.split("TFLOP/s")[0].split()[-1])
        mhz = float(out.split("shader clock")[1].split()[0])
        return {"tflops": tf, "shader_clock_mhz": mhz}
    except Exception:
        return None


DTYPE_SHORT = {"f32": "f32", "f16x3": "f32-class (fp16 hi+lo split operands, 3 MFMAs per product, fp32 accumulate)",
               "f16x2": "fp16 activations x split-fp16 weights (2 MFMAs per product), fp32 accumulate",
               "f16mx": "fp16 activations x split-fp16 weights (weight-lo term as one bf8 MFMA per k64), fp32 accumulate"}
COMPACT_LIMIT = 4000            # bytes; the driver parses the last stdout line and lost the 20 KB one of round 4


def _r(v, nd=4):
    return round(float(v), nd) if isinstance(v, (int, float)) and not isinstance(v, bool) else v


def compact_line(full, detail_path):
    """The ONE stdout line: contract keys + roofline + cpu_baseline + parity + one short row per mode.  `full` is the long result
    (written to `detail_path`); nothing here is computed, only selected."""
    from safe_interactive_crowdnav_amd.forecaster import DEFAULTS as CLASS_DEFAULTS
    cfg = full["config"]
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                "scaling", "vs_baseline") if k in full}
    out["dtype"] = DTYPE_SHORT.get(cfg["precision"], full["dtype"])
    out["data"] = full["data"]
    out["config"] = {"workload": cfg["workload_short"], "episodes_per_gpu": cfg["episodes_per_gpu"],
                     "total_episodes": cfg["total_episodes"], "humans": cfg["humans"], "samples": cfg["samples"],
                     "horizon": cfg["horizon"], "denoise_steps": cfg["denoise_steps"], "net": cfg["net"],
                     "precision": cfg["precision"], "class_default_precision": CLASS_DEFAULTS["precision"],
                     "lanes": cfg["lanes"], "dist_backend": cfg["dist_backend"]}
    out["ranks_seen"] = full.get("ranks_seen")
    out["gather_ms"] = full.get("gather_ms")
    pr = full.get("per_rank_ms_per_step")
    if pr:
        out["per_rank_ms_per_step"] = {"min": pr["min"], "max": pr["max"]}

    def roof_of(r):
        if not r:
            return None
        mb = r.get("mfma_busy") or {}
        ts = r.get("traffic_source") or {}
        return {"bound": r["bound"], "kernel": r["kernel"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"],
                "frac": r["frac"], "frac_of_sustained": r.get("frac_of_sustained"), "traffic": _r(r.get("traffic"), 0),
                "traffic_measured_in_run": ts.get("measured_in_run"), "flops_per_launch": r.get("flops_per_launch"),
                "avg_launch_ms": r["avg_launch_ms"], "launches": r["launches"],
                "mfma_busy": _r(mb.get("dominant_kernel")), "mfma_busy_whole_call": _r(mb.get("whole_call")),
                "path_achieved": r.get("path_achieved"), "path_frac": r.get("path_frac"),
                "sustained_peak": r.get("peak_sustained_random_operands")}
    if full.get("roofline"):
        out["roofline"] = roof_of(full["roofline"])
    if full.get("hbm"):
        h = full["hbm"]
        out["hbm"] = {"bytes_per_trajectory": h["bytes_per_trajectory"], "GBps": h["GBps"], "peak_GBps": h["peak_GBps"],
                      "frac": h["frac"], "measured_in_run": (h.get("source") or {}).get("measured_in_run")}
    if full.get("cpu_baseline"):
        c = full["cpu_baseline"]
        out["cpu_baseline"] = {"value": c["value"], "unit": c["unit"], "cores": c["cores"], "kind": c["kind"],
                               "processes": c["processes"], "threads_per_process": c["threads_per_process"],
                               "hardware_threads": c["hardware_threads"], "host_physical_cores": c["host_physical_cores"],
                               "sample": c["sample_short"], "note": c["cores_short"]}
    if full.get("parity"):
        q = full["parity"]
        out["parity"] = {"mean_ADE_vs_oracle_m": _r(q["mean_ADE_vs_oracle_m"], 9), "gate_m": q["gate_m"], "pass": q["pass"],
                         "episodes": q["episodes"], "precision": q["precision"]}
    modes = {}
    for m, v in full.get("modes", {}).items():
        row = {"value": v["value"], "ms_per_step": v["ms_per_step"]}
        if v.get("roofline"):
            row["roofline_frac"] = v["roofline"]["frac"]
            row["avg_launch_ms"] = v["roofline"]["avg_launch_ms"]
        if v.get("parity"):
            row["mean_ADE_vs_oracle_m"] = _r(v["parity"]["mean_ADE_vs_oracle_m"], 9)
            row["pass"] = v["parity"]["pass"]
        modes[m] = row
    out["modes"] = modes
    if full.get("single_scene"):
        ss = full["single_scene"]
        out["single_scene"] = {"workload": "cfg2: 1 scene", "entry": ss.get("entry"), "ms_per_call": ss["ms_per_call"],
                               "staged_ms_per_call": ss.get("staged_ms_per_call"), "traj_per_s": ss["traj_per_s"],
                               "modes": {m: t["ms_per_call"] for m, t in ss["modes"].items()}}
    if full.get("sweep_metrics"):
        out["sweep_episodes"] = full["sweep_metrics"]["episodes"]
        out["rows_in_episode_order"] = full["sweep_metrics"].get("rows_in_episode_order")
    if full.get("rank0_affinity"):
        out["rank0_affinity"] = full["rank0_affinity"]
    if full.get("host_feed"):
        out["host_feed_ms"] = full["host_feed"]["ms"]
    out["detail"] = os.path.relpath(detail_path, REPO) if os.path.abspath(detail_path).startswith(REPO) else detail_path
    line = json.dumps(out, separators=(",", ":"))
    # never exceed the limit and never lose the line: shed the optional blocks one by one, then the notes inside the kept ones, and
    # at the very end fall back to the contract keys alone (the driver parses this line; everything is in `detail` anyway)
    for drop in ("single_scene", "hbm", "per_rank_ms_per_step", "rank0_affinity", "host_feed_ms", "gather_ms", "modes", "parity"):
        if len(line) <= COMPACT_LIMIT:
            break
        out.pop(drop, None)
        line = json.dumps(out, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT:
        for blk, keys in (("cpu_baseline", ("value", "unit", "cores", "kind", "sample")),
                          ("roofline", ("bound", "achieved", "peak", "unit", "frac", "traffic"))):
            if isinstance(out.get(blk), dict):
                out[blk] = {k: (str(out[blk][k])[:120] if isinstance(out[blk][k], str) else out[blk][k]) for k in keys if k in out[blk]}
        out["config"] = {"workload": str(out["config"].get("workload"))[:160], "precision": out["config"].get("precision")}
        line = json.dumps(out, separators=(",", ":"))
    if len(line) > COMPACT_LIMIT:
        line = json.dumps({k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                               "scaling", "vs_baseline", "detail") if k in out}, separators=(",", ":"))
    json.loads(line)
    return line



_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N copies of this command line as ranks 0..N-1 of one node
    (rendezvous on 127.0.0.1, a free port), wait for all of them, return the worst exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rcs = [p.wait() for p in procs]
    return max((abs(rc) for rc in rcs), default=0)


def _cpulist(text):
    """'0-3,8,10-11' (sysfs cpulist) -> sorted list of ints."""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        out.extend(range(int(lo), int(hi or lo) + 1))
    return sorted(out)


def rank_cpu_share(allowed, local_rank, local_world, numa_of_rank=None, cpus_of_node=None):
    """The host cores of one rank of a multi-rank launch: with NUMA information (``numa_of_rank[r]`` = node of rank r's GPU,
    ``cpus_of_node[n]`` = that node's cores) the ranks whose GPUs hang off one node split THAT node's allowed cores evenly, in rank
    order; without it (or when a node has fewer allowed cores than ranks) the allowed cores are split into ``local_world`` contiguous
    shares.  Never empty, never outside ``allowed``; shares of different ranks are disjoint whenever there are enough cores."""
    allowed = sorted(allowed)

    def split(cores, k, n):
        base, rem = divmod(len(cores), n)
        lo = k * base + min(k, rem)
        return cores[lo: lo + base + (1 if k < rem else 0)]

    if numa_of_rank and cpus_of_node and numa_of_rank[local_rank] is not None and numa_of_rank[local_rank] >= 0:
        node = numa_of_rank[local_rank]
        peers = [r for r in range(local_world) if numa_of_rank[r] == node]
        cores = [c for c in cpus_of_node.get(node, []) if c in set(allowed)]
        if len(cores) >= len(peers):
            return split(cores, peers.index(local_rank), len(peers))
    if len(allowed) >= local_world:
        return split(allowed, local_rank, local_world)
    return allowed


def pin_rank(local_rank, local_world, dev_of_rank):
    """Per-rank CPU placement of a multi-rank launch (the driver's `torch.distributed.run` gives every rank the whole machine): this
    process - its launch loop, its 512 `torch.randn` calls per step, torch's intra-op threads - is confined to its share of the cores,
    on the NUMA node of its GPU when sysfs says which (/sys/bus/pci/devices/<bdf>/numa_node).  Returns what was done, for the JSON."""
    if local_world <= 1 or not hasattr(os, "sched_setaffinity"):
        return None
    numa, nodes = [None] * local_world, {}
    try:
        for r in range(local_world):
            p = torch.cuda.get_device_properties(dev_of_rank(r))
            bdf = f"{getattr(p, 'pci_domain_id', 0):04x}:{p.pci_bus_id:02x}:{p.pci_device_id:02x}.0"
            with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as fh:
                numa[r] = int(fh.read().strip())
        for n in {v for v in numa if v is not None and v >= 0}:
            with open(f"/sys/devices/system/node/node{n}/cpulist") as fh:
                nodes[n] = _cpulist(fh.read())
    except (OSError, AttributeError, ValueError):
        numa, nodes = None, None
    allowed = sorted(os.sched_getaffinity(0))
    share = rank_cpu_share(allowed, local_rank, local_world, numa, nodes)
    try:
        os.sched_setaffinity(0, share)
    except OSError:
        return None
    torch.set_num_threads(max(1, min(len(share), 8)))
    return {"cores": len(share), "first": share[0], "last": share[-1], "numa_node": numa[local_rank] if numa else None}


def cpu_worker(job):
    """One worker process of the CPU baseline: the oracle (oracle/jmid_oracle.py, torch-CPU fp32) on its share of the sample."""
    seed, threads, joint, K, steps, eps = job
    import torch as T
    T.set_num_threads(threads)
    from oracle import jmid_oracle as O
    wt = JMIDWeights.from_seed(NetDims(ctx_dim=256), seed).tensors
    out = []
    with T.no_grad():
        for e, xs, nb, em, xT, p0e in eps:
            ctx_c = O.encode_context(wt, T.from_numpy(xs), T.from_numpy(nb), T.from_numpy(em))
            v = O.denoise(wt, ctx_c, T.from_numpy(xT), sample=K, step=steps, joint=joint)
            out.append((e, O.integrate(v, T.from_numpy(p0e), 0.25).numpy()))
    return out


_CPU_READY = None


def physical_cores():
    """Physical cores of the host (unique (socket, core) pairs of /proc/cpuinfo); None when it cannot be told."""
    try:
        seen, phys, core = set(), None, None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                core = line.split(":")[1].strip()
            elif not line.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or None
    except OSError:
        return None


def cpu_init(ready, seed, threads, joint, K, eps1):
    """Start-up of one CPU-baseline worker (imports, weights from the seed, a 2-step oracle run), kept out of the timed
    region: the parent waits until every worker has counted itself ready."""
    cpu_worker((seed, threads, joint, K, 2, eps1))
    with ready.get_lock():
        ready.value += 1


def forecaster_e2e(weights, dev_id, modes):
    """predict_ret_best() of the drop-in class, host + device, the way the MPC calls it once per control step
    (sicnav_acados.py:1641-1651): BASELINE cfg2 (N=5, K=20 -> 20, H=12, 50 steps) and the reference's SHIPPED operating point
    (test_time_configs/mid_jp.yaml: K=100 -> 15 kept, H=8, 2 denoise steps; N=3 humans)."""
    import tempfile
    from safe_interactive_crowdnav_amd.forecaster import HumanTrajectoryForecasterSim, write_configs

    class St:
        def __init__(self, p):
            self.position = (float(p[0]), float(p[1]))

    out = {}
    rng = np.random.default_rng(5)
    for name, (N, K, k, H, step) in {"cfg2": (5, 20, 20, 12, 50), "shipped": (3, 100, 15, 8, 2)}.items():
        res = {}
        for m in modes:
            with tempfile.TemporaryDirectory() as td:
                env, yp = write_configs(td, joint=True, ctx_dim=256, N=N, K=K, k_ret=k, H=H, step=step, time_step=0.25)
                f = HumanTrajectoryForecasterSim(env, yp, weights=weights, device_id=dev_id, precision=m)
            p0 = rng.uniform(-1.0, 1.0, (N, 2))
            v = rng.uniform(-0.4, 0.4, (N, 2))
            for i in range(7):
                f.update_state_hists(St((0.0, -1.5 + 0.05 * i)), [St(p0[j] + v[j] * 0.25 * i) for j in range(N)], 0.25 * i)
            for _ in range(3):
                f.predict_ret_best()
            reps, acc, t0 = 10, {}, time.perf_counter()
            for i in range(reps):
                f.update_state_hists(St((0.0, -1.15 + 0.05 * i)), [St(p0[j] + v[j] * 0.25 * (7 + i)) for j in range(N)], 0.25 * (7 + i))
                fc, lw = f.predict_ret_best()
                for kk, vv in f.timings.items():
                    acc[kk] = acc.get(kk, 0.0) + vv
            wall = (time.perf_counter() - t0) / reps
            assert fc.shape == (N, k, H + 1, 2) and lw.shape == (N, k)
            res[m] = {"ms_per_call": round(1e3 * wall, 3), "host_scene_ms": round(acc["scene_ms"] / reps, 3),
                      "device_ms": round(acc["device_ms"] / reps, 3), "topk_ms": round(acc["topk_ms"] / reps, 3),
                      "host_assemble_ms": round(acc["assemble_ms"] / reps, 3), "erange_fallbacks": f.erange_fallbacks}
        out[name] = {"workload": f"N={N} humans, K={K} samples -> {k} returned, H={H}, {step} denoise steps; "
                                 "update_state_hists + predict_ret_best() per call, host NumPy scene build included",
                     "topk": "joint KDE on the device (jmid_topk)" if k < K else "not needed (all samples returned)",
                     "modes": res}
    # the batched form of the same call: predict_ret_best() for 64 independent episodes per predict_batch()
    # at the shipped operating point - vectorised host batch builder, one encode + denoise + jmid_topk per cluster size
    from safe_interactive_crowdnav_amd.forecaster import predict_batch
    from safe_interactive_crowdnav_amd.engine import JmidEngine
    Eb, N, K, k, H, F = 64, 3, 100, 15, 8, 6
    p0 = rng.uniform(-1.0, 1.0, (Eb, 1, N, 2))
    v = rng.uniform(-0.4, 0.4, (Eb, 1, N, 2))
    hum = p0 + v * 0.25 * np.arange(F)[None, :, None, None]
    rob = np.stack([np.zeros(F), -1.5 + 0.05 * np.arange(F)], axis=-1)[None].repeat(Eb, axis=0)
    eng = JmidEngine(weights, joint=True, device_id=dev_id, step=2)
    res = {}
    for m in modes:
        kw = dict(num_samples=K, num_ret_samples=k, horizon=H, time_step=0.25, precision=m)
        for _ in range(2):
            predict_batch(eng, hum, rob, range(Eb), **kw)
        reps, t0 = 5, time.perf_counter()
        for _ in range(reps):
            fc, lw, inc = predict_batch(eng, hum, rob, range(Eb), **kw)
        wall = (time.perf_counter() - t0) / reps
        assert fc.shape == (Eb, N, k, H + 1, 2) and lw.shape == (Eb, N, k)
        res[m] = {"ms_per_call": round(1e3 * wall, 3), "ms_per_episode": round(1e3 * wall / Eb, 4),
                  "episodes_per_s": round(Eb / wall, 1)}
    eng.close()
    out["shipped_batched"] = {"workload": f"predict_batch(): {Eb} episodes per call, each N={N}, K={K} -> {k}, H={H}, 2 denoise steps "
                                          "(x_T drawn per episode from its own torch generator, as the per-episode class does)",
                              "topk": "joint KDE on the device, all episodes of a cluster size in one jmid_topk call", "modes": res}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--episodes-per-gpu", type=int, default=0)
    ap.add_argument("--precision", default="f16mx", choices=["f32", "f16x3", "f16x2", "f16mx"], help="the mode `value` is quoted on")
    ap.add_argument("--modes", default="f16mx,f16x2,f16x3",
                    help="comma list of modes measured identically in this run (the --precision mode is always included)")
    ap.add_argument("--net", default="jmid", choices=["jmid", "imid"])
    ap.add_argument("--chunk", type=int, default=0, help="episodes per pass of the denoise loop (0 = auto)")
    ap.add_argument("--cpu-episodes", type=int, default=-1,
                    help="episodes timed on the host for cpu_baseline and checked for parity (0 = skip; default: 12, or one per "
                         "worker process when the host has more workers than that)")
    ap.add_argument("--cpu-one-process", dest="cpu_all_cores", action="store_false",
                    help="cpu_baseline in ONE process (default: as many worker processes as the host cores hold)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the forecaster_e2e leg (class surface, host + device)")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--lanes", type=int, default=2,
                    help="chunks of the denoise loop in flight at once (1..4; the library's default is 2; results are the same bits)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--scenes", default="cv", choices=["cv", "orca", "square", "hallway", "hallway_sfm"],
                    help="synthetic histories: cv = constant-velocity agents (SURVEY 8d), orca = batched circle-crossing "
                         "crowds of ORCA agents, square = the same under the square-crossing rule (episodes.py), hallway = the reference's shipped scenario: a corridor with walls and "
                         "orca_plus humans, hallway_sfm = the same with the reference's social-force humans (crowd_env.py; SURVEY 8f row f3); all "
                         "agents forced in-cluster either way")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (one rank per GPU); gloo = host gather (lets several ranks share one GPU)")
    ap.add_argument("--device", type=int, default=-1, help="HIP device of this rank (-1 = LOCAL_RANK)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="weak = --episodes-per-gpu (workload default) on every GPU; strong = a FIXED total (--total-episodes, default "
                         "4096 = BASELINE configs[4]) block-partitioned over the ranks (sweep.shard_range); auto = weak on one GPU, "
                         "strong on several - per-N values then divide into a speed-up")
    ap.add_argument("--total-episodes", type=int, default=4096, help="episodes of the whole job with --scaling strong")
    ap.add_argument("--no-pmc", action="store_true",
                    help="do not re-run one 51-episode call under rocprofv3 --pmc after the timed region (roofline.traffic / mfma_busy "
                         "then come from the committed profiles/ summary, labelled as such) and do not run the sustained-MFMA probe")
    ap.add_argument("--detail", default=os.path.join(REPO, "bench_detail.json"),
                    help="file the full (long) result goes to; the stdout line stays compact and names this path")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise the process group and run every collective even with one rank (exercises the RCCL calls "
                         "of the N > 1 path on a one-GPU box)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # launched plainly (`python bench.py --gpus N`): spawn the N ranks ourselves, one process per GPU, with the
        # environment torch.distributed.run would give them; rank 0 prints the JSON line
        raise SystemExit(spawn_ranks(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    dev_id = local_rank if args.device < 0 else args.device
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    affinity = pin_rank(local_rank, local_world, (lambda r: r) if args.device < 0 else (lambda r: args.device))
    torch.cuda.set_device(dev_id)
    dev = torch.device("cuda", dev_id)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    coll_dev = dev if args.dist_backend == "nccl" else torch.device("cpu")

    E, N, K, H, steps50 = WORKLOADS[args.workload]
    if args.episodes_per_gpu > 0:
        E = args.episodes_per_gpu
    scaling = args.scaling if args.scaling != "auto" else ("strong" if world > 1 and args.episodes_per_gpu <= 0 else "weak")
    ep_lo = rank * E                       # first episode (global index) of this rank
    if scaling == "strong":                # fixed total, block partition: rank g owns [g T / G, (g + 1) T / G)  
        from safe_interactive_crowdnav_amd.sweep import shard_range
        ep_lo, ep_hi = shard_range(args.total_episodes, rank, world)
        E = ep_hi - ep_lo
        if E <= 0:
            raise SystemExit(f"--total-episodes {args.total_episodes} leaves rank {rank} of {world} without work")
    total_eps = args.total_episodes if scaling == "strong" else E * world
    joint = args.net == "jmid"
    dims = NetDims(ctx_dim=256)
    weights = JMIDWeights.from_seed(dims, args.seed)
    eng = JmidEngine(weights, joint=joint, device_id=dev_id, step=steps50)
    eng.set_chunk_episodes(args.chunk)
    eng.set_tuning("lanes", args.lanes)
    modes = [args.precision] + [m for m in args.modes.split(",") if m and m != args.precision]
    for m in modes:
        if m not in PEAK_TFLOPS:
            raise SystemExit(f"unknown mode {m}")

    # ---- synthetic scene batches, resident in HBM before the timed region (what it costs to put them there is timed too and
    # reported as `host_feed`: it is OUTSIDE `value`, which is quoted with inputs already resident, as the bench contract says)
    t_feed = time.perf_counter()
    if args.scenes in ("orca", "square", "hallway", "hallway_sfm"):
        from safe_interactive_crowdnav_amd.episodes import history_windows, simulate_crossing
        frame = 12                       # 3 s into the episode: the crowd is interacting
        if args.scenes.startswith("hallway"):     # the reference's SHIPPED scenario (env.config [sim] test_sim = hallway): walls and
            # orca_plus humans - or its social-force humans ([humans] policy = sfm: pinned to the reference end to end)
            from safe_interactive_crowdnav_amd.crowd_env import HallwayConfig, simulate_hallway
            hc = HallwayConfig(human_policy="sfm" if args.scenes == "hallway_sfm" else "orca_plus")
            sim = simulate_hallway(E, N, frame + H, seed=args.seed * 1000 + rank, cfg=hc, robot="goal" if args.scenes == "hallway_sfm" else "orca")
        else:
            sim = simulate_crossing(E, N, frame + H, seed=args.seed * 1000 + rank, rule="square_crossing" if args.scenes == "square" else "circle_crossing")
        hum, rob = history_windows(sim, frame)
        syn = build_scenes_batched(hum, rob, 0.25, force_all_in_cluster=True)
        syn["gt"] = np.ascontiguousarray(sim["human_xy"][:, frame + 1:frame + 1 + H].transpose(0, 2, 1, 3), np.float32)
    else:
        syn = synthetic_episodes(E, N, seed=args.seed * 1000 + rank, horizon=H)
    A = N
    x_st = torch.from_numpy(syn["x_st"].reshape(E * A, 6, 6)).to(dev)
    nbr = torch.from_numpy(syn["nbr_sum"].reshape(E * A, 2, 6, 6)).to(dev)
    emask = torch.from_numpy(syn["edge_mask"].reshape(E * A, 2)).to(dev)
    p0 = torch.from_numpy(syn["p0"]).to(dev)
    gt = torch.from_numpy(syn["gt"]).to(dev)
    x_T_host = torch.stack([torch.randn([K * A, H, 2], generator=torch.Generator().manual_seed(
        args.seed + ep_lo + e)) for e in range(E)])      # per GLOBAL episode: results do not depend on the partition
    x_T = x_T_host.to(dev)
    torch.cuda.synchronize()
    host_feed_ms = 1e3 * (time.perf_counter() - t_feed)

    def one_step(precision):
        ctx = eng.encode(x_st, nbr, emask)
        vel, pos = eng.denoise(x_T, ctx.view(E, A, -1), p0, dt=0.25, precision=precision, want_vel=False)
        met = eng.episode_metrics(pos, gt)
        eng.synchronize()            # the library runs on its own stream
        return pos, met

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    fl = algorithmic_flops(dims, joint, E, A, K, H)
    traj = total_eps * A * K               # the whole job's trajectories per step

    def fill_pmc(res, precision, pmc, src):
        """roofline.traffic / mfma_busy / hbm of a mode from a per-call PMC summary: the one measured in this run
        (tools/pmc_summary.py re-runs one 51-episode call under rocprofv3 --pmc after the timed region) or, failing that, the
        newest committed profiles/rNN_pmc_call_<mode>.json - `source` says which, with the file's git blob hash."""
        ctx, roof = res.pop("_pmc_ctx", None), res.get("roofline")
        if not (ctx and roof and pmc and joint and (N, K, H, steps50) == (5, 20, 12, 50)):
            return
        dom = ctx["dom"]
        kname = PMC_KERNEL_OF_CLASS.get(dom)
        knames = (kname,) if isinstance(kname, str) else (kname or ())
        rows = [v for k, v in pmc.get("kernels", {}).items() if any(n in k for n in knames) and v.get("launches", 0) >= 50]
        if rows:
            r = max(rows, key=lambda v: v["hbm_bytes_per_launch"])
            # per launch of the SAME launches `achieved` is quoted on (the exclusive pass runs the one-lane chunk plan)
            chunks_per_step = ctx["launches"] / (steps50 * ctx["passes"] * dims.tf_layer)
            tokens_per_launch = E * A * K * H / chunks_per_step
            roof["traffic"] = r["hbm_bytes_per_launch"] * tokens_per_launch / pmc["tokens"]
            roof["traffic_source"] = dict(src, kernel=r.get("name"), measured_at_tokens=pmc["tokens"],
                                          note="HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE of the production "
                                               "kernel inside one whole predictor call, scaled to this run's tokens per launch")
            if "mfma_busy" in r:
                roof["mfma_busy"] = {"dominant_kernel": r["mfma_busy"], "whole_call": pmc.get("whole_call_mfma_busy"),
                                     "source": src}
        bpt = pmc["call"]["hbm_bytes_per_trajectory"]
        res["hbm"] = {"bytes_per_trajectory": bpt, "GBps": round(bpt * res["value"] / world / 1e9, 1),
                      "peak_GBps": 8000.0, "frac": round(bpt * res["value"] / world / 8e12, 4), "source": src,
                      "note": "PMC bytes (2 x FETCH_SIZE + WRITE_SIZE) of one whole predictor call on one chunk, "
                              "per trajectory, times this run's per-GPU traj/s",
                      "model_bytes_per_trajectory": {"layer_streamed_bf16": 34.9e6, "minimal": 9600}}

    def measure(precision):
        """W warm-up steps, one untimed profiling step, K timed steps between barriers: identical for every mode."""
        for _ in range(args.warmup):
            one_step(precision)
        log(f"[{precision}] warm-up done")
        # HIP-event profiling serialises the launches it brackets (~3 % of a cfg3 step when every GEMM and attention
        # launch carries events).  So: ONE untimed step with events on all classes gives the per-class table and names
        # the dominant class; the timed region then brackets only that class.
        prof_all, dom = {}, None
        if not args.no_profile:
            eng.set_tuning("lanes", 1)
            eng.profile_enable(PROF_CLASSES)
            eng.profile_reset()
            one_step(precision)
            prof_all = eng.profile_get()
            eng.profile_disable()
            eng.set_tuning("lanes", args.lanes)
            dom = max(PROF_CLASSES, key=lambda c: prof_all.get(c, (0, 0.0))[1])
            eng.profile_enable([dom])
            eng.profile_reset()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pos, met = one_step(precision)
        own = time.perf_counter() - t0         # this rank's own K steps (no collective inside: episodes are independent)
        barrier()
        elapsed = time.perf_counter() - t0
        rank_ms = [1e3 * own / args.steps]
        if use_dist:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            mine = torch.tensor([own], dtype=torch.float64, device=coll_dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            rank_ms = [1e3 * float(v.item()) / args.steps for v in every]
        # the sweep's ONE collective: the per-episode metric rows of the last step, gathered to rank 0 after the timed steps
        #  and timed on its own - it is off the data path and says nothing about the kernels
        # (a fifth column, added outside the timed steps: the GLOBAL index of the episode a row belongs to, so that rank 0 can see that
        #  the gathered rows are in episode order)
        met = torch.cat([met, torch.arange(ep_lo, ep_lo + met.shape[0], device=met.device, dtype=met.dtype)[:, None]], dim=1)
        tg = time.perf_counter()
        allm = gather_metrics(met, total_eps, force=args.force_dist)
        gather_ms = 1e3 * (time.perf_counter() - tg)
        prof = eng.profile_get() if not args.no_profile else {}
        eng.profile_disable()
        log(f"[{precision}] timed region: {elapsed:.3f}s for {args.steps} steps")
        res = {"value": round(traj * args.steps / elapsed, 2), "unit": "traj/s", "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(1e3 * elapsed / args.steps, 3), "dtype": DTYPE_TEXT[precision],
               "ranks_seen": int(dist.get_world_size()) if use_dist else 1,
               "per_rank_ms_per_step": {"min": round(min(rank_ms), 3), "max": round(max(rank_ms), 3),
                                        "all": [round(v, 3) for v in rank_ms]},
               "gather_ms": round(gather_ms, 3)}
        if prof and rank == 0:
            per = {}
            for cls in PROF_CLASSES:       # untimed profiling step (one pass over the batch)
                n, ms = prof_all.get(cls, (0, 0.0))
                if n:
                    per[cls] = {"launches": n, "avg_ms": ms / n, "total_ms": ms,
                                "tflops": fl[cls] * steps50 / (ms * 1e-3) / 1e12}
            n, ms = prof[dom]              # the dominant class again, inside the timed region
            dom_t = {"launches": n, "avg_ms": ms / n, "total_ms": ms,
                     "tflops": fl[dom] * steps50 * args.steps / (ms * 1e-3) / 1e12}
            peak = PEAK_TFLOPS[precision]
            passes = (MFMA_PASSES_ATTN if dom == "attention" else MFMA_PASSES)[precision]
            # one chunk in flight: the timed region's own events ARE the kernel's exclusive launch durations (contract);
            # with lanes > 1 they are inflated by the overlap, so the exclusive untimed pass is quoted instead
            dom_x = dom_t if args.lanes == 1 else per[dom]
            # every GEMM / attention class runs in every layer; fused classes absorb the ones they replace
            ran = [c for c in PROF_CLASSES if c in per]
            path_flops = sum(fl[c] for c in ran if not (c == "gemm_ff" and "gemm_ff1" in per))
            path_tflops = path_flops * steps50 * args.steps / elapsed / 1e12
            roof = {"bound": "mfma", "kernel": dom, "achieved": round(dom_x["tflops"], 2), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(dom_x["tflops"] / peak, 4), "traffic": None,
                    "frac_of_split_peak": round(dom_x["tflops"] * passes / peak, 4),
                    "mfma_passes_per_product": passes,
                    "flops_per_launch": fl[dom] * steps50 * (args.steps if args.lanes == 1 else 1) / dom_x["launches"],
                    "avg_launch_ms": round(dom_x["avg_ms"], 4), "launches": dom_x["launches"],
                    "measured": ("HIP events on the library's stream around every launch of this kernel class in the "
                                 "timed region" if args.lanes == 1 else
                                 "HIP events on the library's stream, one full pass over the batch with ONE chunk "
                                 "in flight (the kernel has the GPU to itself; the one-lane chunk plan: 51-52 episodes per "
                                 "launch), untimed, in this run"),
                    "path_achieved": round(path_tflops, 2), "path_frac": round(path_tflops / peak, 4),
                    "timed_region": {"lanes": args.lanes, "launches": dom_t["launches"], "avg_launch_ms": round(dom_t["avg_ms"], 4),
                                     "note": "the same kernel class bracketed by HIP events inside the timed region; with lanes > 1 "
                                             "a launch shares the GPU with the other chunk's kernels, so this is not the kernel's "
                                             "own duration (path_achieved is the overlapped whole-path rate)"},
                    "peak_sustained_random_operands": 1660.0,
                    "note": "peak = dense fp16 MFMA of MI355X_MICROARCH.md; a pure MFMA loop with fresh random operands "
                            "sustains 1.66 PFLOP/s at the 1.4 kW power cap (tools/mfma_peak.hip, warm clocks); this mode "
                            f"spends {passes} MFMA FLOPs per algorithmic FLOP of this kernel; path_achieved = algorithmic "
                            "FLOPs of all MFMA kernel classes / wall time of the timed region"}
            # what fill_pmc() needs to scale a per-call PMC summary to this run's launches
            res["_pmc_ctx"] = {"dom": dom, "launches": dom_x["launches"], "passes": (args.steps if args.lanes == 1 else 1)}
            res["roofline"] = roof
            res["kernels"] = {c: {"launches": v["launches"], "avg_ms": round(v["avg_ms"], 4),
                                  "total_ms": round(v["total_ms"], 2), "tflops": round(v["tflops"], 2)}
                              for c, v in per.items()}
        if rank == 0:
            res["sweep_metrics"] = {"episodes": int(allm.shape[0]), "mean_ADE_m": float(np.nanmean(allm[:, 0])),
                                    "mean_minADE_m": float(np.nanmean(allm[:, 1])),
                                    "mean_FDE_m": float(np.nanmean(allm[:, 2])),
                                    "rows_in_episode_order": bool(np.array_equal(allm[:, -1], np.arange(allm.shape[0], dtype=allm.dtype)))}
        return res, pos

    log(f"inputs resident: E={E} A={A} K={K} H={H} modes={modes} world={world} backend={args.dist_backend}")
    results, last_pos = {}, {}
    for m in modes:
        results[m], last_pos[m] = measure(m)

    if rank != 0:
        if use_dist:
            dist.barrier()
            dist.destroy_process_group()
        return

    # ---- PMC counters of THIS run's build on THIS box (headline mode): one whole 51-episode call re-run under rocprofv3,
    # one counter group per pass (kernel-trace + pmc only), after the timed region; the other modes - and this one if
    # rocprofv3 is missing or a pass fails - fall back to the committed profiles/ summary of the mode
    for m in modes:
        pmc, src = None, None
        if m == args.precision and world == 1 and not args.no_pmc and joint and (N, K, H, steps50) == (5, 20, 12, 50):
            try:
                sys.path.insert(0, os.path.join(REPO, "tools"))
                import pmc_summary as PS
                tp = time.perf_counter()
                pmc = PS.collect(m, episodes=51, timeout_s=150)
                if pmc:
                    src = {"measured_in_run": True, "command": pmc.pop("_command"), "seconds": round(time.perf_counter() - tp, 1)}
                    log(f"[{m}] PMC passes in this run: {src['seconds']} s, {pmc['call']['hbm_bytes_per_trajectory']} B / trajectory")
            except Exception as ex:          # never let a profiler problem take the bench line down
                log(f"[{m}] in-run PMC failed ({type(ex).__name__}: {ex}); falling back to the committed summary")
                pmc = None
        if pmc is None:
            pmc, src = pmc_summary(m)
            if src:
                src = dict(src, measured_in_run=False)
        fill_pmc(results[m], m, pmc, src)

    # the MFMA rate this box sustains (power-limited), measured next to the bench: frac_of_sustained = MFMA FLOPs the dominant
    # kernel ISSUES per second / that rate
    sus = sustained_mfma_tflops() if world == 1 and not args.no_pmc else None
    for m in modes:
        roof = results[m].get("roofline")
        if roof and sus:
            roof["peak_sustained_random_operands"] = sus["tflops"]
            roof["sustained_source"] = {"measured_in_run": True, "tool": "tools/mfma_peak.hip sustained (fp16 32x32x16 MFMA loop, fresh random "
                                        "operands, 2 waves per SIMD on every CU, warm clocks)", "shader_clock_mhz": sus["shader_clock_mhz"]}
            roof["frac_of_sustained"] = round(roof["achieved"] * roof["mfma_passes_per_product"] / sus["tflops"], 4)
    if sus:
        log(f"sustained MFMA rate on this box: {sus['tflops']:.0f} TFLOP/s at {sus['shader_clock_mhz']:.0f} MHz")

    head = results[args.precision]
    out = {
        "metric": "sampled trajectories/sec (N x K, 50 denoise steps)",
        "value": head["value"], "unit": "traj/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": scaling,
        "ranks_seen": head["ranks_seen"], "per_rank_ms_per_step": head["per_rank_ms_per_step"], "gather_ms": head["gather_ms"],
        "rank0_affinity": affinity,
        "vs_baseline": None, "dtype": head["dtype"], "data": "synthetic",
        "config": {"workload": (f"{args.workload}: {E} episodes/GPU" if scaling == "weak" else
                                f"cfg5-style strong scaling: {total_eps} episodes in total, block-partitioned over {world} GPU(s) "
                                f"({E} on rank 0)") + f" x N={N} x K={K} x H={H}, {steps50} DDIM steps, "
                               f"{args.net.upper()} (encoder_dim 256, 3 layers), random-init weights",
                   "workload_short": (f"{args.workload}: {E} episodes/GPU" if scaling == "weak" else
                                      f"strong: {total_eps} episodes over {world} GPU(s)") +
                                     f" x N={N} x K={K} x H={H}, {steps50} DDIM steps, {args.net.upper()}",
                   "episodes_per_gpu": E, "total_episodes": total_eps, "humans": N, "samples": K, "horizon": H, "denoise_steps": steps50,
                   "net": args.net, "precision": args.precision, "lanes": args.lanes, "scenes": args.scenes,
                   "dist_backend": args.dist_backend if use_dist else None},
    }
    for k in ("roofline", "kernels", "hbm", "sweep_metrics"):
        if k in head:
            out[k] = head[k]
    out["host_feed"] = {"ms": round(host_feed_ms, 1), "episodes": E,
                        "what": ("the batch of this rank made resident once, before any timed step: scene histories -> standardised node states, "
                                 "neighbour sums and edge masks on the host (scene.py), one torch.randn per episode for x_T from its own "
                                 "generator (the reference's RNG contract), uploads; for the ORCA scenes also the crowd simulation"),
                        "note": "not part of `value`: the timed steps run on inputs resident in HBM"}
    out["kernels_note"] = ("per-class HIP-event times of ONE untimed pass over the same batch with one chunk in flight "
                           "(one-lane chunk plan: 51-52 episodes per launch; the timed region may run smaller chunks, two at a time)")
    out["sweep_metrics_note"] = "random-init weights: displacement vs the constant-velocity future is not meaningful"
    out["modes"] = results
    out["modes_note"] = ("every mode: same batch, same warm-up, same number of timed steps between the same barriers, same "
                         "parity sample; `value` and the top-level keys are modes[config.precision]")
    if len(modes) > 1:
        out["mean_ADE_between_modes_m"] = {f"{a}_vs_{b}": float(np.linalg.norm(
            (last_pos[a] - last_pos[b]).cpu().numpy(), axis=-1).mean()) for i, a in enumerate(modes) for b in modes[i + 1:]}

    # ---- single scene (BASELINE configs[1]) latency in the same run, every mode.  `ms_per_call` is the call the product issues per MPC
    # step - jmid_predict: host buffers in, encoder -> 50-step loop -> integrator chained on the handle's stream, all K samples back,
    # ONE synchronisation (PCIe both ways included: 9.6 KB per trajectory set).  `staged_ms_per_call`: the same work as two entries on
    # device tensors (jmid_encode + jmid_denoise, each ordered against the caller's stream, a range-flag round trip per call) - what
    # rounds 2-5 quoted here.
    if world == 1:
        ctx1 = eng.encode(x_st[:A], nbr[:A], emask[:A]).view(1, A, -1)
        hx, hn, he = (t[:A].cpu().numpy() for t in (x_st, nbr, emask))
        hxT, hp0 = x_T[:1].cpu().numpy(), p0[:1].cpu().numpy()
        ss = {}
        for m in modes:
            for _ in range(3):
                eng.denoise(x_T[:1], ctx1, p0[:1], dt=0.25, precision=m, want_vel=False)
            eng.synchronize()
            t1 = time.perf_counter()
            reps = 10
            for _ in range(reps):
                c1 = eng.encode(x_st[:A], nbr[:A], emask[:A]).view(1, A, -1)
                eng.denoise(x_T[:1], c1, p0[:1], dt=0.25, precision=m, want_vel=False)
            eng.synchronize()
            dt_staged = (time.perf_counter() - t1) / reps
            ss[m] = {"staged_ms_per_call": round(1e3 * dt_staged, 3)}
            if joint:
                for _ in range(3):
                    pos1, _ = eng.predict(hx, hn, he, hxT, hp0, K, dt=0.25, precision=m)
                t1 = time.perf_counter()
                reps = 20
                for _ in range(reps):
                    pos1, _ = eng.predict(hx, hn, he, hxT, hp0, K, dt=0.25, precision=m)
                dt1 = (time.perf_counter() - t1) / reps
                ss[m]["entry"] = "jmid_predict"
            else:
                dt1 = dt_staged
                ss[m]["entry"] = "jmid_encode + jmid_denoise"
            ss[m].update({"ms_per_call": round(1e3 * dt1, 3), "traj_per_s": round(A * K / dt1, 1)})
        out["single_scene"] = dict(ss[args.precision], workload=f"cfg2: 1 scene x N={N} x K={K} x H={H}, {steps50} steps",
                                   modes=ss)
    log("single-scene done")
    if world == 1 and not args.no_e2e and joint:
        out["forecaster_e2e"] = forecaster_e2e(weights, dev_id, modes)
        log("forecaster end-to-end done")

    # ---- CPU baseline (the oracle, a port of the reference; bounded sample) + parity on the same episodes
    if world == 1 and args.cpu_episodes != 0:
        from oracle import jmid_oracle as O
        ne = min(args.cpu_episodes if args.cpu_episodes > 0 else 12, E)
        # the parity sample is spread over the whole batch, so every chunk of the call (incl. a ragged last one) is
        # held against the oracle, not just the first
        wt = weights.tensors
        # every host core, the way the oracle uses them best: intra-op threads do not scale on these small GEMMs (a 2-step
        # calibration picks the per-process thread count), so the sample runs as avail // threads worker PROCESSES side by side
        avail = len(os.sched_getaffinity(0))
        best = (float("inf"), torch.get_num_threads())
        with torch.no_grad():
            ctx_cal = O.encode_context(wt, x_st[:A].cpu(), nbr[:A].cpu(), emask[:A].cpu())
            for nt in (4, 8, 16, 32):
                if nt > avail:
                    break
                torch.set_num_threads(nt)
                O.denoise(wt, ctx_cal, x_T_host[0], sample=K, step=1, joint=joint)
                tcal = time.perf_counter()
                O.denoise(wt, ctx_cal, x_T_host[0], sample=K, step=2, joint=joint)
                tcal = time.perf_counter() - tcal
                log(f"cpu calibration: {nt} threads -> {tcal:.3f}s / 2 steps ({tcal * nt:.2f} thread-s)")
                best = min(best, (tcal * nt, nt))     # what counts is thread-seconds per episode: the workers run side by side
        threads = best[1]
        torch.set_num_threads(1)           # the parent only waits from here on
        nproc_max = max(1, avail // threads) if args.cpu_all_cores else 1
        xs_c, nb_c, em_c, p0_c = x_st.cpu().numpy(), nbr.cpu().numpy(), emask.cpu().numpy(), p0.cpu().numpy()

        def ep_job(e):
            return (e, xs_c[e * A:(e + 1) * A], nb_c[e * A:(e + 1) * A], em_c[e * A:(e + 1) * A], x_T_host[e].numpy(), p0_c[e])

        import multiprocessing as mp
        mpc = mp.get_context("spawn")
        ready = mpc.Value("i", 0)
        env_keep = {k: os.environ.get(k) for k in ("OMP_NUM_THREADS", "MKL_NUM_THREADS", "OMP_WAIT_POLICY")}
        os.environ.update(OMP_NUM_THREADS=str(threads), MKL_NUM_THREADS=str(threads), OMP_WAIT_POLICY="passive")
        try:
            pool = mpc.Pool(nproc_max, initializer=cpu_init, initargs=(ready, args.seed, threads, joint, K, [ep_job(0)]))
        finally:
            for k, v in env_keep.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        with pool:
            t_wait = time.perf_counter()
            while ready.value < nproc_max:                   # every worker warm before any clock starts
                time.sleep(0.05)
                if time.perf_counter() - t_wait > 600:       # (a worker whose start-up raises is respawned for ever: do not wait for it)
                    raise RuntimeError(f"cpu_baseline: only {ready.value} of {nproc_max} worker processes came up in 600 s")
            # how many workers side by side the host really carries (the affinity mask says `avail`, a CPU quota or shared
            # memory bandwidth may say less - and oversubscribed OpenMP teams collapse): a 2-step job on n workers at once,
            # n = max, max/2, ...; the sample then runs on the n with the best aggregate rate
            scaling, n = {}, nproc_max
            short = (args.seed, threads, joint, K, 2, [ep_job(0)])
            while n >= 1:
                tq = time.perf_counter()
                pool.map(cpu_worker, [short] * n, chunksize=1)
                scaling[n] = round(n / (time.perf_counter() - tq), 2)
                n //= 2
            nproc = max(scaling, key=scaling.get)
            log(f"cpu worker scaling (2-step jobs per second by workers side by side): {scaling} -> {nproc} workers")
            if args.cpu_episodes < 0 and args.cpu_all_cores:     # automatic sample: at least one episode per worker process
                ne = min(E, max(ne, nproc))
            pick = sorted(set(int(round(v)) for v in np.linspace(0, E - 1, ne)))
            ne = len(pick)
            nproc = min(nproc, ne)
            jobs = [[] for _ in range(nproc)]
            for i, e in enumerate(pick):
                jobs[i % nproc].append(ep_job(e))
            jobs = [(args.seed, threads, joint, K, steps50, j) for j in jobs]
            log(f"cpu baseline: {nproc} worker processes x {threads} threads = {nproc * threads} of {avail} hardware threads "
                f"(os.cpu_count={os.cpu_count()}), {ne} episodes")
            tc = time.perf_counter()
            parts = pool.map(cpu_worker, jobs, chunksize=1)
            cpu_s = time.perf_counter() - tc
        got = dict(p for part in parts for p in part)
        log(f"cpu baseline: {ne} episodes in {cpu_s:.1f}s on {nproc} processes")
        pos_ref = np.stack([got[e] for e in pick])
        out["cpu_baseline"] = {"value": round(ne * A * K / cpu_s, 2), "unit": "traj/s", "cores": nproc * threads,
                               "kind": "port", "processes": nproc, "threads_per_process": threads,
                               "hardware_threads": avail, "host_physical_cores": physical_cores(),
                               "cores_note": "cores = processes x threads_per_process actually used by the timed sample (the best "
                                             "of the measured splits); host_physical_cores / hardware_threads = what the host has",
                               "worker_scaling_jobs_per_s": {str(k): v for k, v in scaling.items()},
                               "sample_short": f"{ne} episodes ({ne * A * K} trajectories) of the same workload, {cpu_s:.1f} s wall",
                               "cores_short": (f"oracle/jmid_oracle.py (torch-CPU fp32): best measured split {nproc} processes x {threads} "
                                               f"threads; more workers side by side did not raise the rate on this host "
                                               f"(tried {sorted(scaling)}: {nproc} was fastest)"),
                               "sample": f"{ne} episodes of the same workload ({ne * A * K} trajectories, {cpu_s:.1f} s wall; "
                                         f"oracle/jmid_oracle.py, torch-CPU fp32, {nproc} processes x {threads} threads)"}
        for m in modes:
            ade = float(np.linalg.norm(last_pos[m][pick].cpu().numpy() - pos_ref, axis=-1).mean())
            results[m]["parity"] = {"mean_ADE_vs_oracle_m": ade, "gate_m": 1e-4, "episodes": ne, "pass": ade <= 1e-4,
                                    "episode_ids": pick, "precision": m}
        out["parity"] = results[args.precision]["parity"]
    with open(args.detail, "w") as fh:
        json.dump(out, fh, indent=1)
    log(f"full result -> {args.detail} ({os.path.getsize(args.detail)} bytes)")
    for m in modes:
        rr = results[m].get("roofline") or {}
        log(f"[{m}] {results[m]['value']} traj/s, {results[m]['ms_per_step']} ms/step, {rr.get('kernel')} {rr.get('avg_launch_ms')} ms/launch "
            f"= {rr.get('frac')} of peak, parity {results[m].get('parity', {}).get('mean_ADE_vs_oracle_m')}")
    print(compact_line(out, args.detail), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
