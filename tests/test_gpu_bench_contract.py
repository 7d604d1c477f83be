"""bench.py prints ONE JSON line with the fields the driver and the judge read (contract in the task statement)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODE_KEYS = ("value", "unit", "steps", "warmup", "ms_per_step", "dtype", "roofline", "kernels", "sweep_metrics")


# bench.py is the product's measurement: it runs on the PRODUCTION library, not on the diagnostics flavour the tests load
PROD_ENV = {k: v for k, v in os.environ.items() if k != "JMID_LIB"}


COMPACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "modes", "detail")


def _one_json_line(out, full=True):
    """The stdout contract: exactly ONE line, the LAST one, strict JSON, short enough for the driver's tail buffer (round 4's
    20.7 KB line was not parsed).  Returns the long result from the file the line names (`full`), or the line itself."""
    assert out.returncode == 0, out.stderr[-3000:]
    lines = out.stdout.splitlines()
    assert lines and lines[-1].startswith("{"), out.stdout[-2000:]
    assert len([l for l in lines if l.startswith("{")]) == 1, out.stdout[-2000:]
    assert len(lines[-1]) < 6000, len(lines[-1])
    line = json.loads(lines[-1], parse_constant=lambda c: pytest.fail(f"non-strict JSON constant {c}"))
    for k in COMPACT_KEYS:
        assert k in line, k
    detail = line["detail"] if os.path.isabs(line["detail"]) else os.path.join(REPO, line["detail"])
    long = json.load(open(detail))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "scaling"):
        assert long[k] == line[k], k                        # the line is a selection of the long result, nothing else
    return long if full else line


def test_bench_json_contract(tmp_path):
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
                          "--episodes-per-gpu", "6", "--chunk", "2", "--cpu-episodes", "3", "--detail", str(tmp_path / "d.json")],
                         capture_output=True, text=True, timeout=600, cwd=REPO, env=PROD_ENV)
    c = _one_json_line(out, full=False)
    assert len(out.stderr) < 8000, len(out.stderr)         # the log stays short too: the driver's tail holds stdout + stderr
    # the compact line carries what the driver and the judge read: contract keys, roofline, cpu_baseline, parity, a row per mode
    from safe_interactive_crowdnav_amd import forecaster as FC
    assert c["config"]["precision"] == "f16mx" and c["config"]["class_default_precision"] == FC.DEFAULTS["precision"]
    assert "workload" in c["config"] and "model" not in c["config"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches", "flops_per_launch"):
        assert k in c["roofline"], k
    assert abs(c["roofline"]["frac"] - c["roofline"]["achieved"] / c["roofline"]["peak"]) < 1e-3
    for k in ("value", "unit", "cores", "kind", "sample", "host_physical_cores"):
        assert k in c["cpu_baseline"], k
    assert c["cpu_baseline"]["kind"] == "port" and c["cpu_baseline"]["cores"] >= 1 and c["cpu_baseline"]["value"] > 0
    assert c["parity"]["pass"] is True and c["parity"]["mean_ADE_vs_oracle_m"] <= 1e-4 and c["parity"]["precision"] == "f16mx"
    assert set(c["modes"]) == {"f16x3", "f16x2", "f16mx"}
    for m, v in c["modes"].items():
        assert v["value"] > 0 and v["ms_per_step"] > 0 and v["pass"] is True and v["mean_ADE_vs_oracle_m"] <= 1e-4
    assert c["value"] == c["modes"]["f16mx"]["value"] and set(c["single_scene"]["modes"]) == set(c["modes"])
    j = _one_json_line(out)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 1 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["scaling"] == "weak" and j["vs_baseline"] is None and j["data"] == "synthetic" and j["value"] > 0
    assert "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = j["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["sample"]
    assert c["processes"] * c["threads_per_process"] == c["cores"] <= c["hardware_threads"]
    # the headline is the mode the drop-in class runs by default (round 6: "f16mx" with the first-call self check against "f16x3") ...
    import inspect
    assert inspect.signature(FC.HumanTrajectoryForecasterSim.__init__).parameters["precision"].default is None     # -> DEFAULTS
    assert FC.DEFAULTS["precision"] == "f16mx" and FC.DEFAULTS["self_check"] is True
    assert _one_json_line(out, full=False)["config"]["class_default_precision"] == j["config"]["precision"] == "f16mx"
    assert j["config"]["precision"] == "f16mx" and j["parity"]["precision"] == "f16mx"
    assert j["parity"]["pass"] is True and j["parity"]["mean_ADE_vs_oracle_m"] <= 1e-4
    # ... and all split modes are measured the same way and reported under the same keys
    assert set(j["modes"]) == {"f16x3", "f16x2", "f16mx"}
    for m, v in j["modes"].items():
        for k in MODE_KEYS + ("parity",):
            assert k in v, (m, k)
        assert v["steps"] == 1 and v["warmup"] == 1 and v["value"] > 0
        assert v["parity"]["pass"] is True and v["parity"]["episodes"] == 3
        assert v["parity"]["episode_ids"] == [0, 2, 5]          # spread over the batch: every chunk of 2 is sampled
    assert j["value"] == j["modes"]["f16mx"]["value"] and j["ms_per_step"] == j["modes"]["f16mx"]["ms_per_step"]
    assert j["modes"]["f16x3"]["parity"]["mean_ADE_vs_oracle_m"] <= 1e-5       # fp32-class mode
    assert j["mean_ADE_between_modes_m"]["f16x2_vs_f16x3"] <= 1e-4 and j["mean_ADE_between_modes_m"]["f16mx_vs_f16x2"] <= 1e-4
    assert set(j["single_scene"]["modes"]) == {"f16x3", "f16x2", "f16mx"}
    # the class surface, host + device, the way the MPC calls it: cfg2 and the reference's shipped operating point
    fe = j["forecaster_e2e"]
    assert set(fe) == {"cfg2", "shipped", "shipped_batched"}
    for m, t in fe["shipped_batched"]["modes"].items():          # predict_batch(): 64 episodes per call
        assert t["ms_per_call"] > 0 and abs(t["ms_per_episode"] * 64 - t["ms_per_call"]) < 0.01 * t["ms_per_call"] + 0.01
    for name, v in fe.items():
        assert set(v["modes"]) == {"f16x3", "f16x2", "f16mx"}
        if name == "shipped_batched":
            continue
        for m, t in v["modes"].items():
            assert t["ms_per_call"] > 0 and t["erange_fallbacks"] == 0
            assert abs(t["host_scene_ms"] + t["device_ms"] + t["topk_ms"] + t["host_assemble_ms"] - t["ms_per_call"]) < 0.5 * t["ms_per_call"]
    assert "jmid_topk" in fe["shipped"]["topk"]
    # PMC-derived fields say where they come from: counters collected in this run (one 51-episode call re-run under
    # rocprofv3 --pmc after the timed region) or, failing that, the committed profile they were read from
    for k in ("traffic_source", "mfma_busy"):
        if k in r:
            src = r[k]["source"] if k == "mfma_busy" else r[k]
            assert src["measured_in_run"] is True or (src["file"].startswith("profiles/") and len(src["git_blob_sha1"]) == 40)
    import shutil
    if shutil.which("rocprofv3"):
        assert r["traffic_source"]["measured_in_run"] is True and r["traffic"] > 0, r.get("traffic_source")
        assert j["hbm"]["source"]["measured_in_run"] is True and 0 < j["hbm"]["frac"] < 1
    # one rank: the per-rank fields are there and trivial; the metric gather is timed outside the steps
    assert j["ranks_seen"] == 1 and j["per_rank_ms_per_step"]["min"] == j["per_rank_ms_per_step"]["max"] and j["gather_ms"] >= 0
    assert j["cpu_baseline"]["host_physical_cores"] is None or j["cpu_baseline"]["host_physical_cores"] >= 1


def test_bench_two_ranks_share_one_gpu_over_gloo():
    """The N > 1 path of bench.py, for real, on the one GPU of the test box: two ranks (both on device 0) launched the
    way the driver launches them, host collectives over gloo.  Exercises process-group init, per-rank seeds and episode
    shards, the barrier / MAX all-reduce bracketing of the timed region, gather_metrics of device tensors in episode
    order, and the rank != 0 exit.  (The RCCL flavour of the same calls needs two GPUs; docs/NOTEBOOK.md section 5.)"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(PROD_ENV, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"),
                          "--gpus", "2", "--steps", "2", "--warmup", "1", "--episodes-per-gpu", "5", "--chunk", "2",
                          "--dist-backend", "gloo", "--device", "0", "--modes", "f16x3", "--cpu-episodes", "0"],
                         capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    j = _one_json_line(out)
    assert j["n_gpus"] == 2 and j["steps"] == 2 and j["warmup"] == 1 and j["scaling"] == "weak"
    assert j["config"]["dist_backend"] == "gloo" and j["config"]["episodes_per_gpu"] == 5
    sm = j["sweep_metrics"]
    assert sm["episodes"] == 10                      # 5 episodes of rank 0 followed by 5 of rank 1
    assert sm["mean_ADE_m"] > 0 and sm["mean_ADE_m"] == sm["mean_ADE_m"]     # no NaN padding rows leaked through
    # whole-job rate: 2 ranks x 5 episodes x 5 humans x 20 samples per step
    assert abs(j["value"] - 2 * 5 * 5 * 20 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-3


def test_bench_strong_scaling_two_ranks_share_one_gpu_over_gloo():
    """`--gpus 2` defaults to STRONG scaling: a fixed total (here 7 episodes instead of configs[4]'s 4096) block-partitioned over
    the ranks (sweep.shard_range: 4 + 3), no collective inside the timed steps, ONE gather of the metric rows after them,
    per-rank step times and the ranks the process group saw in the JSON line.  Plain launch (bench.py spawns its ranks)."""
    env = {k: v for k, v in PROD_ENV.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                          "--total-episodes", "7", "--dist-backend", "gloo", "--device", "0", "--modes", "f16mx",
                          "--cpu-episodes", "0", "--no-profile"],
                         capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    j = _one_json_line(out)
    assert j["n_gpus"] == 2 and j["scaling"] == "strong" and j["ranks_seen"] == 2
    assert j["config"]["total_episodes"] == 7 and j["config"]["episodes_per_gpu"] == 4          # rank 0's block
    assert j["sweep_metrics"]["episodes"] == 7
    pr = j["per_rank_ms_per_step"]
    assert len(pr["all"]) == 2 and 0 < pr["min"] <= pr["max"] <= j["ms_per_step"] * 1.05 and j["gather_ms"] >= 0
    # whole-job rate: the FIXED total per step
    assert abs(j["value"] - 7 * 5 * 20 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-3


def test_bench_eight_ranks_share_one_gpu_strong_scaling_ragged_shards():
    """8-rank readiness without an 8-GPU node: the driver's own launch line for N = 8 (`torch.distributed.run --nproc-per-node 8`)
    with all eight ranks on the one GPU of the test box and host collectives over gloo: `--scaling strong` with a total that does NOT
    divide by eight (61 episodes = 5 ranks x 8 + 3 ranks x 7, sweep.shard_range), ranks_seen == 8, eight per-rank step times, the
    gathered rows in GLOBAL episode order (bench.py tags every row with its episode index before the gather), and every rank pinned to
    its own share of the host cores (bench.pin_rank)."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(PROD_ENV, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"),
                          "--gpus", "8", "--steps", "1", "--warmup", "1", "--scaling", "strong", "--total-episodes", "61",
                          "--dist-backend", "gloo", "--device", "0", "--modes", "f16mx", "--cpu-episodes", "0", "--no-profile"],
                         capture_output=True, text=True, timeout=1200, cwd=REPO, env=env)
    c = _one_json_line(out, full=False)
    assert c["n_gpus"] == 8 and c["ranks_seen"] == 8 and c["scaling"] == "strong"
    assert c["config"]["total_episodes"] == 61 and c["config"]["episodes_per_gpu"] == 8 and c["sweep_episodes"] == 61
    assert c["rows_in_episode_order"] is True
    j = _one_json_line(out)
    pr = j["per_rank_ms_per_step"]
    assert len(pr["all"]) == 8 and 0 < pr["min"] <= pr["max"] <= j["ms_per_step"] * 1.05
    assert abs(j["value"] - 61 * 5 * 20 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-3
    sm = j["sweep_metrics"]
    assert sm["episodes"] == 61 and sm["rows_in_episode_order"] is True and sm["mean_ADE_m"] == sm["mean_ADE_m"]
    aff = j["rank0_affinity"]
    if aff is not None and len(os.sched_getaffinity(0)) >= 8:            # rank 0 holds an eighth of the cores this process may use
        assert 1 <= aff["cores"] <= len(os.sched_getaffinity(0)) // 8 + 1


def test_bench_launched_plainly_spawns_its_own_ranks():
    """`python bench.py --gpus 2` without torch.distributed.run (how the driver launched the 1-GPU bench): bench.py starts the
    two ranks itself, rank 0 prints the one JSON line."""
    env = {k: v for k, v in PROD_ENV.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                          "--episodes-per-gpu", "3", "--dist-backend", "gloo", "--device", "0", "--modes", "f16mx",
                          "--cpu-episodes", "0", "--no-profile"],
                         capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    j = _one_json_line(out)
    assert j["n_gpus"] == 2 and j["sweep_metrics"]["episodes"] == 6 and j["value"] > 0


def test_bench_collectives_run_on_rccl_with_one_rank():
    """The RCCL flavour of the N > 1 path, as far as one GPU allows: a one-rank "nccl" process group (torchrun, the
    driver's launch line with --nproc-per-node 1) with --force-dist, so that init_process_group(device_id=...), the
    barriers, the MAX all-reduce of a device tensor and gather_metrics' dist.gather of device tensors all go through
    RCCL.  What is left for the driver's 8-GPU run is only that these same calls see more than one rank."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(PROD_ENV, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(REPO, "bench.py"),
                          "--gpus", "1", "--steps", "2", "--warmup", "1", "--episodes-per-gpu", "4", "--force-dist",
                          "--dist-backend", "nccl", "--modes", "f16x2", "--cpu-episodes", "0"],
                         capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    j = _one_json_line(out)
    assert j["n_gpus"] == 1 and j["config"]["dist_backend"] == "nccl" and j["sweep_metrics"]["episodes"] == 4
    assert j["value"] > 0


def test_bench_two_gpus_over_rccl_strong_scaling():
    """The first multi-GPU lease verifies itself: `bench.py --gpus 2 --dist-backend nccl --scaling strong` - one rank per GPU, the
    process group on RCCL, 64 episodes block-partitioned 32 + 32, ONE gather of the metric rows over xGMI after the timed steps
    (SURVEY 8e).  Skipped on a one-GPU box (where the gloo two-rank and the RCCL one-rank tests above cover the same calls)."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs: RCCL with one rank per GPU")
    env = {k: v for k, v in PROD_ENV.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dist-backend", "nccl", "--scaling", "strong",
                          "--total-episodes", "64", "--steps", "2", "--warmup", "1", "--modes", "f16mx", "--cpu-episodes", "0",
                          "--no-profile"],
                         capture_output=True, text=True, timeout=900, cwd=REPO, env=env)
    c = _one_json_line(out, full=False)
    assert c["n_gpus"] == 2 and c["ranks_seen"] == 2 and c["scaling"] == "strong" and c["config"]["dist_backend"] == "nccl"
    assert c["config"]["total_episodes"] == 64 and c["config"]["episodes_per_gpu"] == 32 and c["sweep_episodes"] == 64
    j = _one_json_line(out)
    pr = j["per_rank_ms_per_step"]
    assert len(pr["all"]) == 2 and 0 < pr["min"] <= pr["max"] <= j["ms_per_step"] * 1.05
    assert abs(j["value"] - 64 * 5 * 20 / (j["ms_per_step"] * 1e-3)) / j["value"] < 1e-3
    sm = j["sweep_metrics"]
    assert sm["episodes"] == 64 and sm["mean_ADE_m"] == sm["mean_ADE_m"]        # rows of both ranks arrived, in episode order
