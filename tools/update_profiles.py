"""Copy the outputs of tools/round_profile.sh (gpurun_out/<round>/) into profiles/ under the round's prefix.
   python tools/update_profiles.py r02"""
import json, os, shutil, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
src, dst = f"gpurun_out/{R}", "profiles"
names = ["bench_cfg3.json", "bench_cfg3_f32.json", "bench_cfg2.json", "bench_cfg4.json", "bench_cfg5_1gpu.json", "bench_cfg3_imid.json",
         "bench_cfg3_orca.json", "episode_sweep.log", "soak.log", "packed_fp32_probe.log", "scaled_mfma_probe.log", "lanes.log", "pmc_call_f16mx.json", "pmc_call_f16x2.json",
         "pmc_call_f16x3.json", "shipped_profile.log", "small_pmc_f16mx_E1.txt", "small_launch_traces.log", "bench_strong_2ranks_1gpu.json", "bench_cfg3.line.json", "robustness.json", "sq_counters_f16mx.json"]
for n in names:
    if os.path.exists(f"{src}/{n}"):
        shutil.copy(f"{src}/{n}", f"{dst}/{R}_{n}")
if os.path.exists(f"{src}/bench_cfg3.err"):
    shutil.copy(f"{src}/bench_cfg3.err", f"{dst}/{R}_bench_cfg3.stderr.log")
hdr = {"f16mx": "python bench.py --precision f16mx --modes f16mx --lanes 1 --steps 1 --warmup 0 --cpu-episodes 0 --episodes-per-gpu 51",
       "cfg2_f16mx": "python bench.py --workload cfg2 --modes f16mx --steps 20 --warmup 3 --cpu-episodes 0 --no-profile",
       "f16x2": "python bench.py --precision f16x2 --modes f16x2 --lanes 1 --steps 1 --warmup 0 --cpu-episodes 0 --episodes-per-gpu 51",
       "f16x3": "python bench.py --precision f16x3 --modes f16x3 --lanes 1 --steps 1 --warmup 0 --cpu-episodes 0 --episodes-per-gpu 51",
       "cfg2_f16x2": "python bench.py --workload cfg2 --modes f16x2 --steps 20 --warmup 3 --cpu-episodes 0 --no-profile"}
for tag, cmd in hdr.items():
    f = f"{src}/{tag}_kernel_stats.csv"
    if os.path.exists(f):
        open(f"{dst}/{R}_{tag}_kernel_stats.csv", "w").write(
            f"# rocprofv3 --kernel-trace --stats --output-format csv -- {cmd}   (durations in ns; the run holds one warm-up-free "
            "profiling step, one timed step and the single-scene calls of that mode)\n" + open(f).read())
for m in ("f16mx", "f16x2", "f16x3"):
    p = f"{dst}/{R}_pmc_call_{m}.json"
    if os.path.exists(p):
        j = json.load(open(p))
        print(m, "HBM bytes / trajectory", j["call"]["hbm_bytes_per_trajectory"], "whole-call MFMA busy", j["whole_call_mfma_busy"])
b = json.load(open(f"{dst}/{R}_bench_cfg3.json"))
print({m: (v["value"], v["roofline"]["kernel"], v["roofline"]["frac"], v["roofline"]["avg_launch_ms"]) for m, v in b["modes"].items()})
