"""Copy the outputs of tools/round_profile.sh (gpurun_out/final/) into profiles/ under the round's prefix and refresh
profiles/<prefix>_pmc_traffic.json.   python tools/update_profiles.py [r01]"""
import json, shutil, sys
pref = sys.argv[1] if len(sys.argv) > 1 else "r01"
raw = json.load(open("gpurun_out/final/pmc_raw.json"))
p = f"profiles/{pref}_pmc_traffic.json"
j = json.load(open(p))


def get(tag, c, sub):
    for k, v in raw.items():
        if k.startswith(f"{tag}:{c}:") and sub in k:
            return v["avg"]
    raise KeyError((tag, c, sub))


af, aw = get("attn", "FETCH_SIZE", "attn_f16x3_dma"), get("attn", "WRITE_SIZE", "attn_f16x3_dma")
gf, gw = get("gemm", "FETCH_SIZE", "dma256"), get("gemm", "WRITE_SIZE", "dma256")
j["attention"].update({"FETCH_SIZE_KB": round(af, 2), "WRITE_SIZE_KB": round(aw, 2), "hbm_bytes_per_launch": int((2 * af + aw) * 1024)})
j["gemm_qkv"].update({"FETCH_SIZE_KB": round(gf, 2), "WRITE_SIZE_KB": round(gw, 2), "hbm_bytes_per_launch": int((2 * gf + gw) * 1024)})
cf, cw = raw["call_total:FETCH_SIZE"], raw["call_total:WRITE_SIZE"]
j["call"].update({"FETCH_SIZE_KB_total": round(cf, 1), "WRITE_SIZE_KB_total": round(cw, 1),
                  "hbm_bytes_per_call": int((2 * cf + cw) * 1024), "hbm_bytes_per_trajectory": int((2 * cf + cw) * 1024 / 5100)})
json.dump(j, open(p, "w"), indent=1)
for a, b in [("bench_cfg3_f16x3.log", "bench_cfg3_f16x3.log"), ("bench_cfg3_f32.log", "bench_cfg3_f32.log"),
             ("bench_cfg2.log", "bench_cfg2.log"), ("bench_cfg4.log", "bench_cfg4.log"), ("bench_cfg5_1gpu.log", "bench_cfg5_1gpu.log"),
             ("bench_cfg3_imid.log", "bench_cfg3_imid.log"), ("pmc_raw.json", "pmc_raw.json")]:
    shutil.copy("gpurun_out/final/" + a, f"profiles/{pref}_{b}")
hdr = ("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --lanes 1 --steps 1 --warmup 0 --cpu-episodes 0 "
       "--episodes-per-gpu 51   (final kernels of the round, F16X3 path; durations in ns; the run holds one untimed "
       "profiling step, one timed step and the single-scene calls bench.py makes afterwards)\n")
open(f"profiles/{pref}_f16x3_kernel_stats.csv", "w").write(hdr + open("gpurun_out/final/f16x3_kernel_stats.csv").read())
print("HBM bytes per trajectory:", j["call"]["hbm_bytes_per_trajectory"])
for w in ("cfg3_f16x3", "cfg3_f32", "cfg2", "cfg4", "cfg5_1gpu", "cfg3_imid"):
    b = json.loads(open(f"profiles/{pref}_bench_{w}.log").read().strip().splitlines()[-1])
    print(f"{w:12s} {b['value']:10.1f} traj/s  {b['ms_per_step']:9.2f} ms/step  single scene {b.get('single_scene', {}).get('ms_per_call')}"
          f"  cpu {b.get('cpu_baseline', {}).get('value')}  parity {b.get('parity', {}).get('mean_ADE_vs_oracle_m')}")
b = json.loads(open(f"profiles/{pref}_bench_cfg3_f16x3.log").read().strip().splitlines()[-1])
print(json.dumps(b["roofline"])[:400]); print(b.get("hbm")); print(b["kernels"])
