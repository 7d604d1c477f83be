"""Copy the outputs of tools/round_profile.sh (gpurun_out/final/) and tools/mfma_busy.sh (gpurun_out/busy/) into
profiles/ under the round's prefix and refresh the per-mode PMC summaries.
   python tools/update_profiles.py [r01] [f16x2|f16x3]      (mode of the kernel-stats / PMC passes just run)"""
import json, os, shutil, sys
pref = sys.argv[1] if len(sys.argv) > 1 else "r01"
mode = sys.argv[2] if len(sys.argv) > 2 else "f16x2"
sfx = "" if mode == "f16x3" else f"_{mode}"        # the f16x3 files keep their round-1 names
raw = json.load(open(f"gpurun_out/final/pmc_raw_{mode}.json"))
p = f"profiles/{pref}_pmc_traffic{sfx}.json"
j = json.load(open(p if os.path.exists(p) else f"profiles/{pref}_pmc_traffic.json"))


def get(tag, c, sub):
    for k, v in raw.items():
        if k.startswith(f"{tag}:{c}:") and sub in k:
            return v["avg"]
    raise KeyError((tag, c, sub))


af, aw = get("attn", "FETCH_SIZE", "attn_f16x3_dma"), get("attn", "WRITE_SIZE", "attn_f16x3_dma")
gf, gw = get("gemm", "FETCH_SIZE", "dma256"), get("gemm", "WRITE_SIZE", "dma256")
j["precision"] = mode
j["attention"].update({"FETCH_SIZE_KB": round(af, 2), "WRITE_SIZE_KB": round(aw, 2), "hbm_bytes_per_launch": int((2 * af + aw) * 1024)})
j["gemm_qkv"].update({"FETCH_SIZE_KB": round(gf, 2), "WRITE_SIZE_KB": round(gw, 2), "hbm_bytes_per_launch": int((2 * gf + gw) * 1024)})
if mode == "f16x2":      # no V^T lo plane read, no O lo plane written
    j["attention"]["algorithmic_bytes_per_launch"] = 61200 * 512 * 2 * (5 + 1)
    j["attention"]["note"] = "algorithmic = Q, K hi/lo planes + V^T hi plane read once (61200 x 512 x 2 B x 5) + O hi plane written once"
    j["gemm_qkv"]["algorithmic_bytes_per_launch"] = 61440 * 512 * 2 + 1536 * 512 * 4 + 61440 * 1536 * 4
    j["gemm_qkv"]["note"] = "algorithmic = A hi plane 61440x512x2 B + W planes 1536x512x4 B read once + C fp32 written once (diagnostics epilogue)"
cf, cw = raw["call_total:FETCH_SIZE"], raw["call_total:WRITE_SIZE"]
j["call"].update({"FETCH_SIZE_KB_total": round(cf, 1), "WRITE_SIZE_KB_total": round(cw, 1),
                  "hbm_bytes_per_call": int((2 * cf + cw) * 1024), "hbm_bytes_per_trajectory": int((2 * cf + cw) * 1024 / 5100)})
json.dump(j, open(p, "w"), indent=1)
shutil.copy(f"gpurun_out/final/pmc_raw_{mode}.json", f"profiles/{pref}_pmc_raw{sfx}.json")
if os.path.exists(f"gpurun_out/busy/mfma_busy_{mode}.json"):
    shutil.copy(f"gpurun_out/busy/mfma_busy_{mode}.json", f"profiles/{pref}_mfma_busy{sfx}.json")
logs = ["cfg3_f16x2", "cfg3_f16x3", "cfg3_f32", "cfg2", "cfg4", "cfg5_1gpu", "cfg3_imid"]
for w in logs:
    if os.path.exists(f"gpurun_out/final/bench_{w}.log"):
        shutil.copy(f"gpurun_out/final/bench_{w}.log", f"profiles/{pref}_bench_{w}.log")
hdr = (f"# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --precision {mode} --lanes 1 --steps 1 --warmup 0 "
       f"--cpu-episodes 0 --episodes-per-gpu 51   (final kernels of the round, {mode.upper()} path; durations in ns; the run holds one "
       "untimed profiling step, one timed step, the single-scene calls and one warm + one timed pass of the other mode)\n")
if os.path.exists(f"gpurun_out/final/{mode}_kernel_stats.csv"):
    open(f"profiles/{pref}_{mode}_kernel_stats.csv", "w").write(hdr + open(f"gpurun_out/final/{mode}_kernel_stats.csv").read())
print("HBM bytes per trajectory:", j["call"]["hbm_bytes_per_trajectory"])
for w in logs:
    f = f"profiles/{pref}_bench_{w}.log"
    if not os.path.exists(f):
        continue
    b = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{w:12s} {b['config']['precision']:6s} {b['value']:10.1f} traj/s  {b['ms_per_step']:9.2f} ms/step  single scene {b.get('single_scene', {}).get('ms_per_call')}"
          f"  cpu {b.get('cpu_baseline', {}).get('value')}  parity {b.get('parity', {}).get('mean_ADE_vs_oracle_m')}  other {b.get('other_modes')}")
b = json.loads(open(f"profiles/{pref}_bench_cfg3_{mode}.log").read().strip().splitlines()[-1])
print(json.dumps(b["roofline"])[:600]); print(b.get("hbm")); print(b["kernels"])
