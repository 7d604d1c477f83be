"""Prints the figures docs/NOTEBOOK.md section 4b / README / profiles/README quote, from profiles/<round>_* and gpurun_out/<round>/.
python tools/doc_numbers.py [r02]"""
import csv, json, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r02"
j = json.load(open(f"profiles/{R}_bench_cfg3.json"))
for m in ("f16mx", "f16x2", "f16x3"):
    v = j["modes"][m]; r = v["roofline"]; k = v["kernels"]
    print(m, v["value"], v["ms_per_step"], "%.2e" % v["parity"]["mean_ADE_vs_oracle_m"], "attn", r["avg_launch_ms"], r["achieved"], r["frac"],
          r["frac_of_split_peak"], "timed", r["timed_region"]["avg_launch_ms"], "path", r["path_achieved"], r["path_frac"], "busy",
          r["mfma_busy"]["dominant_kernel"], r["mfma_busy"]["whole_call"])
    print("  ", {a: (b["avg_ms"], b["tflops"]) for a, b in k.items()})
    s = 3 * sum(k[c]["avg_ms"] for c in ("gemm_qkv", "gemm_attn_out", "gemm_ff1", "gemm_ff2", "attention")) + 2 * k["gemm_tail"]["avg_ms"]
    print("   sum/step", round(s, 3), "attention share", round(3 * k["attention"]["avg_ms"] / s, 3))
print(j["mean_ADE_between_modes_m"], "cpu", j["cpu_baseline"]["value"])
print("hbm", j["hbm"]["bytes_per_trajectory"], j["hbm"]["GBps"], j["hbm"]["frac"])
for m in ("f16mx", "f16x2", "f16x3"):
    p = json.load(open(f"profiles/{R}_pmc_call_{m}.json"))
    ks = p["kernels"]
    it = ks.items() if isinstance(ks, dict) else [(k["name"], k) for k in ks]
    print(m, "whole-call busy", p["whole_call_mfma_busy"], "MB/traj", round(p["call"]["hbm_bytes_per_trajectory"] / 1e6, 1),
          "TB/s at bench rate", round(p["call"]["hbm_bytes_per_trajectory"] * j["modes"][m]["value"] / 1e12, 2))
    for n, k in it:
        if any(t in n for t in ("attn_f16x3_dma", "<0, 2, 4, 2, 2, 4", "256x256_kernel<0, 2", "<1, 1, 4, 2, 2, 4", "256x256_kernel<1, 1", "gemm_ln")):
            print("    ", n[11:60], round(k["hbm_bytes_per_launch"] / 1e6, 1), round(k["mfma_busy"], 4))
for m in ("f16mx", "f16x2", "f16x3"):
    try:
        for l in open(f"gpurun_out/{R}/prof_bench_{m}.log"):
            if l.startswith("{"):
                print(m, "profiled run, attention avg", json.loads(l)["roofline"]["avg_launch_ms"])
    except OSError:
        pass
    for row in csv.reader(open(f"profiles/{R}_{m}_kernel_stats.csv")):
        if row and "attn_f16x3_dma" in row[0]:
            print("    csv", row[1], row[2])
for row in csv.reader(open(f"profiles/{R}_cfg2_f16mx_kernel_stats.csv")):
    if row and row[0][0] != "#" and row[0] != "Name" and int(row[1]) > 1000:
        print(row[0][:70], row[1], row[3])
for f in ("cfg2", "cfg4", "cfg5_1gpu", "cfg3_imid", "cfg3_orca", "cfg3_f32"):
    b = json.load(open(f"profiles/{R}_bench_{f}.json"))
    print(f, {m: (v["value"], v["ms_per_step"], v.get("parity", {}).get("mean_ADE_vs_oracle_m")) for m, v in b["modes"].items()},
          b.get("single_scene", {}).get("modes"))
print(open(f"profiles/{R}_lanes.log").read())
print(open(f"profiles/{R}_episode_sweep.log").read())
