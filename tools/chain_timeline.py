"""Where the wall time of a one-scene call goes, on the GPU's own clock: rocprofv3 --kernel-trace of a few calls, then for the last
call the kernels in launch order with their duration (end - start) and the gap to the next kernel's start.
    python tools/chain_timeline.py [f16mx|f16x3] [episodes]            (SS_SHAPE=A,K,T,steps as tools/single_scene_sweep.py)"""
import csv, glob, os, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--inner":
    import torch
    sys.path.insert(0, ROOT)
    from safe_interactive_crowdnav_amd.engine import JmidEngine
    from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
    prec, E = sys.argv[2], int(sys.argv[3])
    A, K, T, STEP = (int(v) for v in os.environ.get("SS_SHAPE", "5,20,12,50").split(","))
    eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 0), joint=True, step=STEP)
    g = torch.Generator().manual_seed(3)
    ctx = torch.randn([E, A, 256], generator=g).cuda()
    x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
    p0 = torch.randn([E, A, 2], generator=g).cuda()
    for _ in range(12):
        eng.denoise(x_T, ctx, p0, precision=prec, want_vel=False)
    eng.synchronize()
    sys.exit(0)
prec = sys.argv[1] if len(sys.argv) > 1 else "f16mx"
E = sys.argv[2] if len(sys.argv) > 2 else "1"
out = "/tmp/chain_timeline"
subprocess.run(["rm", "-rf", out])
env = dict(os.environ, TMPDIR="/tmp")
subprocess.run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable, os.path.abspath(__file__), "--inner", prec, E],
               check=True, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
f = max(glob.glob(out + "/**/*kernel_trace.csv", recursive=True), key=os.path.getsize)
rows = sorted(({"name": r["Kernel_Name"], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])} for r in csv.DictReader(open(f))), key=lambda r: r["s"])
# the last call = everything after the last encoder-free stretch: take the last N kernels, N = kernels per call
names = [r["name"] for r in rows]
idx = [i for i, n in enumerate(names) if "out_ddim_kernel<false>" in n or "out_ddim_traj_kernel<false>" in n]     # the last step of every call
first, last = idx[-2] + 1, idx[-1]
call = rows[first:last + 1]
wall = (call[-1]["e"] - call[0]["s"]) / 1e3
agg = collections.OrderedDict()
for a, b in zip(call, call[1:] + [None]):
    k = a["name"].split("(")[0].replace("void jmid::", "").replace("jmid::", "")[:70]
    d = agg.setdefault(k, [0, 0.0, 0.0])
    d[0] += 1
    d[1] += (a["e"] - a["s"]) / 1e3
    d[2] += ((b["s"] - a["e"]) / 1e3) if b else 0.0
busy = sum(d[1] for d in agg.values())
gaps = sum(d[2] for d in agg.values())
print(f"{prec} E={E}: last call {len(call)} kernels, {wall:.1f} us from first start to last end: {busy:.1f} us inside kernels, {gaps:.1f} us between them ({gaps / len(call):.2f} us per boundary)")
print(f"{'kernel':70s} {'launches':>8s} {'avg us':>8s} {'gap after':>9s} {'total us':>9s}")
for k, (n, dur, gap) in sorted(agg.items(), key=lambda kv: -kv[1][1] - kv[1][2]):
    print(f"{k:70s} {n:8d} {dur / n:8.2f} {gap / n:9.2f} {dur + gap:9.1f}")
