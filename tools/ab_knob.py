"""A/B of one tuning knob on a 51-episode 50-step call: time per call and bitwise equality of the results.
python tools/ab_knob.py precision knob valueA valueB [episodes] [other=knob ...]     (JMID_JOINT=0: the iMID net)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
prec, knob, va, vb = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
E = int(sys.argv[5]) if len(sys.argv) > 5 else 51
A, K, T = 5, 20, 12
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 23), joint=os.environ.get("JMID_JOINT", "1") == "1", step=50)
eng.set_tuning("lanes", 1)
for kv in sys.argv[6:]:
    k, v = kv.split("=")
    eng.set_tuning(k, int(v))
g = torch.Generator().manual_seed(3)
ctx = torch.randn([E, A, 256], generator=g).cuda()
x_T = torch.randn([E, K * A, T, 2], generator=g).cuda()
outs = {}
for rep in range(3):
    for val in (va, vb):
        eng.set_tuning(knob, val)
        v = eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)[0]
        eng.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            v = eng.denoise(x_T, ctx, None, precision=prec, want_pos=False)[0]
        eng.synchronize(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 4 * 1e3
        outs[val] = v.clone()
        print(f"[{prec}] {knob}={val}: {ms:.2f} ms / call", flush=True)
print("bitwise equal:", bool(torch.equal(outs[va], outs[vb])), " max diff", float((outs[va] - outs[vb]).abs().max()))
