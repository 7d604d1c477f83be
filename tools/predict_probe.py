import sys, time, numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
w = JMIDWeights.from_seed(NetDims(ctx_dim=256), 0)
eng = JmidEngine(w, joint=True, step=50)
A, K, T = 5, 20, 12
g = torch.Generator().manual_seed(1)
hx = torch.randn([A, 6, 6], generator=g).numpy(); hn = torch.randn([A, 2, 6, 6], generator=g).numpy(); he = torch.rand([A, 2], generator=g).numpy()
hxT = torch.randn([1, K * A, T, 2], generator=g).numpy(); hp0 = torch.randn([1, A, 2], generator=g).numpy()
def tp(tag, n=20):
    for _ in range(3): eng.predict(hx, hn, he, hxT, hp0, K, dt=0.25, precision="f16mx")
    t = time.perf_counter()
    for _ in range(n): eng.predict(hx, hn, he, hxT, hp0, K, dt=0.25, precision="f16mx")
    print(tag, "predict ms/call", round(1e3 * (time.perf_counter() - t) / n, 3), flush=True)
def ts(tag, n=20):
    dx, dn, de, dT, dp = (torch.from_numpy(a).cuda() for a in (hx, hn, he, hxT, hp0))
    for _ in range(3):
        c = eng.encode(dx, dn, de).view(1, A, -1); eng.denoise(dT, c, dp, dt=0.25, precision="f16mx", want_vel=False)
    eng.synchronize(); t = time.perf_counter()
    for _ in range(n):
        c = eng.encode(dx, dn, de).view(1, A, -1); eng.denoise(dT, c, dp, dt=0.25, precision="f16mx", want_vel=False)
    eng.synchronize()
    print(tag, "staged ms/call", round(1e3 * (time.perf_counter() - t) / n, 3), flush=True)
tp("fresh"); ts("fresh"); tp("after staged")
E = 64
ctx = torch.randn([E, A, 256], generator=g).cuda(); xT = torch.randn([E, K * A, T, 2], generator=g).cuda(); p0 = torch.randn([E, A, 2], generator=g).cuda()
eng.denoise(xT, ctx, p0, dt=0.25, precision="f16mx", want_vel=False); eng.synchronize()
tp("after a 64-episode call"); ts("after a 64-episode call")
eng.set_tuning("lanes", 1); tp("lanes=1"); eng.set_tuning("lanes", 2)
print("timeouts", eng.timeout_count(), "erange", eng.erange_count())
