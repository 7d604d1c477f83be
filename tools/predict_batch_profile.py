import sys, os, time, cProfile, pstats
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from safe_interactive_crowdnav_amd.forecaster import predict_batch
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims
rng = np.random.default_rng(5)
Eb, N, K, k, H, F = 64, 3, 100, 15, 8, 6
p0 = rng.uniform(-1.0, 1.0, (Eb, 1, N, 2)); v = rng.uniform(-0.4, 0.4, (Eb, 1, N, 2))
hum = p0 + v * 0.25 * np.arange(F)[None, :, None, None]
rob = np.stack([np.zeros(F), -1.5 + 0.05 * np.arange(F)], axis=-1)[None].repeat(Eb, axis=0)
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 0), joint=True, step=2)
kw = dict(num_samples=K, num_ret_samples=k, horizon=H, time_step=0.25, precision="f16mx")
for _ in range(3): predict_batch(eng, hum, rob, range(Eb), **kw)
pr = cProfile.Profile(); pr.enable()
for _ in range(10): predict_batch(eng, hum, rob, range(Eb), **kw)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
