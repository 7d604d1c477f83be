"""End-to-end latency of HumanTrajectoryForecasterSim.predict_ret_best() (what one MPC step pays), split into
host preprocessing / encoder / denoise loop / selection + assembly.  Run on the GPU box."""
import os, sys, tempfile, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd import forecaster as F
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims


class State:
    def __init__(self, p):
        self.position = (float(p[0]), float(p[1]))


PREC = os.environ.get("JMID_PREC", "f16x3")


def run(tag, N, K, k_ret, H, step, reps=20):
    d = tempfile.mkdtemp()
    env, ypath = F.write_configs(d, joint=True, ctx_dim=256, N=N, K=K, k_ret=k_ret, H=H, step=step)
    f = F.HumanTrajectoryForecasterSim(env, ypath, weights=JMIDWeights.from_seed(NetDims(ctx_dim=256), 0), precision=PREC)
    rng = np.random.default_rng(0)
    p = rng.uniform(-1.5, 1.5, (N, 2)); v = rng.uniform(-0.5, 0.5, (N, 2))
    for i in range(8):
        f.update_state_hists(State((0.0, -2.0 + 0.2 * i)), [State(p[j] + v[j] * 0.25 * i) for j in range(N)], 0.25 * i)
    f.predict_ret_best()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); f.predict_ret_best(); ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e3
    print(f"{tag} [{PREC}]: predict_ret_best() median {np.median(ts):.3f} ms  min {ts.min():.3f}  max {ts.max():.3f}   last call: "
          + ", ".join(f"{k} {v:.3f}" for k, v in f.timings.items()), flush=True)
    return f


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "shipped":       # the shipped operating point only (under rocprofv3: tools/shipped_profile.sh)
        run("shipped N=3 K=100->15 H=8 2 steps", 3, 100, 15, 8, 2, reps=int(sys.argv[2]) if len(sys.argv) > 2 else 50)
        raise SystemExit(0)
    f = run("cfg2  N=5 K=20 H=12 50 steps", 5, 20, 20, 12, 50)
    run("shipped N=3 K=100->15 H=8 2 steps", 3, 100, 15, 8, 2)
    run("N=5 K=20 H=12 2 steps", 5, 20, 20, 12, 2)
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        f.predict_ret_best()
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
