"""ONE full predictor call on one chunk of episodes (encoder -> 50-step loop -> integrator), nothing else: the process
rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes run on to get the HBM bytes of a whole call (profiles/*_pmc_traffic.json)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safe_interactive_crowdnav_amd.engine import JmidEngine
from safe_interactive_crowdnav_amd.scene import synthetic_episodes
from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

E = int(sys.argv[1]) if len(sys.argv) > 1 else 51
N, K, H = 5, 20, 12
dev = torch.device("cuda", 0)
eng = JmidEngine(JMIDWeights.from_seed(NetDims(ctx_dim=256), 0), joint=True, device_id=0, step=50)
eng.set_tuning("lanes", 1)      # one chunk, one launch per kernel and step: the per-launch counters are those of a 51-episode launch
syn = synthetic_episodes(E, N, seed=0, horizon=H)
x_st = torch.from_numpy(syn["x_st"].reshape(E * N, 6, 6)).to(dev)
nbr = torch.from_numpy(syn["nbr_sum"].reshape(E * N, 2, 6, 6)).to(dev)
emask = torch.from_numpy(syn["edge_mask"].reshape(E * N, 2)).to(dev)
p0 = torch.from_numpy(syn["p0"]).to(dev)
x_T = torch.randn([E, K * N, H, 2], generator=torch.Generator().manual_seed(0)).to(dev)
ctx = eng.encode(x_st, nbr, emask)
vel, pos = eng.denoise(x_T, ctx.view(E, N, -1), p0, dt=0.25, precision=os.environ.get("JMID_PREC", "f16x2"), want_vel=False)
eng.synchronize()
print("trajectories", E * N * K)
