"""Convert the reference's two-part checkpoint into the neutral flat ``.npz`` the predictor loads.

The reference stores ``{"encoder": registrar.model_dict (a pickled nn.ModuleDict), "ddpm": state_dict}``
(``sicnav_diffusion/JMID/MID/mid.py:1501-1523``); unpickling it needs the reference's module tree importable
(``mid.py:1230``) and ``weights_only=False``.  Run this once where the reference is installed:

    python -m safe_interactive_crowdnav_amd.export_checkpoint \\
        sicnav_diffusion/JMID/MID/checkpoints/sim_inference_checkpoints/sim_gen_sicnav_p_midjp_cvg_epoch121.pt \\
        --encoder-dim 256 --tf-layer 3 [--reference-root /path/to/safe-interactive-crowdnav]

and point ``model_path`` at the ``.pt`` (the ``.npz`` next to it is picked up) or at the ``.npz`` itself.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

from .weights import JMIDWeights, NetDims


def export(pt_path: str, out_path: str, dims: NetDims, reference_root: str | None = None) -> str:
    if reference_root:
        sys.path.insert(0, os.path.join(reference_root, "sicnav_diffusion", "JMID", "MID"))
        sys.path.insert(0, reference_root)
    ckpt = torch.load(pt_path, map_location="cpu", weights_only=False)
    w = JMIDWeights.from_reference_state(dims, ckpt["ddpm"], ckpt["encoder"])
    w.save(out_path)
    return out_path


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("checkpoint")
    ap.add_argument("-o", "--output", default=None)
    ap.add_argument("--encoder-dim", type=int, default=256)
    ap.add_argument("--tf-layer", type=int, default=3)
    ap.add_argument("--reference-root", default=None)
    a = ap.parse_args(argv)
    out = a.output or os.path.splitext(a.checkpoint)[0] + ".npz"
    print(export(a.checkpoint, out, NetDims(ctx_dim=a.encoder_dim, tf_layer=a.tf_layer), a.reference_root))


if __name__ == "__main__":
    main()
