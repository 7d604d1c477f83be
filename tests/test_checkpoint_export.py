"""export_checkpoint on a checkpoint in the REFERENCE'S OWN pickled format.

The trained blobs are absent (/root/reference/.MISSING_LARGE_BLOBS), so every parity number in this repo is on
random-init weights - but the container format can still be exercised for real: the reference's own module tree builds
the predictor (tests/golden/make_golden.py::make_forecaster, seeded weights), MID._save_model's dict
({"encoder": registrar.model_dict - a pickled nn.ModuleDict of the reference's classes -, "ddpm": model.state_dict()},
sicnav_diffusion/JMID/MID/mid.py:1501-1509) is written with torch.save, and the exporter has to unpickle it the way
mid.py:1230-1232 does, pick the live tensors out of it (the state dict also carries the never-executed template
`layer.*` and the var_sched buffers; the ModuleDict every encoder module the registrar created) and reproduce them
bit for bit.  Needs the reference tree importable: runs in the build container, skipped on the GPU box."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "sicnav_diffusion")),
                                reason="needs the reference tree (build container only)")


@pytest.fixture(scope="module")
def golden_tools():
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "make_golden.py")
    argv = sys.argv
    sys.argv = [path]                    # the module reads sys.argv[1] as a fixture-name filter at import
    try:
        spec = importlib.util.spec_from_file_location("make_golden_for_export_test", path)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)     # installs the import shims and imports the reference; generates nothing
    finally:
        sys.argv = argv
    return mod


@pytest.mark.parametrize("joint", [True, False])
def test_exporter_reads_a_checkpoint_pickled_by_the_reference_classes(joint, golden_tools, tmp_path):
    from safe_interactive_crowdnav_amd import export_checkpoint, forecaster
    from safe_interactive_crowdnav_amd.weights import JMIDWeights, NetDims

    dims = NetDims(ctx_dim=32)
    f, weights = golden_tools.make_forecaster(joint, 32, N=3, K=4, k_ret=4, H=8, step=2, wseed=61)
    mid = f.mid_model
    ckpt = {"encoder": mid.registrar.model_dict, "ddpm": mid.model.state_dict()}       # MID._save_model, mid.py:1501-1509
    pt = str(tmp_path / "sim_epoch1.pt")
    torch.save(ckpt, pt)
    sd = ckpt["ddpm"]
    assert any(k.startswith("vel_predictor.net.layer.") for k in sd) or not joint      # template copy rides along (JMID)
    assert any(k.startswith("vel_predictor.var_sched.") for k in sd)
    assert len(mid.registrar.model_dict) >= 5                                           # more modules than the 5 live ones

    out = export_checkpoint.export(pt, str(tmp_path / "sim_epoch1.npz"), dims, reference_root=REF)
    w = JMIDWeights.load(out)
    assert w.dims == dims and w.checksum() == weights.checksum()                        # every live tensor, bit for bit
    # the predictor's loader finds the export next to the .pt path the yaml names (forecaster.load_weights)
    w2 = forecaster.load_weights(pt, dims)
    assert w2.checksum() == weights.checksum()
    # and the command-line entry point does the same
    out2 = str(tmp_path / "cli.npz")
    export_checkpoint.main([pt, "-o", out2, "--encoder-dim", "32", "--tf-layer", "3", "--reference-root", REF])
    assert JMIDWeights.load(out2).checksum() == weights.checksum()
    # sanity: what was exported is what the reference itself computes with (one net evaluation)
    net = mid.model.vel_predictor.net.eval()       # dropout off, as in inference
    g = torch.Generator().manual_seed(1)
    x, ctx = torch.randn([4, 8, 2], generator=g), torch.randn([4, 32], generator=g)
    beta = mid.model.vel_predictor.var_sched.betas[[100] * 4]
    from oracle import jmid_oracle as O
    with torch.no_grad():
        e_ref = net([x, ctx], beta=beta)
        e_ours = O.net_forward(w.tensors, x, ctx, beta, joint=joint)
    np.testing.assert_allclose(e_ours.numpy(), e_ref.numpy(), rtol=0, atol=1e-6)
