"""CPU: the oracle's restatements of VLN-POL's pure-torch pieces against g18 -- outputs of the reference's OWN code, `ast`-located in
the mounted file and executed (tests/golden/gen_golden_policy_pieces.py): `preprocess_depth` (a1, VLN-POL:171-186; bit-exact) and the
prefix MLP stacks + their composition (a14, VLN-POL:83-111, 432-435; float32 summation-order tolerance)."""
import numpy as np
import torch

from dynam3d_amd.policy import prefix_param_spec
from dynam3d_amd.weights import synth_state_dict
from oracle import geometry as G
from oracle import towers_ref as TR
from tests.golden_io import load


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_preprocess_depth_bit_exact_vs_reference_function():
    g = load("g18_policy_pieces.npz")
    for i in range(int(g["n_depth"])):
        scale = tuple(float(v) for v in g[f"depth_scale_{i}"])
        out = G.preprocess_depth(g[f"depth_in_{i}"], scale)
        assert np.array_equal(bits(out), bits(g[f"depth_out_{i}"])), i
    # the all-zero column stays zero * scale + min (VLN-POL:181: its column maximum is zero)
    assert np.all(g["depth_out_0"][0, :, 3] == 0.0) and np.all(g["depth_out_2"][0, :, 3] == np.float32(0.5))


def _g18_inputs(g):
    t = torch.from_numpy
    info6 = torch.cat([t(g["batch_rel_x"]), t(g["batch_rel_y"]), t(g["batch_rel_z"]), torch.sin(t(g["batch_direction"])), torch.cos(t(g["batch_direction"])),
                       t(g["batch_scale"])], -1)
    n = 2
    ifts = torch.cat([t(g[f"batch_instance_fts_{b}"]) for b in range(n)])
    irel = torch.cat([t(g[f"batch_instance_relative_position_{b}"]) for b in range(n)])
    zfts = torch.cat([t(g[f"batch_zone_fts_{b}"]) for b in range(n)])
    zrel = torch.cat([t(g[f"batch_zone_relative_position_{b}"]) for b in range(n)])
    inst_ref = np.concatenate([g[f"instance_tokens_{b}"] for b in range(n)])
    zone_ref = np.concatenate([g[f"zone_tokens_{b}"] for b in range(n)])
    return info6, ifts, irel, zfts, zrel, inst_ref, zone_ref


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / np.linalg.norm(b))


def test_prefix_tokens_vs_reference_modules():
    g = load("g18_policy_pieces.npz")
    sd = synth_state_dict(prefix_param_spec(768), seed=0)
    info6, ifts, irel, zfts, zrel, inst_ref, zone_ref = _g18_inputs(g)
    with torch.no_grad():
        patch, inst, zone = TR.prefix_tokens(info6, ifts, irel, zfts, zrel, sd)
    r = (rel(patch[:, ::48].numpy(), g["patch_position_fts_rows"]), rel(inst.numpy(), inst_ref), rel(zone.numpy(), zone_ref))
    assert max(r) < 2e-6, r
    assert np.abs(patch.double().sum(-1).numpy() - g["patch_position_fts_rowsum"]).max() < 1e-3
