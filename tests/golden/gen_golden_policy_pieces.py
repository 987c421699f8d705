"""g18: the pieces of Policy_Dynam3D_VLN.py ("VLN-POL") that are pure torch, EXECUTED from the mounted reference file.  Container-only.

VLN-POL cannot be imported (habitat, gym, cv2, peft, transformers 4.46 attribute layout), so -- like g7 -- the relevant `ast` nodes
are located in the mounted file and compiled in place; nothing is copied into the repository:

  * `preprocess_depth` (VLN-POL:171-186): the FunctionDef, run on seeded depth images with exact zeros, an all-zero column and two
    depth scales;
  * the five prefix `nn.Sequential`s (VLN-POL:83-111): the `self.<name> = nn.Sequential(...)` assignments of `__init__`, executed with
    `width = 768`, filled with the name-keyed synthetic weights (dynam3d_amd/weights.py);
  * the evaluation branch's prefix expressions (VLN-POL:432-435): the four assignments (`patch_3d_info`, `patch_position_fts`,
    `batch_instance_fts`, `batch_zone_fts`), executed on seeded inputs with those modules.
Pins oracle/geometry.py::preprocess_depth (a1) and oracle/towers_ref.py::prefix_tokens (a14)."""
import ast
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_harness as rh  # noqa: E402
from dynam3d_amd.policy import prefix_param_spec  # noqa: E402
from dynam3d_amd.weights import synth_state_dict  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))
PATH = os.path.join(rh.REF_ROOT, "Dynam3D_VLN/vlnce_baselines/models/Policy_Dynam3D_VLN.py")
NAMES = ("patch_position_embedding", "instance_position_embedding", "zone_position_embedding", "instance_projector", "zone_projector")
EVAL_TARGETS = ("patch_3d_info", "patch_position_fts", "batch_instance_fts", "batch_zone_fts")


def _run(nodes, ns):
    exec(compile(ast.Module(body=list(nodes), type_ignores=[]), PATH, "exec"), ns)


def main():
    tree = ast.parse(open(PATH).read())
    cls = [n for n in ast.walk(tree) if isinstance(n, ast.ClassDef) and n.name == "Dynam3D_VLN"][0]
    out = dict(torch=torch.__version__)
    # ---- preprocess_depth --------------------------------------------------------------------------------------------------------
    fn = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "preprocess_depth"][0]
    ns = {"torch": torch}
    _run([fn], ns)
    rng = np.random.default_rng(180)
    cases = []
    for i, (B, H, W, scale) in enumerate([(2, 24, 24, (0.0, 10.0)), (3, 37, 29, (0.0, 10.0)), (2, 64, 48, (0.5, 5.0))]):
        d = rng.uniform(0.0, 1.0, (B, H, W, 1)).astype(np.float32)
        d[rng.random(d.shape) < 0.05] = 0.0
        d[0, :, 3] = 0.0                                              # a column without any valid pixel stays zero
        d[-1, H // 2, :] = 0.0
        r = ns["preprocess_depth"](None, torch.from_numpy(d.copy()), scale) if scale != (0.0, 10.0) else ns["preprocess_depth"](None, torch.from_numpy(d.copy()))
        out[f"depth_in_{i}"], out[f"depth_out_{i}"], out[f"depth_scale_{i}"] = d, r.numpy(), np.array(scale)
        cases.append(i)
    out["n_depth"] = np.int64(len(cases))
    # ---- the five nn.Sequential definitions of __init__ -----------------------------------------------------------------------
    init = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "__init__"][0]
    assigns = [n for n in ast.walk(init) if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Attribute) and n.targets[0].attr in NAMES]
    assert sorted(a.targets[0].attr for a in assigns) == sorted(NAMES), [a.targets[0].attr for a in assigns]
    me = SimpleNamespace()
    _run(assigns, {"self": me, "nn": nn, "width": 768})
    sd = synth_state_dict(prefix_param_spec(768), seed=0)
    with torch.no_grad():
        for name in NAMES:
            mod = getattr(me, name)
            assert isinstance(mod, nn.Sequential)
            for pn, p in mod.named_parameters():
                p.copy_(sd[f"{name}.{pn}"])
            mod.eval()
    # ---- VLN-POL:432-435 ----------------------------------------------------------------------------------------------------------
    fwd = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name == "forward"][0]
    stmts = [n for n in ast.walk(fwd) if isinstance(n, ast.Assign) and isinstance(n.targets[0], ast.Name) and n.targets[0].id in EVAL_TARGETS]
    with_ = [n for n in ast.walk(fwd) if isinstance(n, ast.With)]
    eval_with = max(with_, key=lambda w: w.lineno)                     # the evaluation branch's `with torch.no_grad():` is the last one
    stmts = sorted([s for s in stmts if s.lineno > eval_with.lineno], key=lambda s: s.lineno)[:4]
    assert [s.targets[0].id for s in stmts] == list(EVAL_TARGETS), [(s.targets[0].id, s.lineno) for s in stmts]
    g = torch.Generator().manual_seed(181)
    B, P = 2, 576
    rnd = lambda *s: torch.randn(*s, generator=g)
    ins = dict(batch_rel_x=rnd(B, P, 1) * 2, batch_rel_y=rnd(B, P, 1).abs() * 3, batch_rel_z=rnd(B, P, 1), batch_direction=torch.rand(B, P, 1, generator=g) * 6.2831853,
               batch_scale=torch.rand(B, P, 1, generator=g) * 0.4)
    n_i, n_z = [5, 9], [3, 1]
    ins["batch_instance_fts"] = [rnd(n, 768) * 0.3 for n in n_i]
    ins["batch_instance_relative_position"] = [rnd(n, 3) * 2 for n in n_i]
    ins["batch_zone_fts"] = [rnd(n, 768) * 0.3 for n in n_z]
    ins["batch_zone_relative_position"] = [rnd(n, 3) * 20 for n in n_z]
    for k, v in ins.items():
        if isinstance(v, list):
            for b, t in enumerate(v):
                out[f"{k}_{b}"] = t.numpy().copy()
        else:
            out[k] = v.numpy().copy()
    ns = dict(ins, self=me, torch=torch, batch_size=B)
    with torch.no_grad():
        _run(stmts, ns)
    out["patch_rows"] = np.arange(0, P, 48)
    out["patch_position_fts_rows"] = ns["patch_position_fts"][:, ::48].numpy()          # (B, 12, 3072)
    out["patch_position_fts_rowsum"] = ns["patch_position_fts"].double().sum(-1).numpy()
    for b in range(B):
        out[f"instance_tokens_{b}"] = ns["batch_instance_fts"][b].numpy()
        out[f"zone_tokens_{b}"] = ns["batch_zone_fts"][b].numpy()
    np.savez_compressed(os.path.join(OUT, "g18_policy_pieces.npz"), **out)
    print("g18 ok:", ns["patch_position_fts"].shape, [t.shape for t in ns["batch_instance_fts"]], [t.shape for t in ns["batch_zone_fts"]])


if __name__ == "__main__":
    main()
