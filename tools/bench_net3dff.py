"""Multi-view memory update of the Pretrain front-end (PRE-POL:136-189) at full size: B environments x 4 views of
ViT-L/14@336 + delete + update per step.  Prints ms/step and views/s (warm memory)."""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from dynam3d_amd.net_3dff import Net_3DFF  # noqa: E402
from dynam3d_amd.policy import PolicyConfig, synth_policy_weights  # noqa: E402
from dynam3d_amd.profiling import TIMER  # noqa: E402
from dynam3d_amd.synthetic import SyntheticEpisodes  # noqa: E402
from dynam3d_amd.weights import ff_param_spec, synth_state_dict  # noqa: E402
from dynam3d_amd.towers import clip_param_spec  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    steps, warm = 6, 3
    cfg = PolicyConfig()
    sd = synth_state_dict(ff_param_spec(768) + clip_param_spec(cfg.vit), seed=0)
    net = Net_3DFF(cfg.vit, sd, device="cuda", batch_size=B, max_steps=steps + warm + 1)
    net.feature_fields.initialize_camera_setting(90.0, 90.0)
    eps = [SyntheticEpisodes(B, seed=30 + a, image_hw=224, depth_hw=224) for a in range(12)]
    times = []
    for t in range(warm + steps):
        frs = [ep.next() for ep in eps]
        obs = {}
        for a, fr in enumerate(frs):
            sfx = "" if a == 0 else f"_{a}"
            obs["rgb" + sfx], obs["depth" + sfx] = torch.from_numpy(fr.rgb).cuda(), torch.from_numpy(fr.depth).cuda()
        segm = np.stack([frs[(12 - v) % 12].patch_segm for v in (0, 3, 6, 9)], 1).reshape(B * 4, 1, 24, 24)
        net.positions, net.headings = [p.tolist() for p in frs[0].positions], list(frs[0].headings)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        net(obs, patch_segm=segm)
        torch.cuda.synchronize()
        times.append((time.perf_counter() - t0) * 1e3)
    ms = float(np.mean(times[warm:]))
    rows = net.feature_fields.state.count(0, net.feature_fields.state.ROWS)
    print(f"B={B} V=4: {ms:.2f} ms/step ({B * 4 / ms * 1e3:.0f} views/s), stored rows/env {rows}, all steps {[round(x, 1) for x in times]}")


if __name__ == "__main__":
    main()
