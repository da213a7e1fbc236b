"""Soak of the in-loop consistency step's two routes: N steps of bench.c3_ss_step_fn's rig from the same initial weights, the same
batches / reference views / coins, through run_nerf_view.ss_step_loss(route="one_render") and (route="two_renders"); prints the loss
trajectory of both (block means) and their final held-in colour terms.  Step-level equality is what the GPU suite asserts; this is the
end-to-end sanity that hundreds of one-render steps train the same way (free-running fp32 Adam trajectories decorrelate: compare block
means, not steps).   usage: python scripts/c3ss_route_soak.py [steps]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa: E402
from consistentnerf_amd import raybank as RB, run_nerf as R, run_nerf_view as V  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
dev = torch.device("cuda:0")
out = {}
for route in ("one_render", "two_renders"):
    torch.manual_seed(0)
    np.random.seed(0)
    sc = bench.c3_scene(dev)
    H, W, K, kw, opt = sc["H"], sc["W"], sc["K"], sc["kw"], sc["opt"]
    rs = np.random.RandomState(3)
    torch.manual_seed(11)
    losses, img = [], []
    for i in range(steps):
        v, r = i % 3, (i + 1 + int(rs.randint(0, 2))) % 3
        rays, target, _, (d_prior,) = RB.sample_patch_rays(sc["img_t"][v], sc["poses"][v], H, W, K, 4096, None, extras=(sc["dep_ss_t"][v],),
                                                           render_kwargs=kw)
        coins = [int(c) for c in rs.randint(0, 2, 4)]
        loss, info = V.ss_step_loss(H, W, K, rays, target, d_prior, sc["poses"][r], sc["img_t"][r], sc["dep_ss_t"][r], kw, chunk=32768,
                                    occlusion_threshold=0.1, with_depth_loss=True, coins=coins, route=route)
        opt.zero_grad()
        R.backward(loss)
        opt.step()
        losses.append(loss.detach())
        img.append(info["img_loss"].detach())
    L = torch.stack(losses).cpu().numpy()
    I_ = torch.stack(img).cpu().numpy()
    nb = 8
    out[route] = {"loss_block_means": [float(x.mean()) for x in np.array_split(L, nb)],
                  "img_loss_block_means": [float(x.mean()) for x in np.array_split(I_, nb)], "finite": bool(np.isfinite(L).all())}
    del sc
    torch.cuda.empty_cache()
print(json.dumps({"steps": steps, **out}))
a, b = out["one_render"], out["two_renders"]
rel = [abs(x - y) / y for x, y in zip(a["img_loss_block_means"][-3:], b["img_loss_block_means"][-3:])]
print("last three blocks, |d img_loss| / img_loss:", [round(x, 3) for x in rel])
assert a["finite"] and b["finite"]
