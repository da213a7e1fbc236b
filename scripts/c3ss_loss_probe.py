"""Diagnostic: where does c3_ss's final loss of ~1e4 come from?  Runs a few steps of bench.c3_ss_step_fn's
scene through run_nerf_view.ss_step_loss and prints every term, the ranges of the maps they compare and the geometry of the warp."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa: E402
from consistentnerf_amd import raybank as RB, run_nerf as R, run_nerf_view as V  # noqa: E402

dev = torch.device("cuda:0")
sc = bench.c3_scene(dev)
H, W, K, kw, opt = sc["H"], sc["W"], sc["K"], sc["kw"], sc["opt"]
rs = np.random.RandomState(3)
for i in range(6):
    v, r = i % 3, (i + 1 + int(rs.randint(0, 2))) % 3
    rays, target, sel, (d_prior,) = RB.sample_patch_rays(sc["img_t"][v], sc["poses"][v], H, W, K, 4096, None, extras=(sc["dep_t"][v],),
                                                         render_kwargs=kw)
    coins = [int(c) for c in rs.randint(0, 2, 4)]
    loss, ss = V.ss_step_loss(H, W, K, rays, target, d_prior, sc["poses"][r], sc["img_t"][r], sc["dep_t"][r], kw, chunk=32768,
                              occlusion_threshold=0.1, with_depth_loss=True, coins=coins)
    opt.zero_grad()
    R.backward(loss)
    opt.step()
    f = lambda t: (float(t.min()), float(t.max()), float(t.mean()))  # noqa: E731
    print(f"step {i} v={v} r={r} coins={coins} loss={float(loss):.4f} primary={float(ss['loss_primary']):.4f} ref={float(ss['loss_ref']):.4f} "
          f"thr={float(ss['threshold']):.4f} M={ss['batch_rays_ref'].shape[1]} nsel={int(ss['sel'].sum())}")
    print("   depth_prior(batch)", f(d_prior), " depth_pred", f(ss["depth_pred"]), " rays_depth_ref", f(ss["rays_depth_ref"]),
          " depth_pred_ref", f(ss["depth_pred_ref"]))
    print("   rays_d_ref z", f(ss["batch_rays_ref"][1][:, 2]), " rays_d z (batch)", f(rays[1][:, 2]), " |rays_d_ref|", f(ss["batch_rays_ref"][1].norm(dim=-1)))
    print("   rgb_ref", f(ss["rgb_ref"]), " rgb_target_ref", f(ss["rgb_target_ref"]), " acc_ref", f(ss["extras_ref"]["acc0"]))
