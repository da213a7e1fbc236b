#!/usr/bin/env python3
"""BASELINE config C3 on the GPU box (side bench, not the headline metric): LLFF-like 3-view training step WITH the
consistency terms — hard masks from the cross-view depth warp (a12/a13), masked photometric + depth losses on both
levels (a14), the monocular-depth patch term on 4 16x16 patches (f-5), gradient value-clip 0.1 + Adam (f-1).
378x504 views, no_ndc, near 1.2 / far 12, 4096 random rays + 1024 patch rays per step, 64 + 128 samples, D=8/W=256.
usage: python scripts/bench_c3.py [steps]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import _inputs as I  # noqa: E402
from consistentnerf_amd import raybank as RB, run_nerf_view as V  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    dev = torch.device("cuda:0")
    H, W, focal, near, far = 378, 504, 407.0, 1.2, 12.0
    args = argparse.Namespace(
        multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=128, netdepth=8, netwidth=256,
        netdepth_fine=8, netwidth_fine=256, netchunk=1024 * 64, lrate=5e-4, basedir=tempfile.mkdtemp(), expname="c3",
        ft_path=None, no_reload=True, perturb=1.0, N_samples=64, white_bkgd=False, raw_noise_std=0.0,
        dataset_type="llff", no_ndc=True, lindisp=False, stable_init=False)
    torch.manual_seed(0)
    np.random.seed(0)
    kw_train, _, _, grad_vars, optimizer = V.create_nerf(args)
    kw_train.update(near=near, far=far)
    optimizer.param_groups[0]['clip_value'] = 0.1                      # V:1983
    K = I.intrinsics(H, W, focal)
    poses = np.stack([I.camera_pose(th, -10.0, 4.0) for th in (0.0, 6.0, -6.0)])
    scene = [I.analytic_scene(H, W, K, p) for p in poses]
    depths = np.stack([s[0] for s in scene]) + np.random.normal(0, 0.02, (3, H, W)).astype(np.float32)
    images = np.stack([s[1] for s in scene])
    t0 = time.perf_counter()
    masks = V.compute_hard_masks(H, W, K, poses, depths, [0, 1, 2], 0.1, device=dev)
    torch.cuda.synchronize()
    t_masks = time.perf_counter() - t0
    mono = 1.0 / np.maximum(depths, 1e-3)                              # a monocular inverse-depth prior stand-in
    img_t = [torch.from_numpy(images[i]).to(dev) for i in range(3)]
    dep_t = [torch.from_numpy(depths[i]).to(dev) for i in range(3)]
    msk_t = [torch.from_numpy(masks[i].astype(np.float32)).to(dev) for i in range(3)]
    mono_t = [torch.from_numpy(mono[i].astype(np.float32)).to(dev) for i in range(3)]
    N_rand, B = 4096, 4096 + 1024

    def step(i):
        v = i % 3
        starts = RB.draw_patch_starts(H, W, 4, 16)
        rays, target, sel, (d_prior, m, mono_s) = RB.sample_patch_rays(
            img_t[v], poses[v], H, W, K, N_rand, starts, extras=(dep_t[v], msk_t[v], mono_t[v]))
        rgb, disp, acc, depth, extras = V.render(H, W, K, chunk=32768, rays=rays, retraw=True, **kw_train)
        optimizer.zero_grad()
        img_loss, depth_loss = V.hardmask_losses(rgb, target, m, 0.2, depth, d_prior, far)
        img_loss0, depth_loss0 = V.hardmask_losses(extras['rgb0'], target, m, 0.2, extras['depth0'], d_prior, far)
        loss = img_loss + img_loss0 + 0.1 * (depth_loss + depth_loss0)
        loss = loss + 0.001 * (V.midas_patch_loss(depth, mono_s, 4, 16) + V.midas_patch_loss(extras['depth0'], mono_s, 4, 16))
        loss.backward()
        optimizer.step()
        for pg in optimizer.param_groups:
            pg['lr'] = 5e-4 * (0.1 ** (i / 250000))
        return loss

    for i in range(3):
        step(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(3 + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    out = {"config": "C3: 3 LLFF-like views 378x504, no_ndc, hard masks + masked rgb/depth losses on both levels + "
                     "monocular patch term, clip 0.1 + Adam, 4096 + 1024 rays/step, 64+128 samples, D=8/W=256",
           "ms_per_step": dt * 1e3, "ray_samples_per_s": B * 256 / dt, "train_tflops": B * 256 * 3489024 / dt / 1e12,
           "hard_masks_3views_s": t_masks, "hard_mask_fraction": float(masks.mean()), "final_loss": float(loss.item()),
           "finite": bool(np.isfinite(loss.item()))}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
