#!/usr/bin/env python3
"""End-to-end training at the C2 configuration on the GPU alone (no oracle in the loop): 3 synthetic DTU-size views
(512x640, analytic sphere-over-floor colours), the reference's train() wiring — RayBank batches of 4096 rays, stratified
jitter, coarse 64 + fine 64+128 samples, two D=8/W=256 networks, mse(rgb) + mse(rgb0), Adam, exponential lr decay — and a
held-out view rendered every EVAL steps (PSNR as H:10).  Shows that the path learns at full size and what a step and an
evaluation cost.   usage: python scripts/train_demo.py [steps] [eval_every]"""
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
from consistentnerf_amd import raybank as RB, run_nerf as R  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
EVAL = int(sys.argv[2]) if len(sys.argv) > 2 else 500
H, W, FOCAL, NEAR, FAR, B = 512, 640, 1446.0, 2.125, 4.67, 4096


def main():
    dev = torch.device("cuda:0")
    K = I.intrinsics(H, W, FOCAL)
    poses = np.stack([I.camera_pose(th, -20.0, 3.0) for th in (0.0, 25.0, -25.0, 12.0)])
    images = np.stack([I.analytic_scene(H, W, K, p)[1] for p in poses])
    args = argparse.Namespace(
        multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=128, netdepth=8, netwidth=256,
        netdepth_fine=8, netwidth_fine=256, netchunk=1024 * 64, lrate=5e-4, basedir=tempfile.mkdtemp(), expname="demo",
        ft_path=None, no_reload=True, perturb=1.0, N_samples=64, white_bkgd=False, raw_noise_std=0.0,
        dataset_type="dtu", no_ndc=True, lindisp=False)
    torch.manual_seed(0)
    kw, kw_test, start, grad_vars, opt = R.create_nerf(args)
    kw.update(near=NEAR, far=FAR); kw_test.update(near=NEAR, far=FAR)
    bank = RB.RayBank(images, poses, H, W, K, [0, 1, 2], device=dev, seed=0)
    test_img = torch.from_numpy(images[3]).to(dev)
    curve, evals = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        batch_rays, target = bank.next_batch(B)
        rgb, disp, acc, ex = R.render(H, W, K, chunk=32768, rays=batch_rays, retraw=True, **kw)
        opt.zero_grad()
        loss = R.img2mse(rgb, target) + R.img2mse(ex["rgb0"], target)
        loss.backward()
        opt.step()
        for g_ in opt.param_groups:
            g_["lr"] = args.lrate * (0.1 ** (i / (250 * 1000)))
        if i % 50 == 0:
            curve.append(round(loss.item(), 6))
        if (i + 1) % EVAL == 0 or i + 1 == STEPS:
            torch.cuda.synchronize()
            te = time.perf_counter()
            with torch.no_grad():
                img, *_ = R.render(H, W, K, chunk=32768, c2w=torch.from_numpy(poses[3][:3, :4]), **kw_test)
                psnr = R.mse2psnr(R.img2mse(img, test_img)).item()
            torch.cuda.synchronize()
            evals.append({"step": i + 1, "heldout_psnr_dB": round(psnr, 3), "loss": round(loss.item(), 6),
                          "eval_s": round(time.perf_counter() - te, 3)})
            print(evals[-1], flush=True)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    ev_s = sum(e["eval_s"] for e in evals)
    print(json.dumps({"config": "C2 on 3 synthetic 512x640 views, 4096 rays/step", "steps": STEPS, "epochs": bank.epochs,
                      "train_s": round(total - ev_s, 2), "ms_per_step_incl_host": round((total - ev_s) / STEPS * 1e3, 3),
                      "evals": evals, "loss_every_50": curve}))


if __name__ == "__main__":
    main()
