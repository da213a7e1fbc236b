#!/usr/bin/env python3
"""End-to-end training at the C2 configuration on the GPU alone (no oracle in the loop): 3 synthetic DTU-size views
(512x640, analytic sphere-over-floor colours), the reference's train() wiring — RayBank batches of 4096 rays, stratified
jitter, coarse 64 + fine 64+128 samples, two D=8/W=256 networks, mse(rgb) + mse(rgb0), Adam, exponential lr decay — and a
held-out view rendered every EVAL steps (PSNR as H:10).  Shows that the path learns at full size and what a step and an
evaluation cost.   usage: python scripts/train_demo.py [steps] [eval_every] [consistency]
With a third argument the ConsistentNeRF terms are switched on (run_nerf_view surface): hard masks from the cross-view
depth warp of noisy depth priors (a12/a13), masked photometric + depth losses on both levels (a14, hardmask_coef 0.2)."""
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
from consistentnerf_amd import raybank as RB, run_nerf as R, run_nerf_view as V  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
EVAL = int(sys.argv[2]) if len(sys.argv) > 2 else 500
CONSIST = len(sys.argv) > 3
H, W, FOCAL, NEAR, FAR, B = 512, 640, 1446.0, 2.125, 4.67, 4096


def main():
    dev = torch.device("cuda:0")
    K = I.intrinsics(H, W, FOCAL)
    poses = np.stack([I.camera_pose(th, -20.0, 3.0) for th in (0.0, 25.0, -25.0, 12.0)])
    scene = [I.analytic_scene(H, W, K, p) for p in poses]
    images = np.stack([s_[1] for s_ in scene])
    true_depth = np.stack([s_[0] for s_ in scene])
    args = argparse.Namespace(
        multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=128, netdepth=8, netwidth=256,
        netdepth_fine=8, netwidth_fine=256, netchunk=1024 * 64, lrate=5e-4, basedir=tempfile.mkdtemp(), expname="demo",
        ft_path=None, no_reload=True, perturb=1.0, N_samples=64, white_bkgd=False, raw_noise_std=0.0,
        dataset_type="dtu", no_ndc=True, lindisp=False)
    torch.manual_seed(0)
    kw, kw_test, start, grad_vars, opt = (V if CONSIST else R).create_nerf(args)
    kw.update(near=NEAR, far=FAR); kw_test.update(near=NEAR, far=FAR)
    bank = RB.RayBank(images, poses, H, W, K, [0, 1, 2], device=dev, seed=0)
    test_img = torch.from_numpy(images[3]).to(dev)
    test_depth = torch.from_numpy(true_depth[3]).to(dev)
    if CONSIST:
        # depth priors = true depth + 2 % noise (an MVS-like prior); hard masks from the warp; per-ray prior / mask follow
        # the bank's own permutation (same seed -> same shuffle, R:692) and its epoch reshuffles
        rs = np.random.RandomState(1)
        priors = (true_depth[:3] * (1 + 0.02 * rs.normal(size=true_depth[:3].shape))).astype(np.float32)
        t_m = time.perf_counter()
        masks = V.compute_hard_masks(H, W, K, poses[:3], priors, [0, 1, 2], 0.1, device=dev)
        torch.cuda.synchronize()
        print("hard masks:", round(time.perf_counter() - t_m, 3), "s, fraction", float(masks.mean()), flush=True)
        perm = torch.as_tensor(RB._perm_like_numpy_shuffle(3 * H * W, 0), device=dev)
        prior_r = torch.from_numpy(priors.reshape(-1)).to(dev)[perm]
        mask_r = torch.from_numpy(masks.reshape(-1).astype(np.float32)).to(dev)[perm]
    curve, evals = [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        lo = bank.i_batch
        if CONSIST and lo + B >= len(bank):      # the bank reshuffles inside next_batch: apply the same permutation
            ridx = torch.randperm(len(bank), device=dev)
            batch_rays, target = bank.next_batch(B, rand_idx=ridx)
            pr_s, m_s = prior_r[lo:lo + B], mask_r[lo:lo + B]
            prior_r, mask_r = prior_r[ridx], mask_r[ridx]
        else:
            batch_rays, target = bank.next_batch(B)
            if CONSIST:
                pr_s, m_s = prior_r[lo:lo + B], mask_r[lo:lo + B]
        opt.zero_grad()
        if CONSIST:
            rgb, disp, acc, depth, ex = V.render(H, W, K, chunk=32768, rays=batch_rays, retraw=True, **kw)
            n = rgb.shape[0]
            il, dl = V.hardmask_losses(rgb, target, m_s[:n], 0.2, depth, pr_s[:n], FAR)
            il0, dl0 = V.hardmask_losses(ex["rgb0"], target, m_s[:n], 0.2, ex["depth0"], pr_s[:n], FAR)
            loss = il + il0 + dl + dl0
        else:
            rgb, disp, acc, ex = R.render(H, W, K, chunk=32768, rays=batch_rays, retraw=True, **kw)
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
                out = V.render(H, W, K, chunk=32768, c2w=torch.from_numpy(poses[3][:3, :4]), **kw_test)
                img, dep = out[0], out[3]
                psnr = R.mse2psnr(R.img2mse(img, test_img)).item()
                derr = (dep - test_depth).abs().mean().item()
            torch.cuda.synchronize()
            evals.append({"step": i + 1, "heldout_psnr_dB": round(psnr, 3), "heldout_depth_mae": round(derr, 4),
                          "loss": round(loss.item(), 6),
                          "eval_s": round(time.perf_counter() - te, 3)})
            print(evals[-1], flush=True)
    torch.cuda.synchronize()
    total = time.perf_counter() - t0
    ev_s = sum(e["eval_s"] for e in evals)
    print(json.dumps({"config": "C2 on 3 synthetic 512x640 views, 4096 rays/step" + (" + hard-mask / depth consistency terms" if CONSIST else ""), "steps": STEPS, "epochs": bank.epochs,
                      "train_s": round(total - ev_s, 2), "ms_per_step_incl_host": round((total - ev_s) / STEPS * 1e3, 3),
                      "evals": evals, "loss_every_50": curve}))


if __name__ == "__main__":
    main()
