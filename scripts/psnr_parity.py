#!/usr/bin/env python3
"""PSNR-parity run (BASELINE metric, second half): train the HIP path and the CPU oracle from the SAME initial
weights on the SAME ray batches of a synthetic DTU-like 3-view scene (analytic sphere-over-floor colours), with the
reference's deterministic RNG hook (pytest=True: identical jitter / resampling streams on both sides), then render
a held-out view with both and compare PSNR (definition H:10 / V:2047: -10 log10 of the mean MSE) and loss curves.
Reduced size so the CPU side finishes in minutes: 64x80 images, 512-ray batches, coarse 64 + fine 64+128,
D=8/W=256, viewdirs.   usage: python scripts/psnr_parity.py [steps]"""
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
from consistentnerf_amd import run_nerf as R  # noqa: E402
from oracle import nerf_oracle as O  # noqa: E402

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 150
H, W, FOCAL, NEAR, FAR, B = 64, 80, 180.0, 2.0, 6.0, 512


def main():
    dev = torch.device("cuda:0")
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    K = I.intrinsics(H, W, FOCAL)
    train_poses = [I.camera_pose(th, -20.0, 4.0) for th in (0.0, 25.0, -25.0)]
    test_pose = I.camera_pose(12.0, -15.0, 4.0)
    rays, cols = [], []
    for p in train_poses:
        ro, rd = O.get_rays(H, W, K, torch.from_numpy(p))
        rays.append(O.build_ray_batch(ro, rd, NEAR, FAR, True))
        cols.append(torch.from_numpy(I.analytic_scene(H, W, K, p)[1]).reshape(-1, 3))
    bank, target = torch.cat(rays), torch.cat(cols)
    perm = torch.from_numpy(np.random.RandomState(0).permutation(bank.shape[0]))
    bank, target = bank[perm], target[perm]
    ro, rd = O.get_rays(H, W, K, torch.from_numpy(test_pose))
    test_rays = O.build_ray_batch(ro, rd, NEAR, FAR, True)
    test_rgb = torch.from_numpy(I.analytic_scene(H, W, K, test_pose)[1]).reshape(-1, 3)

    # same initial weights on both sides (small-gain init so the random net is well conditioned, like a real run)
    sds = [I.nerf_state_dict(8, 256, 10, 4, 5, True, seed=s, gain=0.6) for s in (1, 2)]
    args = argparse.Namespace(
        multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=128, netdepth=8, netwidth=256,
        netdepth_fine=8, netwidth_fine=256, netchunk=1024 * 64, lrate=5e-4, basedir=tempfile.mkdtemp(), expname="p",
        ft_path=None, no_reload=True, perturb=1.0, N_samples=64, white_bkgd=False, raw_noise_std=0.0,
        dataset_type="dtu", no_ndc=True, lindisp=False)
    kw, kw_test, _, grad_vars, opt = R.create_nerf(args)
    kw["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in sds[0].items()})
    kw["network_fine"].load_state_dict({k: torch.from_numpy(v) for k, v in sds[1].items()})
    kw.update(near=NEAR, far=FAR); kw_test.update(near=NEAR, far=FAR)
    osd = [O.as_tensors(sd, True) for sd in sds]
    net, cfg = O.NetCfg(8, 256, output_ch=5), O.RenderCfg(64, 128, 1.0)
    params = [p for d in osd for p in d.values()]
    m = [torch.zeros_like(p) for p in params]; v = [torch.zeros_like(p) for p in params]

    hl, ol, t_hip, t_cpu = [], [], 0.0, 0.0
    lr = 5e-4
    for i in range(STEPS):
        lo = (i * B) % (bank.shape[0] - B)
        rb, tg = bank[lo:lo + B], target[lo:lo + B]
        # --- HIP path
        t0 = time.perf_counter()
        rgb, disp, acc, ex = R.render(H, W, K, chunk=32768, rays=torch.stack([rb[:, 0:3], rb[:, 3:6]]).to(dev),
                                      retraw=True, pytest=True, **kw)
        opt.zero_grad()
        loss = R.img2mse(rgb, tg.to(dev)) + R.img2mse(ex["rgb0"], tg.to(dev))
        loss.backward(); opt.step()
        for g_ in opt.param_groups:
            g_["lr"] = 5e-4 * (0.1 ** (i / 250000))
        hl.append(loss.item()); t_hip += time.perf_counter() - t0
        # --- CPU oracle
        t0 = time.perf_counter()
        out = O.render_rays_pytest(rb, osd[0], osd[1], net, cfg)
        lo_ = O.mse(out["rgb_map"], tg) + O.mse(out["rgb0"], tg)
        grads = torch.autograd.grad(lo_, params, allow_unused=True)
        with torch.no_grad():
            for p, g_, mm, vv in zip(params, grads, m, v):
                if g_ is not None:
                    O.adam_step(p, g_, mm, vv, i + 1, lr)
        lr = O.lr_at(5e-4, i, 250)
        ol.append(lo_.item()); t_cpu += time.perf_counter() - t0
        if i % 25 == 0 or i == STEPS - 1:
            print(f"step {i:4d}  loss hip {hl[-1]:.6f}  oracle {ol[-1]:.6f}  |d| {abs(hl[-1]-ol[-1]):.2e}", flush=True)
    with torch.no_grad():
        rgb_h, *_ = R.render(H, W, K, chunk=32768, rays=torch.stack([test_rays[:, 0:3], test_rays[:, 3:6]]).to(dev),
                             **kw_test)
        out = O.render_rays(test_rays, osd[0], osd[1], net, O.RenderCfg(64, 128, 0.0))
    psnr_h = O.psnr_from_mse(O.mse(rgb_h.cpu(), test_rgb)).item()
    psnr_o = O.psnr_from_mse(O.mse(out["rgb_map"], test_rgb)).item()
    cross = O.psnr_from_mse(O.mse(rgb_h.cpu(), out["rgb_map"])).item()
    res = {"steps": STEPS, "rays_per_step": B, "final_loss_hip": hl[-1], "final_loss_oracle": ol[-1],
           "max_abs_loss_diff": float(np.max(np.abs(np.array(hl) - np.array(ol)))),
           "heldout_psnr_hip_dB": psnr_h, "heldout_psnr_oracle_dB": psnr_o, "psnr_hip_vs_oracle_image_dB": cross,
           "s_per_step_hip": t_hip / STEPS, "s_per_step_cpu_oracle": t_cpu / STEPS,
           "loss_curve_hip": hl[::10], "loss_curve_oracle": ol[::10]}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
