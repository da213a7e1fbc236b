#!/usr/bin/env python3
"""bench.py — training ray-samples/sec of the MI355X NeRF hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one full optimisation step of BASELINE config C2 on synthetic DTU-like data, per GPU:
4096 rays -> coarse 64 samples -> fine 64+128 samples (D=8/W=256 MLPs with view directions, stratified
jitter, hierarchical resampling) -> mse(rgb)+mse(rgb0) -> backward (dgrad+wgrad of both nets) ->
[N>1: one RCCL all-reduce of the flat fp32 gradient] -> Adam + lr decay.  1 048 576 ray-samples per GPU per
step (64 + 192 network evaluations per ray); weak scaling (every rank renders its own 4096-ray
slice of the global batch).  Inputs (ray bank, targets, weights) are resident in HBM before the timed region.

Extra objects on the JSON line:
  roofline     — dominant kernel of the step, algorithmic FLOPs per launch / its average duration measured
                 with HIP events on the launch stream inside the timed region; peak = fp32 MFMA 157.3 TFLOP/s.
  cpu_baseline — the CPU oracle ("port" of the reference step: stock ATen, fp32) timed on this host's cores on
                 a bounded sample (rank 0, N=1 only).
  dist         — what the gradient exchange did: ranks RCCL saw, message count / bytes per step, and the part of the
                 exchange the step waited for (HIP events on the launch stream).  Present whenever a process group
                 exists (N>1, or CNERF_FORCE_DIST=1 on one GPU).
  extra        — (N=1, after the timed C2 region, not part of `value`) BASELINE configs[4] and configs[2] under the same
                 clock: c5 = one 756x1008 NDC frame through render() (perturb=0, chunk 32768) incl. the D2H of the frame;
                 c3 = 20 training steps with hard masks + masked rgb/depth losses on both levels + the monocular patch
                 term + clip 0.1 + Adam.  --no-extra skips them.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

MAC_FWD, MAC_DGRAD, MAC_WGRAD = 593408, 557696, 593408   # per ray-sample, D=8/W=256/viewdirs
PEAK_FP32_MFMA_TFLOPS = 157.3                              # MI355X_MICROARCH.md
B_PER_GPU, NC, NF = 4096, 64, 128
H_IMG, W_IMG, FOCAL, NEAR, FAR = 512, 640, 1446.0, 2.125, 4.67   # DTU-like


def make_args(tmpdir):
    return argparse.Namespace(
        multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=NF, netdepth=8, netwidth=256,
        netdepth_fine=8, netwidth_fine=256, netchunk=1024 * 64, lrate=5e-4, basedir=tmpdir, expname="bench",
        ft_path=None, no_reload=True, perturb=1.0, N_samples=NC, white_bkgd=False, raw_noise_std=0.0,
        dataset_type="dtu", no_ndc=True, lindisp=False)


def build_ray_bank(device, seed=0):
    """3 DTU-like views on a ring of radius 3 -> [3*H*W, 11] rays + U[0,1) targets, shuffled with a fixed seed
    (identical on every rank)."""
    import _inputs as I
    from consistentnerf_amd import ops
    K = I.intrinsics(H_IMG, W_IMG, FOCAL)
    banks = [ops.gen_rays(H_IMG, W_IMG, K, I.camera_pose(th, -20.0, 3.0), NEAR, FAR, True, False, device)
             for th in (0.0, 25.0, -25.0)]
    rays = torch.cat(banks, 0)
    g = torch.Generator(device="cpu").manual_seed(seed)
    perm = torch.randperm(rays.shape[0], generator=g).to(device)
    rays = rays[perm].contiguous()
    target = torch.rand(rays.shape[0], 3, generator=torch.Generator(device="cpu").manual_seed(seed + 1)).to(device)
    return K, rays, target


def cpu_baseline(seconds_budget=25.0):
    """The CPU oracle's training step (same workload shape, B=256 rays) on this host's cores."""
    import _inputs as I
    from oracle import nerf_oracle as O
    # threads: the cores this process may run on, capped at 32 (ATen's intra-op parallelism stops scaling —
    # and with hundreds of threads on 256x256 GEMMs it collapses — well before that)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    ncores = max(1, min(avail, 32))
    torch.set_num_threads(ncores)
    Bc = 256
    sd = [O.as_tensors(I.nerf_state_dict(8, 256, 10, 4, 5, True, seed=s), True) for s in (21, 22)]
    net, cfg = O.NetCfg(8, 256, output_ch=5), O.RenderCfg(NC, NF, 1.0)
    rays = torch.from_numpy(I.ray_batch(Bc, seed=3, near=NEAR, far=FAR))
    target = torch.rand(Bc, 3)
    params = [p for d in sd for p in d.values()]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]

    def step(i):
        out = O.render_rays(rays, sd[0], sd[1], net, cfg, torch.rand(Bc, NC), torch.rand(Bc, NF))
        loss = O.mse(out["rgb_map"], target) + O.mse(out["rgb0"], target)
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        with torch.no_grad():
            for p, g, mm, vv in zip(params, grads, m, v):
                if g is not None:
                    O.adam_step(p, g, mm, vv, i + 1, 5e-4)
    tw = time.perf_counter()
    step(0)
    tw = time.perf_counter() - tw
    t0, n = time.perf_counter(), 0
    while n < 1 or (time.perf_counter() - t0 + tw < seconds_budget and n < 20):
        step(n + 1)
        n += 1
    dt = (time.perf_counter() - t0) / n
    # single-thread figure (SURVEY §8d asks for both): one warm + one timed step of 64 rays
    torch.set_num_threads(1)
    Bc1 = 64
    rays1, target1 = rays[:Bc1], target[:Bc1]

    def step1(i):
        out = O.render_rays(rays1, sd[0], sd[1], net, cfg, torch.rand(Bc1, NC), torch.rand(Bc1, NF))
        loss = O.mse(out["rgb_map"], target1) + O.mse(out["rgb0"], target1)
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        with torch.no_grad():
            for p, g, mm, vv in zip(params, grads, m, v):
                if g is not None:
                    O.adam_step(p, g, mm, vv, n + 2 + i, 5e-4)
    step1(0)
    t1 = time.perf_counter()
    step1(1)
    dt1 = time.perf_counter() - t1
    # all usable cores (SURVEY 8d asks for the figure even where it is slower: the 32-thread cap above is then evidence, not
    # assertion): ATen's intra-op parallelism collapses on these GEMM sizes with hundreds of threads, so the probe is ONE step
    # of the 64-ray batch (a 256-ray step takes over a minute on 256 threads)
    all_val = None
    if avail > ncores:
        torch.set_num_threads(avail)
        ta = time.perf_counter()
        step1(2)
        ta = time.perf_counter() - ta
        all_val = Bc1 * (NC + NC + NF) / ta
    torch.set_num_threads(ncores)
    return {"value": Bc * (NC + NC + NF) / dt, "unit": "ray-samples/s", "cores": ncores, "kind": "port",
            "single_thread_value": Bc1 * (NC + NC + NF) / dt1, "all_cores_value": all_val, "all_cores": avail,
            "sample": f"{n} training steps of {Bc} rays (same C2 shapes: 64+192 samples, D=8/W=256, fwd+bwd+Adam), "
                      f"{dt:.2f} s/step, torch {torch.__version__} CPU fp32, {ncores} threads of {avail} usable / "
                      f"{os.cpu_count()} logical cpus"}


def c5_leg(dev):
    """BASELINE configs[4]: one full-res LLFF frame (756x1008, NDC, 64 + 128 samples, D=8/W=256, perturb=0, chunk 32768)
    through render(c2w=...) incl. the frame's D2H (render_path does it per frame, R:158)."""
    import tempfile
    import _inputs as I
    from consistentnerf_amd import run_nerf as R
    H, W, focal = 756, 1008, 815.0
    a = make_args(tempfile.mkdtemp())
    a.dataset_type, a.no_ndc, a.raw_noise_std = "llff", False, 1.0
    torch.manual_seed(0)
    _, kw_test, *_ = R.create_nerf(a)
    kw_test.update(near=0.0, far=1.0)
    K = I.intrinsics(H, W, focal)
    poses = [torch.from_numpy(I.camera_pose(5.0 * i, 0.0, 4.0)) for i in range(2)]
    with torch.no_grad():
        R.render(H // 4, W // 4, K, chunk=32768, c2w=poses[0], **kw_test)      # warm-up (1/16 of a frame)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        rgb, disp, acc, extras = R.render(H, W, K, chunk=32768, c2w=poses[1], **kw_test)
        e1.record()
        rgb_host = rgb.cpu().numpy()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    n = H * W * (NC + NC + NF)
    tf = n * 2 * MAC_FWD / dt / 1e12
    # the same frame through the OPT-IN reduced-precision inference forward (bf16 matrix cores, 1 / 2 / 3 bf16 planes per
    # operand, fp32 accumulation; csrc/mlp_fwd_bf.hip): reported beside the exact render with its own dtype and its image
    # PSNR against it — never part of `value`, never the default path
    reduced = {}
    nets = [kw_test["network_fn"], kw_test["network_fine"]]
    try:
        for prec in ("bf16x3", "bf16x2", "bf16"):
            for m in nets:
                m.inference_precision = prec
            with torch.no_grad():
                R.render(H // 4, W // 4, K, chunk=32768, c2w=poses[0], **kw_test)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                rgb_r, *_ = R.render(H, W, K, chunk=32768, c2w=poses[1], **kw_test)
                rgb_r_host = rgb_r.cpu().numpy()
                torch.cuda.synchronize()
                dtr = time.perf_counter() - t1
            mse = float(np.mean((rgb_r_host.astype(np.float64) - rgb_host.astype(np.float64)) ** 2))
            reduced[prec] = {"frame_s": dtr, "speedup_vs_fp32": dt / dtr, "ray_samples_per_s": n / dtr,
                             "image_psnr_vs_fp32_render_dB": (None if mse == 0 else -10.0 * np.log10(mse)),
                             "dtype": {"bf16": "bf16 x bf16 -> f32", "bf16x2": "2 bf16 planes per operand, 3 cross terms -> f32",
                                       "bf16x3": "3 bf16 planes per operand, 6 cross terms -> f32"}[prec]}
    finally:
        for m in nets:
            m.inference_precision = "fp32"
    return {"frame_s": dt, "gpu_frame_s": e0.elapsed_time(e1) * 1e-3, "rays": H * W, "ray_samples_per_s": n / dt,
            "opt_in_reduced_precision": reduced,
            "frame": f"{H}x{W} NDC, chunk 32768, perturb 0, 64+128 samples, D=8 W=256 (random init), D2H of the frame included",
            "finite": bool(np.isfinite(rgb_host).all()),
            "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4), "basis": "whole frame, 2*593408 FLOP per ray-sample"}}


def c3_leg(dev, steps=20):
    """BASELINE configs[2]: LLFF-like 3-view training step WITH the consistency terms — hard masks from the cross-view depth
    warp (V:994-1046), masked rgb + depth losses on both levels (V:1645-1865), the monocular-depth patch term on 4 16x16
    patches (V:1678-1720), clip 0.1 + Adam (V:1983).  378x504 views, no_ndc, near 1.2 / far 12, 4096 random + 1024 patch
    rays per step."""
    import tempfile
    import _inputs as I
    from consistentnerf_amd import raybank as RB, run_nerf_view as V
    H, W, focal, near, far = 378, 504, 407.0, 1.2, 12.0
    a = make_args(tempfile.mkdtemp())
    a.dataset_type, a.stable_init = "llff", False
    torch.manual_seed(0)
    np.random.seed(0)
    kw_train, _, _, grad_vars, optimizer = V.create_nerf(a)
    kw_train.update(near=near, far=far)
    optimizer.param_groups[0]['clip_value'] = 0.1
    K = I.intrinsics(H, W, focal)
    poses = np.stack([I.camera_pose(th, -10.0, 4.0) for th in (0.0, 6.0, -6.0)])
    scene = [I.analytic_scene(H, W, K, p) for p in poses]
    depths = np.stack([s_[0] for s_ in scene]) + np.random.normal(0, 0.02, (3, H, W)).astype(np.float32)
    images = np.stack([s_[1] for s_ in scene])
    V.compute_hard_masks(H, W, K, poses, depths, [0, 1, 2], 0.1, device=dev)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    masks = V.compute_hard_masks(H, W, K, poses, depths, [0, 1, 2], 0.1, device=dev)
    torch.cuda.synchronize()
    t_masks = time.perf_counter() - t0
    mono = 1.0 / np.maximum(depths, 1e-3)
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
        il, dl = V.hardmask_losses(rgb, target, m, 0.2, depth, d_prior, far)
        il0, dl0 = V.hardmask_losses(extras['rgb0'], target, m, 0.2, extras['depth0'], d_prior, far)
        loss = il + il0 + 0.1 * (dl + dl0)
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
    n = B * (NC + NC + NF)
    tf = n * 2 * (MAC_FWD + MAC_DGRAD + MAC_WGRAD) / dt / 1e12
    return {"ms_per_step": dt * 1e3, "steps": steps, "rays_per_step": B, "ray_samples_per_s": n / dt,
            "hard_masks_3views_ms": t_masks * 1e3, "hard_mask_fraction": float(masks.mean()), "final_loss": float(loss.item()),
            "finite": bool(np.isfinite(loss.item())),
            "step": "3 LLFF-like 378x504 views, no_ndc; hard masks + masked rgb/depth losses on both levels + monocular patch "
                    "term (4 x 16x16) + clip 0.1 + Adam; 4096 random + 1024 patch rays, 64+128 samples, D=8 W=256",
            "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4), "basis": "whole step, 3489024 FLOP per ray-sample"}}


def pmc_traffic(kernel, points):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    each in their own `--pmc` run, scripts/gpu_pmc.sh -> profiles/r02_pmc/, r01_pmc_final/ as a fallback): KB units,
    FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md (16-byte-per-lane streaming reads are tallied at
    half).  A STATIC LOOKUP, not a measurement of this run (counters cannot be sampled from inside this process); None when
    no pass of that kernel at that launch size is on file."""
    import csv
    here = os.path.dirname(os.path.abspath(__file__))
    name = {"mlp_wgrad": "wgrad", "mlp_dgrad": "mlp_dgrad", "mlp_fwd_train": "mlp_fwd_train", "mlp_fwd": "mlp_fwd_inf"}.get(kernel)
    # grid threads of the launch: forward / dgrad run 64 threads per 32 points; wgrad's grid is (point ranges) x (GEMMs) x
    # 256 threads: 128 x 14 for the fine level alone (round 1), 128 x 28 for both levels in one grid (round 2)
    want = {"wgrad": {786432: 128 * 14 * 256, 1048576: 128 * 28 * 256}.get(points)}.get(name, 2 * points)
    for d in ("r02_pmc", "r01_pmc_final"):
        val = {}
        for f, ctr in (("pass2_summary.csv", "FETCH_SIZE"), ("pass3_summary.csv", "WRITE_SIZE")):
            path = os.path.join(here, "profiles", d, f)
            if not os.path.exists(path):
                break
            pick = [r for r in csv.DictReader(open(path)) if r["kernel"] == name and r["counter"] == ctr
                    and int(r["grid_threads"]) == want]
            if not pick:
                break
            val[ctr] = float(pick[0]["avg_per_launch"]) * 1024.0
        if len(val) == 2:
            return int(2 * val["FETCH_SIZE"] + val["WRITE_SIZE"])
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a timed region of ~5 s (200 x 26 ms), long enough for a coarse (seconds) GPU-busy sampler to see it
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the C5 / C3 legs that follow the timed C2 region at N=1")
    ap.add_argument("--graph", action="store_true",
                    help="(N=1) replay the step as ONE captured hipGraph (consistentnerf_amd/graph.py); the per-kernel table then "
                         "comes from a separate eager pass of the same steps, and the JSON says so")
    a = ap.parse_args()

    # stdout carries exactly ONE line, the JSON: everything else that writes to fd 1 — the reference-style prints of
    # create_nerf(), and RCCL's version banner, which the C library flushes at exit, i.e. AFTER the JSON — goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch.distributed as dist
    from consistentnerf_amd import distributed as D, ops
    from consistentnerf_amd import run_nerf as R
    # backend: RCCL ("nccl") for a real multi-GPU run.  CNERF_DIST_BACKEND=gloo lets the N>1 code path of this script (sharded
    # steps, GradReducer, barrier, max-over-ranks) be exercised on a box with fewer GPUs than ranks — the ranks then share
    # devices (rank % device_count) and the numbers mean nothing as a scaling measurement; the JSON's dist.backend says so.
    backend = os.environ.get("CNERF_DIST_BACKEND", "nccl")
    ngpu = torch.cuda.device_count()
    if backend == "nccl" and int(os.environ.get("LOCAL_RANK", "0")) >= max(ngpu, 1):
        raise SystemExit(f"LOCAL_RANK {os.environ.get('LOCAL_RANK')} but only {ngpu} GPU(s) visible: one rank per GPU over RCCL")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(ngpu, 1))
    rank, world, local = D.init_from_env(backend if (a.gpus > 1 or os.environ.get("CNERF_FORCE_DIST") == "1") else None)
    assert world == a.gpus, f"--gpus {a.gpus} but WORLD_SIZE={world}"
    local = local % max(ngpu, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist.is_initialized():   # build the RCCL communicator now (seconds), not inside the first step
        t = torch.zeros(1, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
    ok, name, cus, _ = ops.device_info(local)

    import tempfile
    torch.manual_seed(1234)                       # identical init on every rank (replicated weights)
    with tempfile.TemporaryDirectory() as tmp:
        kw_train, _, start, grad_vars, optimizer = R.create_nerf(make_args(tmp))
    kw_train.update(near=NEAR, far=FAR)
    K, bank, targets = build_ray_bank(dev)
    nbank = bank.shape[0]
    torch.manual_seed(99 + rank)                  # per-rank jitter streams (RegNeRF/train.py:364-365 precedent)
    gstep = B_PER_GPU * world
    # the step's gradient exchange: per-network slices of the flat fp32 gradient, all-reduced (RCCL) as _MlpFn.backward
    # reports them final; a no-op without a process group
    reducer = D.GradReducer(optimizer, [kw_train['network_fn'], kw_train['network_fine']], mean=True,
                            timing=dist.is_initialized())

    def body(rays, tgt):
        rays_od = torch.stack([rays[:, 0:3], rays[:, 3:6]], 0)
        rgb, disp, acc, extras = R.render(H_IMG, W_IMG, K, chunk=32768, rays=rays_od, retraw=True, **kw_train)
        optimizer.zero_grad()
        loss = R.img2mse(rgb, tgt) + R.img2mse(extras['rgb0'], tgt)
        loss.backward()
        reducer.finish()
        optimizer.step()
        return loss

    graphed = None
    if a.graph:
        assert world == 1, "--graph is a single-GPU option"
        from consistentnerf_amd.graph import GraphedStep
        graphed = GraphedStep(body, optimizer, (bank[0:B_PER_GPU], targets[0:B_PER_GPU]), warmup=3)

    def step(i, eager=False):
        lo = (i * gstep + rank * B_PER_GPU) % (nbank - B_PER_GPU)
        rays, tgt = bank[lo:lo + B_PER_GPU], targets[lo:lo + B_PER_GPU]
        loss = graphed(rays, tgt) if (graphed is not None and not eager) else body(rays, tgt)
        lr = 5e-4 * (0.1 ** (i / (250 * 1000)))
        for pg in optimizer.param_groups:
            pg['lr'] = lr
        return loss

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    ops.PROFILE = []
    reducer.exposed.clear()
    t0 = time.perf_counter()
    for i in range(a.steps):
        loss = step(a.warmup + i)
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    prof, ops.PROFILE = ops.PROFILE, None
    if graphed is not None:      # a replayed graph carries no events: the per-kernel table from an eager pass of the same steps
        ops.PROFILE = []
        for i in range(a.steps):
            step(a.warmup + a.steps + i, eager=True)
        torch.cuda.synchronize()
        prof, ops.PROFILE = ops.PROFILE, None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    final_loss = loss.item()

    # live per-kernel timing (HIP events on the launch stream, inside the timed region)
    kern = {}
    for nme, units, e0, e1 in prof:
        k = kern.setdefault((nme, units), [0.0, 0])
        k[0] += e0.elapsed_time(e1)
        k[1] += 1
    flops = {"mlp_fwd_train": 2 * MAC_FWD, "mlp_fwd": 2 * MAC_FWD, "mlp_dgrad": 2 * MAC_DGRAD, "mlp_wgrad": 2 * MAC_WGRAD}
    table = []
    for (nme, units), (ms, n) in kern.items():
        avg_ms = ms / n
        table.append({"kernel": nme, "points": units, "launches": n, "avg_ms": round(avg_ms, 4),
                      "tflops": round(flops[nme] * units / (avg_ms * 1e-3) / 1e12, 2),
                      "share_of_step": round(ms / (elapsed * 1e3), 4)})
    table.sort(key=lambda r: -r["avg_ms"] * r["launches"])
    dom = table[0]
    roofline = {"bound": "mfma", "kernel": f'{dom["kernel"]} (M={dom["points"]} points)', "achieved": dom["tflops"],
                "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(dom["tflops"] / PEAK_FP32_MFMA_TFLOPS, 4),
                "traffic": pmc_traffic(dom["kernel"], dom["points"]),
                "traffic_source": "static lookup: committed rocprofv3 PMC passes of this kernel at this launch size "
                                  "(profiles/*_pmc*/pass2+pass3 summaries, 2*FETCH_SIZE + WRITE_SIZE); NOT sampled in this run",
                "avg_launch_ms": dom["avg_ms"], "kernels": table}
    if graphed is not None:
        roofline["kernels_measured"] = "separate eager pass of the same steps after the timed (graph-replayed) region"
    dist_info = None
    if dist.is_initialized():
        ex = reducer.exposed_ms()
        dist_info = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(),
                     "messages_per_step": len(reducer.slices), "message_bytes": [4 * (hi - lo) for lo, hi in reducer.slices.values()],
                     "bytes_per_step": reducer.bytes_per_step,
                     "allreduce_exposed_ms": round(sum(ex) / max(len(ex), 1), 4), "allreduce_exposed_ms_max": round(max(ex), 4) if ex else None,
                     "measured": "HIP events on the launch stream: last slice issued -> launch stream released (this rank)"}

    if rank == 0:
        samples_per_step = B_PER_GPU * (NC + NC + NF) * world
        out = {
            "metric": "train_ray_samples_per_sec", "value": samples_per_step * a.steps / elapsed,
            "unit": "ray-samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "hip_graph": graphed is not None,
            "config": {"workload": "DTU scan8 3-view (synthetic 512x640 ray bank), 4096 rays/GPU/step, coarse 64 + "
                                   "fine 64+128 samples, D=8 W=256 viewdirs MLPs (random init), perturb=1, "
                                   "mse(rgb)+mse(rgb0), backward, Adam; BASELINE configs[1] (configs[3] when N>1)",
                       "rays_per_gpu": B_PER_GPU, "ray_samples_per_ray": NC + NC + NF, "device": name, "cus": cus,
                       "parallelism": f"ray-shard dp{world}" + (", RCCL all-reduce of the flat fp32 grad" if world > 1 else ""),
                       "final_loss": final_loss},
            "roofline": roofline,
        }
        if dist_info is not None:
            out["dist"] = dist_info
        if world == 1 and not a.no_extra:
            del kw_train, optimizer, grad_vars, bank, targets, reducer
            torch.cuda.empty_cache()
            out["extra"] = {"note": "same process, after the timed C2 region; not part of `value`", "c5": c5_leg(dev), "c3": c3_leg(dev)}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
