#!/usr/bin/env python3
"""bench.py — training ray-samples/sec of the MI355X NeRF hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--rays-per-gpu R] [--graph] [--pmc]
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one full optimisation step of BASELINE config C2 on synthetic DTU-like data, per GPU:
R rays -> coarse 64 samples -> fine 64+128 samples (D=8/W=256 MLPs with view directions, stratified
jitter, hierarchical resampling) -> mse(rgb)+mse(rgb0) -> backward (dgrad+wgrad of both nets) ->
[N>1: one RCCL all-reduce of the flat fp32 gradient, 1/N folded into the Adam kernel] -> Adam + lr decay.
256 ray-samples per ray (64 + 192 network evaluations).  Inputs (ray bank, targets, weights) are
resident in HBM before the timed region.

  --scaling weak   (default) R = 4096 rays per GPU, global batch 4096*N: BASELINE configs[1] at N=1.
  --scaling strong the 4096-ray C2 batch sharded N ways, R = 4096/N rays per GPU — BASELINE configs[3] (C4) as SURVEY §8(e)
                   defines it: same global batch as 1 GPU.
  --rays-per-gpu R times exactly that per-GPU shard (e.g. `--gpus 1 --rays-per-gpu 512` = the per-GPU work of the 8-way C4
                   shard on one GPU); `config.workload` names it.
  --graph          replay the step as captured hipGraphs (consistentnerf_amd/graph.py): one graph at N=1; at N>1 two graphs
                   around the eager gradient exchange (--graph-collective split, default) or the RCCL all-reduce recorded
                   inside one graph (--graph-collective capture).
  --pmc [auto|on|off]  re-runs this command twice under `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (separate passes, kernel-trace
                   only, 3 steps each) and fills roofline.traffic from THOSE runs instead of the committed lookup.  Default auto: on at
                   N=1 when the extra legs run and rocprofv3 is on PATH (adds ~40 s); any failure falls back to the lookup.

Extra objects on the JSON line:
  roofline     — dominant kernel of the step, algorithmic FLOPs per launch / its average duration measured
                 with HIP events on the launch stream inside the timed region; peak = fp32 MFMA 157.3 TFLOP/s.
  cpu_baseline — the CPU oracle ("port" of the reference step: stock ATen, fp32) timed on this host's cores on
                 a bounded sample (rank 0, N=1 only).
  dist         — what the gradient exchange did: ranks RCCL saw, backend, message count / bytes per step, and the part of the
                 exchange the step waited for (HIP events on the launch stream).  Present whenever a process group
                 exists (N>1, or CNERF_FORCE_DIST=1 on one GPU).
  extra        — (after the timed region, not part of `value`)
                 N=1: c4_shard = the 512-ray per-GPU step of the 8-way strong-scaling shard, eager and graphed, with its
                 per-kernel table; c5 = BASELINE configs[4]: render_path over 4 poses of the 60-pose LLFF spiral, 756x1008 NDC
                 frames (perturb=0, chunk 32768) incl. the D2H of every frame; c3 = configs[2]: 20 training steps with hard masks
                 + masked rgb/depth losses on both levels + the monocular patch term + clip 0.1 + Adam; hbm_kernels = the
                 HBM-bound kernels of the path (compositing fwd/bwd, resampling, losses, Adam, weight packing) against the 8 TB/s
                 roof at the C2 batch and at the C5 chunk.
                 N>1 (weak run): c4_strong = the same process group timing the 4096-ray batch sharded N ways, eager and
                 graphed (split), under a watchdog that prints the line without this leg if it does not finish.
                 --no-extra skips them.
"""
import argparse
import json
import os
import sys
import time

# multi-process GPU work on this pool needs dmabuf IPC (the host driver has no legacy IPC): RCCL's communicator setup fails with
# `hipIpcGetMemHandle: invalid argument` otherwise.  Set before the HIP runtime initialises; an explicit setting wins.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

MAC_FWD, MAC_DGRAD, MAC_WGRAD = 593408, 557696, 593408   # per ray-sample, D=8/W=256/viewdirs
PEAK_FP32_MFMA_TFLOPS = 157.3                              # MI355X_MICROARCH.md
PEAK_BF16_MFMA_TFLOPS = 2500.0                             # dense bf16 (MI355X_MICROARCH.md; no sparsity)
PEAK_BF16X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0           # fp32-equivalent FLOPs of a 3-plane split: 6 bf16 products each
B_PER_GPU, NC, NF = 4096, 64, 128
H_IMG, W_IMG, FOCAL, NEAR, FAR = 512, 640, 1446.0, 2.125, 4.67   # DTU-like

# ---- the stdout contract line -----------------------------------------------------------------------------------------------
# The driver parses ONE JSON line from stdout and keeps only so much of it: round 5's 20-27 KB line came back as `parsed: null`
# (VERDICT r05 item 1).  The line is therefore built by compact_line(): contract scalars, `config` (workload + flat leg_* scalars),
# `roofline` (scalars + a <= 4-row kernel table), `cpu_baseline` (scalars, short strings), `dist`; everything else — the legs'
# per-kernel tables, launch lists, hbm_kernels, PMC detail — goes to a side file whose path the line carries.  LINE_LIMIT is asserted
# by tests/test_host.py (worst case) and by a -m gpu test on the real line.
LINE_LIMIT = 6000
TOP_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "hip_graph", "error")
KERNEL_ROW_KEYS = ("kernel", "points", "launches", "avg_ms", "tflops", "frac", "share_of_step")


def _scalars(d, maxstr):
    """The scalar fields of a dict, strings cut to `maxstr` characters (nested objects are the side file's)."""
    o = {}
    for k, v in (d or {}).items():
        if isinstance(v, str):
            o[k] = v if len(v) <= maxstr else v[:maxstr - 3] + "..."
        elif isinstance(v, float):
            o[k] = v if abs(v) >= 1e6 or v != v else float(f"{v:.6g}")
        elif v is None or isinstance(v, (bool, int)):
            o[k] = v
    return o


def compact_line(out, detail_path=None, limit=LINE_LIMIT):
    """`out` (the full result object) -> the dict printed on stdout.  Deterministic, never raises, always <= `limit` bytes as JSON."""
    def build(maxstr, rows, with_route, with_pmc):
        line = {k: out[k] for k in TOP_KEYS if k in out}
        line["config"] = _scalars(out.get("config"), maxstr)
        if isinstance(out.get("config"), dict) and isinstance(out["config"].get("workload"), str):
            line["config"]["workload"] = out["config"]["workload"][:max(maxstr, 240)]
        rf = out.get("roofline")
        if isinstance(rf, dict):
            r = _scalars(rf, maxstr)
            if rows and isinstance(rf.get("kernels"), list):
                r["kernels"] = [{k: row.get(k) for k in KERNEL_ROW_KEYS if k in row} for row in rf["kernels"][:rows]]
            if with_pmc and isinstance(rf.get("pmc"), dict):
                r["pmc"] = _scalars(rf["pmc"], 60)
            line["roofline"] = r
        if isinstance(out.get("cpu_baseline"), dict):
            line["cpu_baseline"] = _scalars(out["cpu_baseline"], maxstr)
        if isinstance(out.get("dist"), dict):
            line["dist"] = _scalars(out["dist"], maxstr)
            sb = out["dist"].get("slice_bytes")
            if isinstance(sb, list):
                line["dist"]["slice_bytes"] = sb[:8]
        if with_route and isinstance(out.get("route"), dict):
            line["route"] = _scalars(out["route"], 40)
        if detail_path:
            line["detail"] = detail_path
        return line
    for maxstr, rows, with_route, with_pmc in ((200, 4, True, True), (120, 4, True, False), (80, 4, False, False),
                                               (48, 2, False, False), (24, 0, False, False)):
        line = build(maxstr, rows, with_route, with_pmc)
        if len(json.dumps(line)) <= limit:
            return line
    # last resort (only reachable with hundreds of scalar keys): the contract scalars and the three required objects' numbers
    line = {k: out[k] for k in TOP_KEYS if k in out}
    for k in ("roofline", "cpu_baseline"):
        if isinstance(out.get(k), dict):
            line[k] = {kk: vv for kk, vv in _scalars(out[k], 16).items()
                       if kk in ("bound", "achieved", "peak", "unit", "frac", "traffic", "value", "cores", "kind", "sample")}
    line["config"] = {"workload": str((out.get("config") or {}).get("workload", ""))[:200]}
    return line


def write_detail(out, world):
    """The full result object -> gpurun_out/bench_detail[_<N>gpus].json (directory created; relative to the cwd, which is the repo
    root for the driver and for gpurun) and one line on stderr.  Returns the path, or None when the file could not be written."""
    blob = json.dumps(out)
    sys.stderr.write("[bench detail] " + blob + "\n")
    path = os.path.join("gpurun_out", "bench_detail.json" if world == 1 else f"bench_detail_{world}gpus.json")
    try:
        os.makedirs("gpurun_out", exist_ok=True)
        with open(path, "w") as f:
            f.write(blob + "\n")
        return path
    except OSError:
        return None


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


def physical_cores():
    """(physical cores this process may run on, usable logical cpus): distinct (package, core id) pairs of the cpus in the
    affinity mask, from /proc/cpuinfo; falls back to the logical count."""
    try:
        avail = sorted(os.sched_getaffinity(0))
    except AttributeError:
        avail = list(range(os.cpu_count() or 1))
    try:
        cores, cur = {}, {}
        for line in open("/proc/cpuinfo"):
            if ":" in line:
                k, v = [x.strip() for x in line.split(":", 1)]
                cur[k] = v
            elif cur:
                if "processor" in cur:
                    cores[int(cur["processor"])] = (cur.get("physical id", "0"), cur.get("core id", cur["processor"]))
                cur = {}
        if cur and "processor" in cur:
            cores[int(cur["processor"])] = (cur.get("physical id", "0"), cur.get("core id", cur["processor"]))
        phys = len({cores[c] for c in avail if c in cores})
        return (phys or len(avail)), len(avail)
    except OSError:
        return len(avail), len(avail)


def cpu_baseline(seconds_budget=25.0):
    """The CPU oracle's training step (the C2 shapes: 64+192 samples, D=8/W=256, fwd+bwd+Adam) on this host's cores.
    Headline figure: B = 1024 rays on min(physical cores, 32) threads — ATen's intra-op parallelism stops
    scaling on 256x256 GEMMs well before 32 threads; a second figure at threads = ALL physical cores (count printed) is
    measured on a bounded probe and, when it is not slower, on the full batch; a single-thread figure on 64 rays."""
    import _inputs as I
    from oracle import nerf_oracle as O
    phys, avail = physical_cores()
    ncores = max(1, min(phys, 32))
    sd = [O.as_tensors(I.nerf_state_dict(8, 256, 10, 4, 5, True, seed=s), True) for s in (21, 22)]
    net, cfg = O.NetCfg(8, 256, output_ch=5), O.RenderCfg(NC, NF, 1.0)
    rays_all = torch.from_numpy(I.ray_batch(1024, seed=3, near=NEAR, far=FAR))
    target_all = torch.rand(1024, 3)
    params = [p for d in sd for p in d.values()]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    count = [0]

    def step(Bc):
        count[0] += 1
        out = O.render_rays(rays_all[:Bc], sd[0], sd[1], net, cfg, torch.rand(Bc, NC), torch.rand(Bc, NF))
        loss = O.mse(out["rgb_map"], target_all[:Bc]) + O.mse(out["rgb0"], target_all[:Bc])
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        with torch.no_grad():
            for p, g, mm, vv in zip(params, grads, m, v):
                if g is not None:
                    O.adam_step(p, g, mm, vv, count[0], 5e-4)

    def timed(Bc, threads, budget, max_steps=20):
        torch.set_num_threads(threads)
        t0 = time.perf_counter()
        step(Bc)                               # warm-up (also the bound on what the timed steps will cost)
        tw = time.perf_counter() - t0
        if tw > budget:                        # one step already blew the budget: report it rather than spend more
            return Bc * (NC + NC + NF) / tw, 0, tw         # n = 0: only the cold step ran
        t0, n = time.perf_counter(), 0
        while n < 1 or (time.perf_counter() - t0 + tw < budget and n < max_steps):
            step(Bc)
            n += 1
        dt = (time.perf_counter() - t0) / n
        return Bc * (NC + NC + NF) / dt, n, dt

    # 256-ray probe decides whether the 1024-ray batch fits the budget (3.4 s/step expected at 32 threads)
    v256, _, dt256 = timed(256, ncores, 6.0, max_steps=2)
    Bc = 1024 if 4 * dt256 * 3 < seconds_budget else 256
    val, n, dt = timed(Bc, ncores, seconds_budget - 8.0)
    # all physical cores (SURVEY 8d "N = all physical cores"): the same protocol as the headline figure — one warm-up step, then
    # >= 3 timed steps of 256 rays inside a 12 s budget (round 4 reported one COLD 32-ray step here, which measured thread start-up)
    phys_val = phys_sample = None
    if phys > ncores:
        pv, pn, pdt = timed(256, phys, 12.0, max_steps=3)
        phys_val = pv
        phys_sample = (f"{pn} warm training steps of 256 rays, {pdt:.2f} s/step on {phys} threads" if pn > 0 else
                       f"ONE cold 256-ray step ({pdt:.2f} s: over the 12 s budget, no warm step was affordable)")
    v1, _, dt1 = timed(64, 1, 0.0)
    torch.set_num_threads(ncores)
    # forward only (the C5 shapes: perturb 0, no gradient): 2048 rays, one warm + one timed call (SURVEY 8d asks for both views)
    cfg_inf = O.RenderCfg(NC, NF, 0.0)
    with torch.no_grad():
        O.render_rays(rays_all[:512], sd[0], sd[1], net, cfg_inf)
        t0 = time.perf_counter()
        for c in range(0, 2048, 1024):
            O.render_rays(rays_all[:1024], sd[0], sd[1], net, cfg_inf)
        inf_val = 2048 * (NC + NC + NF) / (time.perf_counter() - t0)
    best, cores = (val, ncores) if (phys_val is None or val >= phys_val) else (phys_val, phys)
    return {"value": best, "unit": "ray-samples/s", "cores": cores, "kind": "port",
            "threads_main": ncores, "value_main": val, "physical_cores": phys, "value_physical_cores": phys_val,
            "value_all_physical_cores": phys_val, "value_32_threads": val,
            "sample_physical_cores": phys_sample, "single_thread_value": v1, "logical_cpus_usable": avail,
            "inference_value": inf_val, "inference_sample": f"forward only, 2 x 1024 rays, {ncores} threads (the C5 shapes)",
            "sample": f"{max(n, 1)} {'warm' if n else 'COLD'} training steps of {Bc} rays (same C2 shapes: 64+192 samples, D=8/W=256, fwd+bwd+Adam), "
                      f"{dt:.2f} s/step on {ncores} threads, torch {torch.__version__} CPU fp32; host has {phys} physical cores "
                      f"({avail} usable logical cpus of {os.cpu_count()}); `value` = the better of the {ncores}-thread and the "
                      f"{phys}-thread figure"}


def llff_rig(n_views=6, seed=17):
    """A seeded forward-facing rig of `n_views` cameras in the raw LLFF poses_bounds layout [N, 17] (load_llff.py:64-66)."""
    rs = np.random.RandomState(seed)
    arr = np.zeros((n_views, 17))
    for k in range(n_views):
        a, b = rs.uniform(-0.15, 0.15, 2)
        Rm = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]]) @ \
            np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
        t = rs.uniform(-1.5, 1.5, 3) * [1, 0.6, 0.2]
        arr[k, :15] = np.concatenate([Rm[:, [1, 0, 2]] * [1, -1, 1], t[:, None], np.array([[3024.], [4032.], [3260.]])], 1).reshape(-1)
        arr[k, 15:] = [rs.uniform(8, 12), rs.uniform(60, 110)]
    return arr


def c5_leg(dev, n_frames=4):
    """BASELINE configs[4]: LLFF 6-view full-res `render_path` (R:140-178) — `n_frames` poses of the 60-pose spiral
    (load_llff.py:178-202) of a 6-view rig, 756x1008 NDC frames, 64 + 128 samples, D=8/W=256, perturb=0, chunk 32768, every
    frame handed to the host as the reference does (R:157-159)."""
    import contextlib
    import tempfile
    from consistentnerf_amd import io_formats as F, run_nerf as R
    H, W = 756, 1008
    poses, bds, render_poses, i_test = F.llff_poses(llff_rig(6), (H, W), factor=4, n_render=60)
    focal = float(poses[0, 2, 4])                                   # 3260 / 4 = 815
    K = np.array([[focal, 0, 0.5 * W], [0, focal, 0.5 * H], [0, 0, 1]], dtype=np.float32)
    a = make_args(tempfile.mkdtemp())
    a.dataset_type, a.no_ndc, a.raw_noise_std = "llff", False, 1.0
    torch.manual_seed(0)
    _, kw_test, *_ = R.create_nerf(a)
    kw_test.update(near=0.0, far=1.0)
    sel = [int(round(k * 60 / n_frames)) for k in range(n_frames)]   # spread over the spiral
    rp = torch.from_numpy(render_poses[sel]).to(dev)
    with torch.no_grad(), contextlib.redirect_stdout(sys.stderr):
        R.render(H // 4, W // 4, K, chunk=32768, c2w=rp[0], **kw_test)      # warm-up (1/16 of a frame)
        torch.cuda.synchronize()
        # per-frame host time: render_path prints it (R:147-149); measured here around single-pose calls of the same function
        frame_s = []
        t_all = time.perf_counter()
        rgbs, disps = [], []
        for k in range(n_frames):
            t0 = time.perf_counter()
            r_, d_ = R.render_path(rp[k:k + 1], (H, W, focal), K, 32768, kw_test)
            frame_s.append(time.perf_counter() - t0)
            rgbs.append(r_[0]); disps.append(d_[0])
        dt_all = time.perf_counter() - t_all
        # all frames in ONE render_path call, and through the sharded driver (world 1 here: in-kernel rays of the row block,
        # D2H of frame i under the render of frame i+1)
        t0 = time.perf_counter()
        rgbs2, _ = R.render_path(rp, (H, W, focal), K, 32768, kw_test)
        dt_path = time.perf_counter() - t0
        from consistentnerf_amd import distributed as D
        t0 = time.perf_counter()
        rgbs3, _ = D.render_path_sharded(rp, (H, W, focal), K, 32768, kw_test)
        dt_sharded = time.perf_counter() - t0
    rgb_host = np.stack(rgbs, 0)
    same = bool(np.array_equal(rgb_host, rgbs2) and np.array_equal(rgb_host, rgbs3))
    dt = float(np.median(frame_s))
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
                R.render(H // 4, W // 4, K, chunk=32768, c2w=rp[0], **kw_test)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                rgb_r, *_ = R.render(H, W, K, chunk=32768, c2w=rp[1], **kw_test)
                rgb_r_host = rgb_r.cpu().numpy()
                torch.cuda.synchronize()
                dtr = time.perf_counter() - t1
            mse = float(np.mean((rgb_r_host.astype(np.float64) - rgb_host[1].astype(np.float64)) ** 2))
            reduced[prec] = {"frame_s": dtr, "speedup_vs_fp32": dt / dtr, "ray_samples_per_s": n / dtr,
                             "image_psnr_vs_fp32_render_dB": (None if mse == 0 else -10.0 * np.log10(mse)),
                             "dtype": {"bf16": "bf16 x bf16 -> f32", "bf16x2": "2 bf16 planes per operand, 3 cross terms -> f32",
                                       "bf16x3": "3 bf16 planes per operand, 6 cross terms -> f32"}[prec]}
    finally:
        for m in nets:
            m.inference_precision = "fp32"
    return {"frame_s": dt, "frames": n_frames, "frame_s_each": [round(x, 4) for x in frame_s], "spiral_pose_indices": sel,
            "render_path_s": dt_path, "render_path_sharded_s": dt_sharded, "identical_frames_across_drivers": same,
            "rays": H * W, "ray_samples_per_s": n / dt, "ray_samples_per_s_render_path": n * n_frames / dt_path,
            "opt_in_reduced_precision": reduced,
            "frame": f"{H}x{W} NDC, chunk 32768, perturb 0, 64+128 samples, D=8 W=256 (random init), spiral poses of a 6-view LLFF "
                     f"rig, D2H of every frame included",
            "finite": bool(np.isfinite(rgb_host).all()),
            "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4), "basis": "median frame, 2*593408 FLOP per ray-sample"}}


HBM_PEAK_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E ~8 TB/s


def hbm_kernels(dev, reps=20):
    """The HBM-bound kernels of the path, each timed alone with HIP events (`reps` back-to-back launches after a warm-up) on
    synthetic inputs of the C2 batch (4096 rays) and of the C5 chunk (32768 rays): algorithmic bytes (SURVEY §8d per-unit
    figures x units) / average launch time against the 8 TB/s roof."""
    from consistentnerf_amd import ops
    from consistentnerf_amd.run_nerf_helpers import NeRF
    rows = []

    def timeit(fn):
        """`reps` launches recorded into ONE hipGraph and replayed: what the events bracket is GPU time of back-to-back kernels
        (an eager loop of 10-us kernels measures the host's launch rate instead)"""
        fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        g_ = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g_):
            for _ in range(reps):
                fn()
        g_.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g_.replay()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    floor = [None]

    def add(kernel, size, nbytes, ms, basis):
        gbps = nbytes / (ms * 1e-3) / 1e9
        # the bound of a launch this small is not the HBM roof but the back-to-back launch interval: bytes / max(bytes / 8 TB/s, floor)
        bound_ms = max(nbytes / (HBM_PEAK_GBPS * 1e9) * 1e3, floor[0])
        rows.append({"kernel": kernel, "at": size, "bytes_algorithmic": int(nbytes), "avg_ms": round(ms, 5), "GBps": round(gbps, 1),
                     "frac_of_8TBps": round(gbps / HBM_PEAK_GBPS, 4), "bound_ms_hbm_or_launch_floor": round(bound_ms, 5),
                     "frac_of_that_bound": round(bound_ms / ms, 4), "basis": basis})

    import ctypes as C
    from consistentnerf_amd import _lib
    lib, P = _lib.load(), ops._p
    st = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
    # launch floor: the same harness around a kernel that does nothing measurable (one workgroup writing one float)
    one = torch.empty(1, device=dev)
    rng1 = _lib.Rng(1, 0, None, 0)
    floor[0] = timeit(lambda: lib.cnerf_uniform_rng(C.byref(rng1), 1, 1, P(one), st()))
    g = torch.Generator(device="cpu").manual_seed(5)
    for B, tag in ((4096, "C2 batch, 4096 rays"), (32768, "C5 chunk, 32768 rays")):
        # DTU-like rays and densities: unit-ish directions, near / far of the scene, sigma ~ N(0, 3) (a random-init network's)
        rays = torch.randn(B, 11, generator=g)
        rays[:, 3:6] /= rays[:, 3:6].norm(dim=-1, keepdim=True)
        rays[:, 6], rays[:, 7] = NEAR, FAR
        rays = rays.to(dev)
        for S in (NC, NC + NF):
            raw = (torch.randn(B, S, 4, generator=g) * 3.0).to(dev)
            z = torch.sort(torch.rand(B, S, generator=g) * (FAR - NEAR) + NEAR, -1)[0].to(dev)
            # outputs allocated once: only the kernel sits between the events (C ABI called directly)
            rgb, disp, acc, depth = torch.empty(B, 3, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev), torch.empty(B, device=dev)
            wts, d_raw = torch.empty(B, S, device=dev), torch.empty(B, S, 4, device=dev)
            ms = timeit(lambda: lib.cnerf_composite_fwd(P(raw), 4, P(z), P(rays), 11, None, B, S, 0, P(rgb), P(disp), P(acc), P(depth), P(wts), st()))
            add("composite_fwd_k", f"{tag}, S={S}", B * S * 24 + B * (44 + 28), ms, "24 B per ray-sample (raw 16 + z 4 in, weights 4 out) + 72 B per ray")
            tgt_, loss_ = torch.rand(B, 3, generator=g).to(dev), torch.empty(1, device=dev)
            ws_ = torch.empty(lib.cnerf_composite_mse_ws_floats(B) // 2, device=dev, dtype=torch.float64)
            ctr_ = torch.zeros(int(lib.cnerf_composite_mse_counter_words()), device=dev, dtype=torch.int32)
            ms = timeit(lambda: lib.cnerf_composite_fwd_mse(P(raw), 4, P(z), P(rays), 11, None, B, S, 0, P(tgt_), None, P(rgb), P(disp), P(acc),
                                                            P(depth), P(wts), P(loss_), P(ws_), P(ctr_), st()))
            add("composite_fwd_k + img2mse (render_loss)", f"{tag}, S={S}", B * S * 24 + B * (44 + 28 + 12), ms,
                "as composite_fwd_k + 12 B per ray of target; the loss leaves the same launch (last-ticket workgroup sums the partials)")
            ms = timeit(lambda: lib.cnerf_composite_bwd_mse(P(raw), 4, P(z), P(rays), 11, None, B, S, 0, P(rgb), P(tgt_), None, P(d_raw), st()))
            add("composite_bwd_k + img2mse seed (render_loss)", f"{tag}, S={S}", B * S * 36 + B * (44 + 24), ms,
                "as composite_bwd_k with the seed formed from rgb_map and target (24 B per ray) instead of read (12 B)")
            gr, gd = torch.randn(B, 3, generator=g).to(dev), torch.randn(B, generator=g).to(dev)
            ms = timeit(lambda: lib.cnerf_composite_bwd(P(raw), 4, P(z), P(rays), 11, None, B, S, 0, P(gr), None, None, P(gd), P(d_raw), st()))
            add("composite_bwd_k", f"{tag}, S={S}", B * S * 36 + B * (44 + 16), ms, "36 B per ray-sample (raw 16 + z 4 in, d_raw 16 out) + 60 B per ray")
        zc = torch.sort(torch.rand(B, NC, generator=g) * (FAR - NEAR) + NEAR, -1)[0].to(dev)
        w = (torch.rand(B, NC, generator=g) ** 8).to(dev)
        u = torch.rand(B, NF, generator=g).to(dev)
        zf, zs = torch.empty(B, NC + NF, device=dev), torch.empty(B, device=dev)
        ms = timeit(lambda: lib.cnerf_resample(P(zc), P(w), P(u), NF, B, NC, NF, P(zf), P(zs), None, None, st()))
        add("resample_k", tag, B * 4 * (NC + NC + NF + NC + NF + 1), ms, "per ray: z 64 + weights 64 + u 128 floats in, z_fine 192 + z_std out = 1796 B")
        x, y = torch.rand(B, 3, generator=g).to(dev), torch.rand(B, 3, generator=g).to(dev)
        ms = timeit(lambda: ops.mse(x, y))
        add("mse_k", tag, B * 3 * 12, ms, "12 B per element (x, y in, d_x out); one workgroup, fixed order")
        d, pr, mk = torch.rand(B, generator=g).to(dev), torch.rand(B, generator=g).to(dev), (torch.rand(B, generator=g) < 0.6).float().to(dev)
        ms = timeit(lambda: ops.masked_loss(x, y, d, pr, mk, FAR, 0.2))
        add("masked_loss_k", tag, B * (24 + 12 + 12 + 4), ms, "52 B per ray (rgb, target, depth, prior, mask in; d_rgb, d_depth out)")
    m = NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True).to(dev)
    nparam = sum(p.numel() for p in m.kernel_tensors())
    n2 = 2 * nparam
    p_, g_, m_, v_ = (torch.rand(n2, device=dev) for _ in range(4))
    ms = timeit(lambda: ops.adam_step(p_, g_, m_, v_, 3, 5e-4))
    add("adam_k", f"{n2} parameters (coarse + fine)", n2 * 28, ms, "28 B per parameter (p, m, v read + written, g read)")
    spec = m.spec()
    packed = ops.pack_weights(spec, m.kernel_tensors())
    ms = timeit(lambda: ops.pack_weights(spec, m.kernel_tensors(), packed))
    add("pack_weights (per network)", f"{nparam} parameters -> {packed.numel()} packed floats", 4 * (nparam + packed.numel()), ms,
        "parameters read once, forward + transposed panels written")
    return {"peak_GBps": HBM_PEAK_GBPS, "reps": reps, "launch_floor_ms": round(floor[0], 5),
            "launch_floor": "back-to-back interval of a one-workgroup kernel that writes one float, same harness: no kernel can show less; "
                            "`frac_of_that_bound` = max(bytes / 8 TB/s, floor) / measured",
            "note": "each kernel alone: `reps` launches recorded in one hipGraph, HIP events around its "
            "replay (back-to-back GPU time incl. the inter-kernel gap, no host launch cost); the C2-batch working sets (7-30 MB) sit in "
            "the 256 MB Infinity Cache between launches, so those rows are cache-resident rates; these kernels are 0.3 % of a C2 step "
            "(5 % of the 512-ray C4-shard step) — the table says how far from the HBM roof they sit at both sizes",
            "kernels": rows}


LOSS_ENTRY = "render_loss"      # (main() records --loss-entry here for the passes this command spawns)


def pmc_rerun(per_rank, dom_kernel):
    """--pmc: this benchmark re-run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (one counter per pass, kernel
    trace only — MI355X_MICROARCH.md's HBM recipe; counters cannot be sampled from inside the measuring process), a few steps
    each; -> per-launch HBM bytes of the dominant kernel = 2 x FETCH_SIZE (gfx950: 16-byte-per-lane streaming reads are tallied
    at half) + WRITE_SIZE, both in KB units."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    pat = {"mlp_wgrad": "wgrad_k", "mlp_dgrad": "mlp_dgrad_k", "mlp_fwd_train": "mlp_fwd_k", "mlp_fwd": "mlp_fwd_k"}[dom_kernel]
    if shutil.which("rocprofv3") is None:
        return {"traffic": None, "source": "rocprofv3 not on PATH"}
    res, launches = {}, {}
    dispatches = {"FETCH_SIZE": {}, "WRITE_SIZE": {}}
    env = dict(os.environ, TMPDIR="/tmp")
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="cnerf_pmc_")
        try:
            cmd = ["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "3", "--warmup", "2", "--no-extra", "--no-cpu-baseline",
                   "--rays-per-gpu", str(per_rank), "--loss-entry", LOSS_ENTRY]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=180, env=env, cwd="/tmp")
            except subprocess.TimeoutExpired:
                return {"traffic": None, "source": f"rocprofv3 --pmc {ctr} timed out"}
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if r.returncode != 0 or not files:
                return {"traffic": None, "source": f"rocprofv3 --pmc {ctr} rc={r.returncode}: {r.stderr[-300:]}"}
            vals = {}
            for f in files:
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] == ctr:
                            dispatches[ctr][int(row["Dispatch_Id"])] = row["Kernel_Name"]
                        if pat in row["Kernel_Name"] and "reduce" not in row["Kernel_Name"] and row["Counter_Name"] == ctr:
                            vals.setdefault(row["Dispatch_Id"], 0.0)
                            vals[row["Dispatch_Id"]] += float(row["Counter_Value"])
            if not vals:
                return {"traffic": None, "source": f"no {pat} dispatch in the {ctr} pass"}
            v = sorted(vals.values())
            big = [x for x in v if x >= 0.5 * v[-1]]          # the largest launch size of that kernel (fine / merged level)
            res[ctr], launches[ctr] = sum(big) / len(big) * 1024.0, len(big)
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"launches": launches_per_step(dispatches["FETCH_SIZE"]),
            "traffic": int(2 * res["FETCH_SIZE"] + res["WRITE_SIZE"]), "FETCH_SIZE_bytes_x2": int(2 * res["FETCH_SIZE"]),
            "FETCH_SIZE_bytes_raw": int(res["FETCH_SIZE"]), "WRITE_SIZE_bytes": int(res["WRITE_SIZE"]), "launches_averaged": launches,
            "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes of THIS command (3 steps each), per launch of the "
                      "dominant kernel: 2*FETCH_SIZE + WRITE_SIZE"}


def bf16x3_leg(dev, rank, world, per_rank, f32_ms, steps=40, warmup=10):
    """The SAME C2 training step with the opt-in "bf16x3" training arithmetic (consistentnerf_amd/run_nerf.py::training_precision):
    the MLP GEMMs that have a bf16x3 kernel run on the bf16 matrix cores with three bf16 planes per operand (6 cross terms, fp32
    accumulation: fp32-equivalent, the parity suite passes at the fp32 tolerances in this mode); everything else — and every GEMM
    without such a kernel yet — is the exact-fp32 path.  A SECOND line: never `value`, never `dtype` of the headline."""
    wl = Workload(dev, rank, world, precision="bf16x3")
    wl.set_sharding(per_rank, strong=False)
    el, loss, prof = wl.run(per_rank, steps, warmup)
    table = per_kernel_table(prof, el * 1e3)
    ms = el / steps * 1e3
    n = per_rank * (NC + NC + NF)
    tf = n * 2 * (MAC_FWD + MAC_DGRAD + MAC_WGRAD) / (ms * 1e-3) / 1e12
    which = sorted({r["kernel"] for r in table})
    return {"ms_per_step": round(ms, 4), "ray_samples_per_s": n / (ms * 1e-3), "speedup_vs_f32_step": round(f32_ms / ms, 4),
            "steps": steps, "warmup": warmup, "final_loss": float(loss.item()),
            "dtype": "bf16x3: 3 bf16 planes per operand, 6 cross terms, f32 accumulate (where a bf16x3 kernel exists), else f32",
            "kernels_in_bf16x3": [k for k in which if k.endswith("_bf3")], "kernels_in_f32": [k for k in which if not k.endswith("_bf3")],
            "roofline": {"bound": "mfma", "achieved": round(tf, 2), "unit": "TFLOP/s (fp32-equivalent, whole step)",
                         "peak": round(PEAK_BF16X3_TFLOPS, 1), "frac": round(tf / PEAK_BF16X3_TFLOPS, 4),
                         "basis": "2.5 PFLOP/s dense bf16 / 6 products per fp32-equivalent MAC", "kernels": table}}


def route_of(table, steps):
    """Which autograd route the timed steps took (consistentnerf_amd/run_nerf.py): "merged" = ONE dgrad + ONE wgrad launch per step
    for both networks (the engine query `torch._C._will_engine_execute_node` is usable and CNERF_MERGE_BWD != 0), "plain" = one
    backward per level.  Read off the launch counts of the timed region, with what the module selected at import beside it."""
    from consistentnerf_amd import run_nerf as R
    wg = [r for r in table if r["kernel"] == "mlp_wgrad"]
    per_step = sum(r["launches"] for r in wg) / max(steps, 1)
    return {"backward": "merged" if abs(per_step - 1.0) < 1e-9 else "plain", "wgrad_launches_per_step": per_step,
            "engine_query_usable": R._ENGINE_QUERY is not None, "merge_bwd_enabled": bool(R.MERGE_BWD),
            "direct_accumulation_into_flat_grad": R._ENGINE_QUERY is not None}


def launches_per_step(dispatches):
    """Kernel launches of ONE training step, from the ordered dispatch list of a rocprofv3 pass over this command: every step ends
    with exactly one `adam_k`, so the dispatches between two consecutive ones are one step's.  -> {"total", "own" (kernels of
    libcnerf_hip.so), "aten" (torch glue: RNG, fills, copies, cat ...), "kernels": {short name: count}} of the last full step."""
    ids = sorted(dispatches)
    names = [dispatches[i] for i in ids]
    ends = [k for k, n in enumerate(names) if "adam_k" in n]
    if len(ends) < 2:
        return None
    step = names[ends[-2] + 1:ends[-1] + 1]
    own = [n for n in step if "anonymous namespace)::" in n and "at::native" not in n]
    hist = {}
    for n in step:
        short = n.split("(anonymous namespace)::")[1].split("(")[0] if n in own else (
            "aten:" + n.split("at::native::")[-1].split("<")[0].split("(")[0] if "at::native" in n else n.split("(")[0])
        hist[short] = hist.get(short, 0) + 1
    return {"total": len(step), "own": len(own), "aten_and_runtime": len(step) - len(own), "kernels": hist,
            "source": "ordered dispatch list of the rocprofv3 pass of THIS command; one step = the dispatches between two adam_k"}


def c3_scene(dev):
    """The C3 leg's LLFF-like rig: 3 analytic 378x504 views, noisy depth priors, hard masks (V:994-1046), monocular priors."""
    import tempfile
    import _inputs as I
    from consistentnerf_amd import run_nerf_view as V
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
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(dev)  # noqa: E731
    # the in-loop consistency leg compares UN-normalised depths (VT:933, VT:950: no / far): the analytic scene's background depth of
    # ~1417 (rays that miss the geometry) would make its loss a sum of (1417 - d)^2 outliers (~1e4: round 5's `final_loss`), so that
    # leg reads priors bounded by the far plane
    dep_ss = np.minimum(depths, far)
    return dict(H=H, W=W, K=K, near=near, far=far, poses=poses, kw=kw_train, opt=optimizer, images=images, depths=depths, masks=masks,
                dep_ss_t=[t(dep_ss[i]) for i in range(3)],
                img_t=[t(images[i]) for i in range(3)], dep_t=[t(depths[i]) for i in range(3)],
                msk_t=[t(masks[i]) for i in range(3)], mono_t=[t(mono[i]) for i in range(3)], t_masks=t_masks)


def c3_step_fn(sc, route="render_loss"):
    """One C3 training step (V:1452-1517 batch, V:1636-1865 loss, V:1982-1994 tail) on the rig of c3_scene().
    route "render_loss": the round-5 surface — raybank.sample_patch_rays (ONE sampling launch that also writes the packed ray rows),
    run_nerf_view.render_loss (every loss term in the compositing launches + one tail launch), run_nerf.backward;
    route "reference_lines": the same step as the reference's statements (round 4's form: separate loss launches + ATen glue)."""
    from consistentnerf_amd import raybank as RB, run_nerf as R, run_nerf_view as V
    H, W, K, far, kw, opt = sc["H"], sc["W"], sc["K"], sc["far"], sc["kw"], sc["opt"]

    def step(i):
        v = i % 3
        starts = RB.draw_patch_starts(H, W, 4, 16)
        rays, target, sel, (d_prior, m, mono_s) = RB.sample_patch_rays(
            sc["img_t"][v], sc["poses"][v], H, W, K, 4096, starts, extras=(sc["dep_t"][v], sc["msk_t"][v], sc["mono_t"][v]),
            render_kwargs=kw)
        if route == "render_loss":
            loss = V.render_loss(H, W, K, target, mask=m, depth_prior=d_prior, chunk=32768, rays=rays, hardmask_coef=0.2,
                                 depth_w=0.1, mono=mono_s, patch_num=4, patch_size=16, patch_w=0.001, retraw=True, **kw)[0]
            opt.zero_grad()
            R.backward(loss)
        else:
            rgb, disp, acc, depth, extras = V.render(H, W, K, chunk=32768, rays=rays, retraw=True, **kw)
            opt.zero_grad()
            il, dl = V.hardmask_losses(rgb, target, m, 0.2, depth, d_prior, far)
            il0, dl0 = V.hardmask_losses(extras['rgb0'], target, m, 0.2, extras['depth0'], d_prior, far)
            loss = il + il0 + 0.1 * (dl + dl0)
            loss = loss + 0.001 * (V.midas_patch_loss(depth, mono_s, 4, 16) + V.midas_patch_loss(extras['depth0'], mono_s, 4, 16))
            loss.backward()
        opt.step()
        for pg in opt.param_groups:
            pg['lr'] = 5e-4 * (0.1 ** (i / 250000))
        return loss
    return step


def c3_ss_step_fn(sc, route="ss_step_loss"):
    """The in-loop consistency variant of the step (run_nerf_view_test.py VT:895-972, `--ss_loss --with_depth_loss`): primary render of
    4096 random rays of view v; their depth-prior points warped into a random other training view (a12, VT variant), occlusion test,
    a SECOND full render on the warped rays + its four loss terms; the primary render's terms restricted by the masks per coin;
    backward through both renders; Adam.  ~2x the MLP work of a plain step.
    route "ss_step_loss": the one-call surface, ONE render (round 6: run_nerf_view.ss_step_loss -> the warp launch assembles the
    combined 2 x 4096-row batch and its device-side live-row count, the MLP launches stop at the count, two-segment loss tail; no host
    synchronisation); "two_renders": round 5's one-call form (two renders, one read-back); "reference_lines": render, ss_consistency,
    ss_primary_losses as separate calls in the reference's order."""
    from consistentnerf_amd import raybank as RB, run_nerf as R, run_nerf_view as V
    H, W, K, kw, opt = sc["H"], sc["W"], sc["K"], sc["kw"], sc["opt"]
    rs = np.random.RandomState(3)

    def step(i):
        v, r = i % 3, (i + 1 + int(rs.randint(0, 2))) % 3
        rays, target, sel, (d_prior,) = RB.sample_patch_rays(sc["img_t"][v], sc["poses"][v], H, W, K, 4096, None,
                                                             extras=(sc["dep_ss_t"][v],), render_kwargs=kw)
        coins = [int(c) for c in rs.randint(0, 2, 4)]
        if route in ("ss_step_loss", "two_renders"):
            loss, ss = V.ss_step_loss(H, W, K, rays, target, d_prior, sc["poses"][r], sc["img_t"][r], sc["dep_ss_t"][r], kw, chunk=32768,
                                      occlusion_threshold=0.1, with_depth_loss=True, coins=coins,
                                      route=None if route == "ss_step_loss" else "two_renders")
            opt.zero_grad()
            R.backward(loss)
        else:
            rgb, disp, acc, depth, extras = V.render(H, W, K, chunk=32768, rays=rays, retraw=True, **kw)
            ss = V.ss_consistency(rays[0], rays[1], d_prior, sc["poses"][r], K, sc["img_t"][r], sc["dep_ss_t"][r], H, W, kw, chunk=32768,
                                  occlusion_threshold=0.1, with_depth_loss=True)
            opt.zero_grad()
            lp, _, _ = V.ss_primary_losses(rgb, depth, extras, target, d_prior, ss["mask_bound"], ss["mask"], with_depth_loss=True,
                                           coins=coins, sel=ss["sel"])
            loss = ss["loss"] + lp
            loss.backward()
        opt.step()
        if "live" in ss:          # one render: the row count stays on the device — kept as tensors, read once after the timed region
            step.lives.append(ss["live"])
            step.terms = ss["terms"]
        else:
            step.lives.append(4096 + int(ss["batch_rays_ref"].shape[1]))
        # VT:959 / VT:966 with both coarse coins 0: the primary rays' coarse level has no backward (as in the reference's graph)
        step.skips.append(4096 if not (coins[2] or coins[3]) else 0)
        return loss
    step.lives, step.skips, step.terms = [], [], None
    return step


def _time_steps(step, steps, warm=3):
    from consistentnerf_amd import ops
    for i in range(warm):
        step(i)
    torch.cuda.synchronize()
    ops.PROFILE = []
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(warm + i)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    prof, ops.PROFILE = ops.PROFILE, None
    return dt, loss, prof


def c3_dispatch_pass(leg):
    """Launches of ONE step of a C3 leg: a `rocprofv3 --kernel-trace` pass over `bench.py --only-leg <leg>` (6 steps), the ordered
    dispatch list cut at the adam_k launches (launches_per_step)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return {"total": None, "source": "rocprofv3 not on PATH"}
    d = tempfile.mkdtemp(prefix="cnerf_c3_")
    try:
        cmd = ["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
               os.path.abspath(__file__), "--only-leg", leg]
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=240, env=dict(os.environ, TMPDIR="/tmp"), cwd="/tmp")
        except subprocess.TimeoutExpired:
            return {"total": None, "source": "rocprofv3 --kernel-trace pass timed out"}
        files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return {"total": None, "source": f"rocprofv3 pass rc={r.returncode}: {r.stderr[-300:]}"}
        disp = {}
        for f in files:
            with open(f) as fh:
                for row in csv.DictReader(fh):
                    disp[(int(row["Start_Timestamp"]), int(row["Dispatch_Id"]))] = row["Kernel_Name"]
        keys = sorted(disp)
        out = launches_per_step({k: disp[key] for k, key in enumerate(keys)})
        if out is not None:
            out["source"] = f"rocprofv3 --kernel-trace pass of `bench.py --only-leg {leg}`; one step = the dispatches between two adam_k"
        return out or {"total": None, "source": "fewer than two adam_k launches in the trace"}
    finally:
        shutil.rmtree(d, ignore_errors=True)


def only_leg(dev, leg):
    """`--only-leg c3 | c3_ss`: a few steps of that leg and nothing else (what c3_dispatch_pass profiles)."""
    sc = c3_scene(dev)
    step = c3_step_fn(sc) if leg == "c3" else c3_ss_step_fn(sc)
    for i in range(6):
        step(i)
    torch.cuda.synchronize()


def c3_leg(dev, steps=20):
    """BASELINE configs[2]: LLFF-like 3-view training step WITH the consistency terms — hard masks from the cross-view depth
    warp (V:994-1046), masked rgb + depth losses on both levels (V:1645-1865), the monocular-depth patch term on 4 16x16
    patches (V:1678-1720), clip 0.1 + Adam (V:1983).  378x504 views, no_ndc, near 1.2 / far 12, 4096 random + 1024 patch
    rays per step.  Timed through the round-5 one-call surface (c3_step_fn "render_loss"); the reference-lines form beside it."""
    sc = c3_scene(dev)
    B = 4096 + 1024
    n = B * (NC + NC + NF)
    dt, loss, prof = _time_steps(c3_step_fn(sc), steps)
    table = per_kernel_table(prof, dt * steps * 1e3)
    tf = n * 2 * (MAC_FWD + MAC_DGRAD + MAC_WGRAD) / dt / 1e12
    dt_lines, loss_lines, _ = _time_steps(c3_step_fn(sc, "reference_lines"), steps)
    masks = sc["masks"]
    out = {"ms_per_step": dt * 1e3, "steps": steps, "rays_per_step": B, "ray_samples_per_s": n / dt,
           "ms_per_step_reference_lines": dt_lines * 1e3,
           "hard_masks_3views_ms": sc["t_masks"] * 1e3, "hard_mask_fraction": float(masks.mean()), "final_loss": float(loss.item()),
           "finite": bool(np.isfinite(loss.item())),
           "step": "3 LLFF-like 378x504 views, no_ndc; ONE sampling launch (patch + random pixels drawn in-kernel, rays / colours / "
                   "priors gathered, ray rows packed); hard masks + masked rgb/depth losses on both levels + monocular patch term "
                   "(4 x 16x16) folded into the compositing launches (run_nerf_view.render_loss); clip 0.1 + Adam; 4096 random + 1024 "
                   "patch rays, 64+128 samples, D=8 W=256",
           "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4), "basis": "whole step, 3489024 FLOP per ray-sample",
                        "kernels": table},
           "launches_per_step": c3_dispatch_pass("c3")}
    del sc
    torch.cuda.empty_cache()
    return out


def c3_ss_leg(dev, steps=10):
    """a15 timed (VERDICT r04 missing 4): the in-loop consistency step, c3_ss_step_fn."""
    sc = c3_scene(dev)
    dt_lines, _, _ = _time_steps(c3_ss_step_fn(sc, "reference_lines"), steps)
    dt_two, _, _ = _time_steps(c3_ss_step_fn(sc, "two_renders"), steps)
    step = c3_ss_step_fn(sc)
    dt, loss, prof = _time_steps(step, steps)
    table = per_kernel_table(prof, dt * steps * 1e3)
    # MFMA work of a step = the ray-samples of its LIVE rows (4096 primary + M warped; the launches are sized for 8192 rows and stop
    # at the device-side count): 256 ray-samples per live row through forward, dgrad and wgrad
    lives, skips = [int(x) for x in step.lives[-steps:]], step.skips[-steps:]
    pts = 256.0 * sum(lives) / steps                       # forward: every live row, both levels
    pts_bwd = pts - 64.0 * sum(skips) / steps              # backward: minus the primary rays' coarse level on the steps that skip it
    for r in table:      # (the HIP-event records carry the launch CAPACITY: rescale every MFMA row to the points it really covered)
        r["points_capacity"] = r["points"]
        r["points"] = r["points"] * (sum(lives) / steps) / 8192.0 if r["kernel"] == "mlp_fwd_train" else pts_bwd
    tf = (pts * 2 * MAC_FWD + pts_bwd * 2 * (MAC_DGRAD + MAC_WGRAD)) / dt / 1e12
    # (the second render's ray count differs from step to step: one row per kernel here, sizes pooled — the launch-size-resolved
    #  table would be ~40 rows of this line)
    pooled = {}
    for r in table:
        a = pooled.setdefault(r["kernel"], {"kernel": r["kernel"], "launches": 0, "ms": 0.0, "points": 0})
        a["launches"] += r["launches"]; a["ms"] += r["launches"] * r["avg_ms"]; a["points"] += r["launches"] * r["points"]
    fl = {"mlp_fwd_train": 2 * MAC_FWD, "mlp_dgrad": 2 * MAC_DGRAD, "mlp_wgrad": 2 * MAC_WGRAD}
    table = [{"kernel": a["kernel"], "launches_per_step": a["launches"] / steps, "ms_per_step": round(a["ms"] / steps, 4),
              "points_per_step": a["points"] / steps,
              "frac": round(fl[a["kernel"]] * a["points"] / (a["ms"] * 1e-3) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4) if a["kernel"] in fl else None,
              "share_of_step": round(a["ms"] / (dt * steps * 1e3), 4)} for a in pooled.values()]
    out = {"ms_per_step": dt * 1e3, "ms_per_step_two_renders": dt_two * 1e3, "ms_per_step_reference_lines": dt_lines * 1e3, "steps": steps,
           "rays_primary": 4096, "rays_second_render_last_step": lives[-1] - 4096, "live_rows_avg": sum(lives) / steps,
           "ray_samples_per_step_avg": pts, "ray_samples_backward_per_step_avg": pts_bwd, "ray_samples_per_s": pts / dt,
           "final_loss": float(loss.item()),
           "final_terms": None if step.terms is None else {k: float(v) for k, v in step.terms.items()},
           "finite": bool(np.isfinite(loss.item())),
           "loss_note": "un-normalised depth MSEs (VT:933, VT:950) against priors bounded by the far plane; VT's warp has no "
                        "OpenGL->OpenCV flip (VT:451-501), so on this OpenGL-convention rig the warped rays look away from the scene "
                        "and the occlusion threshold doubles to 6.4: the values are what the reference's lines compute here",
           "step": "VT:895-972 with --ss_loss --with_depth_loss through run_nerf_view.ss_step_loss as ONE render: the warp launch "
                   "assembles the 8192-row batch [4096 primary | M warped | padding] + its device-side live-row count, both levels' "
                   "MLP launches stop at the count, every loss term of both segments in the compositing launches + one two-segment "
                   "tail; merged backward; Adam.  No host synchronisation.",
           "roofline": {"bound": "mfma", "achieved": round(tf, 2), "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                        "frac": round(tf / PEAK_FP32_MFMA_TFLOPS, 4),
                        "basis": "whole step / (live ray-samples x forward FLOPs + the ray-samples the backward covered x (dgrad + wgrad) "
                                 "FLOPs): the coarse level of the primary rays has no backward when neither coarse coin selects it",
                        "kernels": table},
           "launches_per_step": c3_dispatch_pass("c3_ss")}
    del sc
    torch.cuda.empty_cache()
    return out


def pmc_traffic(kernel, points):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE
    each in their own `--pmc` run, scripts/gpu_pmc.sh -> profiles/r03_pmc/): KB units,
    FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md (16-byte-per-lane streaming reads are tallied at
    half).  A STATIC LOOKUP, not a measurement of this run (counters cannot be sampled from inside this process); None when
    no pass of that kernel at that launch size is on file."""
    import csv
    here = os.path.dirname(os.path.abspath(__file__))
    name = {"mlp_wgrad": "wgrad", "mlp_dgrad": "mlp_dgrad", "mlp_fwd_train": "mlp_fwd_train", "mlp_fwd": "mlp_fwd_inf"}.get(kernel)
    # grid threads of the launch: forward / dgrad run 64 threads per 32 points; wgrad's grid is (point ranges) x (GEMMs) x
    # 256 threads: 128 x 14 for the fine level alone (round 1), 128 x 28 for both levels in one grid (round 2)
    # round 3: a 1-D grid of (ranges) x 256 threads: 64 ranges x 14 GEMMs for the fine level alone, 64 x 28 for both levels
    want = {"wgrad": {786432: 64 * 14 * 256, 1048576: 64 * 28 * 256}.get(points)}.get(name, 2 * points)
    for d in ("r03_pmc",):
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


def per_kernel_table(prof, elapsed_ms):
    """HIP-event records of ops._timed -> rows (kernel, points, launches, avg_ms, TFLOP/s, share of the timed region)."""
    kern = {}
    for nme, units, e0, e1 in prof:
        k = kern.setdefault((nme, units), [0.0, 0])
        k[0] += e0.elapsed_time(e1)
        k[1] += 1
    flops = {"mlp_fwd_train": 2 * MAC_FWD, "mlp_fwd": 2 * MAC_FWD, "mlp_dgrad": 2 * MAC_DGRAD, "mlp_wgrad": 2 * MAC_WGRAD,
             "mlp_fwd_train_bf3": 2 * MAC_FWD, "mlp_dgrad_bf3": 2 * MAC_DGRAD, "mlp_wgrad_bf3": 2 * MAC_WGRAD}
    table = []
    for (nme, units), (ms, n) in kern.items():
        if nme not in flops:
            continue
        avg_ms = ms / n
        tf = flops[nme] * units / (avg_ms * 1e-3) / 1e12
        peak = PEAK_BF16X3_TFLOPS if nme.endswith("_bf3") else PEAK_FP32_MFMA_TFLOPS
        table.append({"kernel": nme, "points": units, "launches": n, "avg_ms": round(avg_ms, 4), "tflops": round(tf, 2),
                      "frac": round(tf / peak, 4), "share_of_step": round(ms / elapsed_ms, 4)})
    table.sort(key=lambda r: -r["avg_ms"] * r["launches"])
    return table


class Workload:
    """The C2 training step on this rank's shard: model, optimizer, ray bank, exchange."""

    def __init__(self, dev, rank, world, seed=1234, precision="fp32", loss_entry="render_loss"):
        import tempfile
        import torch.distributed as dist
        from consistentnerf_amd import distributed as D, run_nerf as R
        self.R, self.D, self.dev, self.rank, self.world = R, D, dev, rank, world
        self.loss_entry = loss_entry
        if loss_entry == "render_loss":
            self.fwd_bwd = self.fwd_bwd_fused
        torch.manual_seed(seed)                       # identical init on every rank (replicated weights)
        with tempfile.TemporaryDirectory() as tmp:
            self.kw, _, _, self.grad_vars, self.opt = R.create_nerf(make_args(tmp))
        self.kw.update(near=NEAR, far=FAR)
        for k_ in ("network_fn", "network_fine"):     # (the headline line is "fp32" = exact fp32 MFMA; "bf16x3" = the opt-in leg)
            self.kw[k_].training_precision = precision
        self.K, self.bank, self.targets = build_ray_bank(dev)
        # render() takes (rays_o, rays_d) (R:70-137): kept as two contiguous banks, so that a batch is two row slices (no
        # stack / cat launch per step; the reference slices its pre-shuffled rays_rgb the same way, R:720-729)
        self.bank_o, self.bank_d = self.bank[:, 0:3].contiguous(), self.bank[:, 3:6].contiguous()
        torch.manual_seed(99 + rank)                  # per-rank jitter streams (RegNeRF/train.py:364-365 precedent)
        # the step's gradient exchange: per-network slices of the flat fp32 gradient, all-reduced (RCCL) as _MlpFn.backward
        # reports them final, the 1/world folded into the Adam kernel; a no-op without a process group
        self.reducer = D.GradReducer(self.opt, [self.kw['network_fn'], self.kw['network_fine']], mean=True,
                                     timing=dist.is_initialized(), fold_scale=True)

    def set_sharding(self, per_rank, strong):
        """strong (the global batch cut N ways): every rank draws the jitter / resampling streams of the WHOLE batch from the same
        generator state and takes its rows (`_global_rows`), so that the N-rank step sees the random numbers of the 1-rank step
        on that batch; weak: independent per-rank streams (RegNeRF/train.py:364-365)."""
        self.global_rows = (self.rank * per_rank, per_rank * self.world) if (strong and self.world > 1) else None
        torch.manual_seed(99 if self.global_rows is not None else 99 + self.rank)

    def fwd_bwd(self, rays_o, rays_d, tgt):
        R = self.R
        rays_od = (rays_o, rays_d)    # origins / directions of this batch: two contiguous [B, 3] row blocks of the pre-split bank
        extra_kw = {} if getattr(self, "global_rows", None) is None else {"_global_rows": self.global_rows}
        rgb, disp, acc, extras = R.render(H_IMG, W_IMG, self.K, chunk=32768, rays=rays_od, retraw=True, **self.kw, **extra_kw)
        self.opt.zero_grad()
        loss = R.img2mse(rgb, tgt) + R.img2mse(extras['rgb0'], tgt)
        loss.backward()
        return loss

    def fwd_bwd_fused(self, rays_o, rays_d, tgt):
        """The same step through run_nerf.render_loss (R:764-775 as one call: the two img2mse terms ride in the compositing launches,
        their seeds are formed inside the compositing backward; bit-identical loss-side gradients, tests/test_gpu_fused_step.py) and
        run_nerf.backward (no ones_like fill): 14 launches per step instead of 20."""
        R = self.R
        extra_kw = {} if getattr(self, "global_rows", None) is None else {"_global_rows": self.global_rows}
        loss = R.render_loss(H_IMG, W_IMG, self.K, tgt, chunk=32768, rays=(rays_o, rays_d), retraw=True, **self.kw, **extra_kw)[0]
        self.opt.zero_grad()
        R.backward(loss)
        return loss

    def body(self, rays_o, rays_d, tgt):
        loss = self.fwd_bwd(rays_o, rays_d, tgt)
        self.reducer.finish()
        self.opt.step(grad_scale=self.reducer.grad_scale)
        return loss

    def timed_body(self, rays_o, rays_d, tgt):
        """body() with a device synchronisation after each phase -> host wall ms per phase (diagnostic; cf. GraphedStep.timed_call)."""
        t = {}

        def lap(name, t0):
            torch.cuda.synchronize()
            t[name] = (time.perf_counter() - t0) * 1e3
            return time.perf_counter()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        self.fwd_bwd(rays_o, rays_d, tgt)
        t0 = lap("fwd_bwd_ms", t0)
        self.reducer.finish()
        t0 = lap("exchange_ms", t0)
        self.opt.step(grad_scale=self.reducer.grad_scale)
        lap("adam_ms", t0)
        return t

    def batch(self, i, per_rank):
        """rank's contiguous slice of global batch i (every rank holds the identical, identically shuffled bank)"""
        gstep = per_rank * self.world
        lo = (i * gstep + self.rank * per_rank) % (self.bank.shape[0] - per_rank)
        return self.bank_o[lo:lo + per_rank], self.bank_d[lo:lo + per_rank], self.targets[lo:lo + per_rank]

    def graphed(self, per_rank, collective):
        from consistentnerf_amd.graph import GraphedStep
        ex = (self.bank_o[0:per_rank], self.bank_d[0:per_rank], self.targets[0:per_rank])
        if self.world == 1 and not self.reducer_active():
            return GraphedStep(self.body, self.opt, ex, warmup=3)
        return GraphedStep(self.fwd_bwd, self.opt, ex, warmup=3, reducer=self.reducer, collective=collective)

    def reducer_active(self):
        import torch.distributed as dist
        return dist.is_initialized()

    def run(self, per_rank, steps, warmup, i0=0, graphed=None, profile=True):
        """`warmup` untimed + `steps` timed steps bracketed by barrier + synchronize; -> (elapsed s (max over ranks), last
        loss tensor, per-kernel HIP-event records)."""
        from consistentnerf_amd import ops
        import torch.distributed as dist

        def step(i):
            b = self.batch(i, per_rank)
            loss = graphed(*b) if graphed is not None else self.body(*b)
            lr = 5e-4 * (0.1 ** (i / (250 * 1000)))
            for pg in self.opt.param_groups:
                pg['lr'] = lr
            return loss
        for i in range(warmup):
            step(i0 + i)
        torch.cuda.synchronize()
        self.D.barrier()
        torch.cuda.synchronize()
        ops.PROFILE = [] if (profile and graphed is None) else None
        self.reducer.exposed.clear()
        t0 = time.perf_counter()
        for i in range(steps):
            loss = step(i0 + warmup + i)
        torch.cuda.synchronize()
        self.D.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        prof, ops.PROFILE = (ops.PROFILE or []), None
        if self.world > 1:
            t = torch.tensor([elapsed], device=self.dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = t.item()
        return elapsed, loss, prof

    def host_enqueue_ms(self, per_rank, n=20):
        """Host time to ENQUEUE one eager step (launch-side cost: ~100 ctypes / ATen launches + autograd), measured with the
        GPU drained before and after so that the host never waits on it inside."""
        ts = []
        for i in range(n):
            b = self.batch(i, per_rank)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            self.body(*b)
            ts.append(time.perf_counter() - t0)
        torch.cuda.synchronize()
        ts.sort()
        return 1e3 * ts[len(ts) // 2]


def shard_leg(wl, per_rank, steps, warmup, collective="split", i0=100000, also_capture=False):
    """The `per_rank`-ray step of a strong-scaling shard on this process group: eager and graphed, per-kernel table.
    also_capture (RCCL groups): a third timing with the all-reduce recorded INSIDE the graph, attempted last and reported as an
    error string if the recording fails."""
    ms = {}
    wl.set_sharding(per_rank, strong=True)
    el, loss, prof = wl.run(per_rank, steps, warmup, i0=i0)
    ms["eager"] = el / steps * 1e3
    table = per_kernel_table(prof, el * 1e3)
    host = wl.host_enqueue_ms(per_rank)
    g = wl.graphed(per_rank, collective)
    el_g, loss_g, _ = wl.run(per_rank, steps, warmup, i0=i0 + steps + warmup, graphed=g)
    ms["graph"] = el_g / steps * 1e3
    # per-phase view of both forms (a device synchronisation after every phase: the sum bounds a step from above, the split says
    # where a slow data-parallel step spends its time — VERDICT r05 item 7: the graphed `split` step at world 2 over gloo on one
    # GPU ran at 88 ms against 29.8 eager, and nothing said whether that was the exchange, the replay or the host)
    import statistics
    phases = {"graph": {}, "eager": {}}
    for k in range(6):
        b = wl.batch(i0 + 3 * (steps + warmup) + k, per_rank)
        _, t = g.timed_call(*b)
        for name, v in t.items():
            phases["graph"].setdefault(name, []).append(v)
        te = wl.timed_body(*wl.batch(i0 + 3 * (steps + warmup) + 6 + k, per_rank))
        for name, v in te.items():
            phases["eager"].setdefault(name, []).append(v)
    wl.D.barrier()
    phase_ms = {form: {name: round(statistics.median(v), 3) for name, v in d.items()} for form, d in phases.items()}
    ms_capture = None
    if also_capture and wl.reducer_active():
        try:
            gc = wl.graphed(per_rank, "capture")
            el_c, _, _ = wl.run(per_rank, steps, warmup, i0=i0 + 2 * (steps + warmup), graphed=gc)
            ms_capture = round(el_c / steps * 1e3, 4)
        except Exception as e:  # noqa: BLE001
            ms_capture = f"{type(e).__name__}: {e}"
    n = per_rank * (NC + NC + NF) * wl.world
    mfma_ms = sum(r["avg_ms"] * r["launches"] for r in table) / steps
    ideal_ms = n / wl.world * 2 * (MAC_FWD + MAC_DGRAD + MAC_WGRAD) / (PEAK_FP32_MFMA_TFLOPS * 1e12) * 1e3
    return {"rays_per_gpu": per_rank, "global_batch": per_rank * wl.world, "steps": steps, "warmup": warmup,
            "ms_per_step_eager": round(ms["eager"], 4), "ms_per_step_graph": round(ms["graph"], 4),
            "graph_collective": collective if wl.reducer_active() else None, "ms_per_step_graph_capture": ms_capture,
            "ray_samples_per_s_eager": n / (ms["eager"] * 1e-3), "ray_samples_per_s_graph": n / (ms["graph"] * 1e-3),
            "host_enqueue_ms_per_eager_step": round(host, 3),
            "mfma_kernels_ms_per_step": round(mfma_ms, 4), "step_ms_at_mfma_peak": round(ideal_ms, 4),
            "frac_of_peak_graph": round(ideal_ms / ms["graph"], 4), "frac_of_peak_eager": round(ideal_ms / ms["eager"], 4),
            "phase_ms_synchronised": phase_ms,
            "final_loss": float(loss_g.item()), "kernels": table}


def _free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def self_launch(n):
    """`python bench.py --gpus N` without torchrun's environment: re-run THIS command line under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` (one rank per GPU), pass rank 0's one
    JSON line through, return the launcher's exit status.  Fewer than N visible GPUs under RCCL, a launcher failure or a run that
    produced no line -> ONE JSON error line on stdout and a non-zero status (no traceback).  (Single-host multi-device launch from
    one command: RegNeRF/train.py:326-328 does the same with jax.pmap over the local devices.)"""
    import subprocess
    backend = os.environ.get("CNERF_DIST_BACKEND", "nccl")
    ngpu = torch.cuda.device_count()

    def fail(msg, rc, **kw):
        err = {"metric": "train_ray_samples_per_sec", "value": None, "unit": "ray-samples/s", "n_gpus": n, "error": msg}
        err.update(kw)
        sys.stdout.write(json.dumps(err) + "\n")
        sys.stdout.flush()
        return rc

    if backend == "nccl" and ngpu < n:
        return fail(f"--gpus {n} needs {n} visible GPUs (one rank per GPU over RCCL); this host shows {ngpu}", 2, gpus_visible=ngpu)
    if ngpu < 1:
        return fail("no GPU visible (there is no CPU execution path)", 2, gpus_visible=0)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    try:
        limit = float(os.environ.get("CNERF_BENCH_LAUNCH_TIMEOUT", "1500"))
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=None, timeout=limit)
        rc, text = r.returncode, r.stdout.decode(errors="replace")
    except subprocess.TimeoutExpired as e:
        return fail(f"the {n}-rank run did not finish within {limit:.0f} s", 3, launcher=" ".join(cmd[:9]),
                    stdout_tail=(e.stdout or b"").decode(errors="replace")[-400:])
    line = None
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and '"metric"' in ln:
            try:
                json.loads(ln)
                line = ln
            except ValueError:
                pass
    if line is None:
        return fail(f"the {n}-rank run ended with status {rc} and printed no result line", rc or 4, launcher=" ".join(cmd[:9]),
                    stdout_tail=text[-400:])
    sys.stdout.write(line + "\n")
    sys.stdout.flush()
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: a timed region of ~5 s (200 x 26 ms), long enough for a coarse (seconds) GPU-busy sampler to see it
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: 4096 rays per GPU (global batch 4096 N); strong: the 4096-ray batch sharded N ways (C4)")
    ap.add_argument("--rays-per-gpu", type=int, default=0, help="override the per-GPU shard (e.g. 512 = the 8-way C4 shard)")
    ap.add_argument("--loss-entry", choices=("render_loss", "reference_lines"), default="render_loss",
                    help="render_loss: R:764-775 as one call (run_nerf.render_loss + run_nerf.backward, the loss folded into the "
                         "compositing launches); reference_lines: render() + img2mse() + img2mse() + loss.backward() as the "
                         "reference's loop writes them")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the legs that follow the timed region")
    ap.add_argument("--graph", action="store_true",
                    help="replay the step as captured hipGraph(s) (consistentnerf_amd/graph.py); the per-kernel table then comes "
                         "from a separate eager pass of the same steps, and the JSON says so")
    ap.add_argument("--graph-collective", choices=("split", "capture"), default="split")
    ap.add_argument("--pmc", nargs="?", const="on", default="auto", choices=("auto", "on", "off"),
                    help="fill roofline.traffic from two rocprofv3 --pmc passes of THIS command (auto: at N=1 when the extra legs run and "
                         "rocprofv3 is on PATH; the committed lookup is the fallback)")
    ap.add_argument("--only-leg", choices=("c3", "c3_ss"), default=None, help=argparse.SUPPRESS)
    a = ap.parse_args()
    if a.only_leg:
        sys.stdout.flush()
        os.dup2(2, 1)
        torch.cuda.set_device(0)
        only_leg(torch.device("cuda", 0), a.only_leg)
        return
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` launched plainly (no torchrun around it): become the launcher
        raise SystemExit(self_launch(a.gpus))

    # stdout carries exactly ONE line, the JSON: everything else that writes to fd 1 — the reference-style prints of
    # create_nerf(), and RCCL's version banner, which the C library flushes at exit, i.e. AFTER the JSON — goes to stderr
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch.distributed as dist
    from consistentnerf_amd import distributed as D, ops
    # backend: RCCL ("nccl") for a real multi-GPU run.  CNERF_DIST_BACKEND=gloo lets the N>1 code path of this script (sharded
    # steps, GradReducer, barrier, max-over-ranks) be exercised on a box with fewer GPUs than ranks — the ranks then share
    # devices (rank % device_count) and the numbers mean nothing as a scaling measurement; the JSON's dist.backend says so.
    backend = os.environ.get("CNERF_DIST_BACKEND", "nccl")
    ngpu = torch.cuda.device_count()
    if backend == "nccl" and int(os.environ.get("LOCAL_RANK", "0")) >= max(ngpu, 1):
        raise SystemExit(f"LOCAL_RANK {os.environ.get('LOCAL_RANK')} but only {ngpu} GPU(s) visible: one rank per GPU over RCCL")
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(ngpu, 1))
    rank, world, local = D.init_from_env(backend if (a.gpus > 1 or os.environ.get("CNERF_FORCE_DIST") == "1") else None)
    if world != a.gpus:
        if rank == 0:
            os.write(json_fd, (json.dumps({"metric": "train_ray_samples_per_sec", "value": None, "unit": "ray-samples/s",
                                           "n_gpus": a.gpus, "error": f"--gpus {a.gpus} but the launcher's WORLD_SIZE is {world}"})
                               + "\n").encode())
        raise SystemExit(2)
    local = local % max(ngpu, 1)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if dist.is_initialized():   # build the RCCL communicator now (seconds), not inside the first step
        t = torch.zeros(1, device=dev)
        dist.all_reduce(t)
        torch.cuda.synchronize()
    ok, name, cus, _ = ops.device_info(local)

    if a.rays_per_gpu > 0:
        per_rank = a.rays_per_gpu
    elif a.scaling == "strong":
        if B_PER_GPU % world:
            raise SystemExit(f"--scaling strong shards the {B_PER_GPU}-ray batch evenly: {world} ranks do not divide it")
        per_rank = B_PER_GPU // world
    else:
        per_rank = B_PER_GPU
    global LOSS_ENTRY
    LOSS_ENTRY = a.loss_entry
    wl = Workload(dev, rank, world, loss_entry=a.loss_entry)
    wl.set_sharding(per_rank, strong=(per_rank * world == B_PER_GPU and world > 1))
    graphed = wl.graphed(per_rank, a.graph_collective) if a.graph else None
    elapsed, loss, prof = wl.run(per_rank, a.steps, a.warmup, graphed=graphed)
    if graphed is not None:      # a replayed graph carries no events: the per-kernel table from an eager pass of the same steps
        el_e, _, prof = wl.run(per_rank, a.steps, 0, i0=a.warmup + a.steps)
        table = per_kernel_table(prof, el_e * 1e3)
    else:
        table = per_kernel_table(prof, elapsed * 1e3)
    final_loss = loss.item()
    dom = table[0]
    traffic, traffic_source = pmc_traffic(dom["kernel"], dom["points"]), (
        "static lookup: committed rocprofv3 PMC passes of this kernel at this launch size (profiles/*_pmc*/pass2+pass3 "
        "summaries, 2*FETCH_SIZE + WRITE_SIZE); NOT sampled in this run")
    pmc_live = None
    import shutil
    want_pmc = a.pmc == "on" or (a.pmc == "auto" and not a.no_extra and shutil.which("rocprofv3") is not None)
    if want_pmc and rank == 0 and world == 1:
        pmc_live = pmc_rerun(per_rank, dom["kernel"])
        if pmc_live and pmc_live.get("traffic") is not None:
            traffic, traffic_source = pmc_live["traffic"], pmc_live["source"]
    roofline = {"bound": "mfma", "kernel": f'{dom["kernel"]} (M={dom["points"]} points)', "achieved": dom["tflops"],
                "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(dom["tflops"] / PEAK_FP32_MFMA_TFLOPS, 4),
                "traffic": traffic, "traffic_source": traffic_source, "avg_launch_ms": dom["avg_ms"], "kernels": table}
    launches_live = pmc_live.pop("launches", None) if pmc_live else None
    if pmc_live:
        roofline["pmc"] = pmc_live
    if graphed is not None:
        roofline["kernels_measured"] = "separate eager pass of the same steps after the timed (graph-replayed) region"
    dist_info = None
    if dist.is_initialized():
        ex = wl.reducer.exposed_ms()
        be = dist.get_backend()
        dist_info = {"rccl_ranks": dist.get_world_size() if be == "nccl" else 0, "ranks": dist.get_world_size(), "backend": be,
                     "messages_per_step": round(wl.reducer.messages / max(wl.reducer.steps, 1), 3),
                     "slice_bytes": [4 * (hi - lo) for lo, hi in wl.reducer.slices.values()],
                     "bytes_per_step": wl.reducer.bytes_per_step, "one_over_world": "folded into the Adam kernel (grad_scale)",
                     "allreduce_exposed_ms": round(sum(ex) / max(len(ex), 1), 4), "allreduce_exposed_ms_max": round(max(ex), 4) if ex else None,
                     "measured": "HIP events on the launch stream: last slice issued -> launch stream released (this rank)"}

    out = None
    if rank == 0:
        samples_per_step = per_rank * (NC + NC + NF) * world
        exch = ""
        if world > 1:
            exch = (", RCCL all-reduce of the flat fp32 grad" if dist.get_backend() == "nccl"
                    else f", {dist.get_backend()} all-reduce of the flat fp32 grad (NOT RCCL: ranks share devices, no scaling claim)")
        cfgname = "BASELINE configs[1]" if (world == 1 and per_rank == B_PER_GPU) else (
            "BASELINE configs[3] (C4: the 4096-ray C2 batch sharded)" if per_rank * world == B_PER_GPU else
            ("BASELINE configs[3] shapes, weak scaling" if per_rank == B_PER_GPU else "a per-GPU shard of BASELINE configs[3]"))
        out = {
            "metric": "train_ray_samples_per_sec", "value": samples_per_step * a.steps / elapsed,
            "unit": "ray-samples/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if (a.scaling == "strong" and a.rays_per_gpu <= 0) else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic", "hip_graph": graphed is not None,
            "route": route_of(table, a.steps),
            "config": {"workload": f"DTU scan8 3-view (synthetic 512x640 ray bank), {per_rank} rays/GPU/step (global batch "
                                   f"{per_rank * world}), coarse 64 + fine 64+128 samples, D=8 W=256 viewdirs MLPs (random init), "
                                   f"perturb=1, mse(rgb)+mse(rgb0), backward, Adam; {cfgname}",
                       "loss_entry": ("run_nerf.render_loss + run_nerf.backward (R:764-775 as one call, loss folded into the compositing "
                                      "launches)" if a.loss_entry == "render_loss" else
                                      "render() + img2mse() + img2mse() + loss.backward() (the reference's lines)"),
                       "random_streams": "generated inside coarse_z_k / resample_k (Philox4x32-10 on the torch generator's seed / offset)",
                       "rays_per_gpu": per_rank, "global_batch": per_rank * world, "ray_samples_per_ray": NC + NC + NF,
                       "device": name, "cus": cus, "parallelism": f"ray-shard dp{world}" + exch, "final_loss": final_loss},
            "roofline": roofline,
        }
        if dist_info is not None:
            out["dist"] = dist_info

    import threading
    emit_lock, emitted, status = threading.Lock(), [False], [0]

    def emit():
        """ONE line, whoever gets here first (the watchdog thread or the main thread): compact_line(out) on stdout, the full
        object in the side file."""
        with emit_lock:
            if rank == 0 and not emitted[0]:
                emitted[0] = True
                sys.stdout.flush()
                path = write_detail(out, world)
                os.write(json_fd, (json.dumps(compact_line(out, path)) + "\n").encode())

    force_leg = os.environ.get("CNERF_BENCH_FORCE_LEG") == "1" and dist.is_initialized()     # (exercise the leg on a 1-rank group)
    if not a.no_extra and (world > 1 or force_leg) and per_rank == B_PER_GPU and B_PER_GPU % world == 0:
        # the same process group on the strong-scaling shard of C4 — a SIDE experiment (eager / split-graph / RCCL recorded inside the
        # graph) that has never seen real multi-GPU hardware.  At world > 1 the contract line therefore goes out FIRST: nothing this
        # leg does (an exception, a collective that never completes, an abort inside the communicator) can void the measurement of
        # the timed region.  Its result follows on stderr as its own JSON object (and in gpurun_out/ when that directory exists);
        # a watchdog ends the process if the leg has not finished (a hung collective cannot be interrupted from Python).  The exit
        # status stays the main leg's.
        if world > 1:
            emit()
        done = threading.Event()

        def watchdog():
            if not done.wait(float(os.environ.get("CNERF_BENCH_LEG_TIMEOUT", "120"))):
                if rank == 0:
                    out.setdefault("extra", {})["c4_strong"] = {"error": "leg did not finish within the watchdog's limit"}
                    sys.stderr.write(json.dumps({"after_the_line": out["extra"]}) + "\n")
                    emit()
                sys.stderr.flush()
                os._exit(status[0] if world > 1 else 6)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            leg = shard_leg(wl, B_PER_GPU // (8 if force_leg and world == 1 else world), 100, 10, collective="split",
                            # RCCL recorded INSIDE the graph has only ever run on a 1-rank group here: opt-in on real multi-GPU
                            # hardware (CNERF_BENCH_CAPTURE_RCCL=1), so that the default run cannot end in an abort of the communicator
                            also_capture=(dist.get_backend() == "nccl"
                                          and (world == 1 or os.environ.get("CNERF_BENCH_CAPTURE_RCCL") == "1")))
        except Exception as e:  # noqa: BLE001 — the main line must survive a failure of the side leg
            leg = {"error": f"{type(e).__name__}: {e}"}
            if world == 1:
                status[0] = 5
        done.set()
        if rank == 0:
            out["extra"] = {"note": "same process group, after the timed region; not part of `value`", "c4_strong": leg}
            if world > 1:
                blob = json.dumps({"after_the_line": out["extra"], "n_gpus": world})
                sys.stderr.write(blob + "\n")
                try:
                    if os.path.isdir("gpurun_out"):
                        with open(os.path.join("gpurun_out", f"bench_c4_strong_{world}gpus.json"), "w") as f:
                            f.write(blob + "\n")
                except OSError:
                    pass
    if rank == 0:
        if world == 1 and not a.no_extra and not force_leg:
            extra = {"note": "same process, after the timed C2 region; not part of `value`"}
            if per_rank == B_PER_GPU:
                extra["c4_shard"] = shard_leg(wl, B_PER_GPU // 8, 200, 20)
                extra["c4_shard"]["what"] = ("the 512-ray per-GPU step of the 8-way strong-scaling shard of the 4096-ray C2 batch, "
                                             "on this one GPU (no exchange: world 1)")
            del wl, graphed
            torch.cuda.empty_cache()
            if per_rank == B_PER_GPU:
                try:
                    extra["c2_bf16x3"] = bf16x3_leg(dev, rank, world, per_rank, elapsed / a.steps * 1e3)
                except Exception as e:  # noqa: BLE001 — the headline line must survive a failure of the opt-in leg
                    extra["c2_bf16x3"] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.empty_cache()
            extra["c5"] = c5_leg(dev)
            extra["c3"] = c3_leg(dev)
            try:
                extra["c3_ss"] = c3_ss_leg(dev)
            except Exception as e:  # noqa: BLE001 — the headline line must survive a failure of a side leg
                extra["c3_ss"] = {"error": f"{type(e).__name__}: {e}"}
            extra["hbm_kernels"] = hbm_kernels(dev)
            extra["launches_per_step"] = launches_live if launches_live is not None else {
                "total": None, "source": "needs the rocprofv3 pass (--pmc auto|on with rocprofv3 on PATH)"}
            out["extra"] = extra
            # the legs' headline scalars once more as FLAT keys of `config` (the driver's record keeps the scalar fields of config /
            # roofline / cpu_baseline and only lists the names of everything else: VERDICT r04 item 7)
            def g(d, *ks):
                for k in ks:
                    d = d.get(k) if isinstance(d, dict) else None
                return d
            flat = {"leg_launches_per_step_c2": g(extra, "launches_per_step", "total"),
                    "leg_aten_launches_per_step_c2": g(extra, "launches_per_step", "aten_and_runtime"),
                    "leg_c4_shard_ms_per_step_graph": g(extra, "c4_shard", "ms_per_step_graph"),
                    "leg_c4_shard_ms_per_step_eager": g(extra, "c4_shard", "ms_per_step_eager"),
                    "leg_c4_shard_frac_of_peak_graph": g(extra, "c4_shard", "frac_of_peak_graph"),
                    "leg_c2_bf16x3_ms_per_step": g(extra, "c2_bf16x3", "ms_per_step"),
                    "leg_c2_bf16x3_speedup_vs_f32_step": g(extra, "c2_bf16x3", "speedup_vs_f32_step"),
                    "leg_c5_frame_s": g(extra, "c5", "frame_s"),
                    "leg_c3_ms_per_step": g(extra, "c3", "ms_per_step"),
                    "leg_c3_frac_of_peak": g(extra, "c3", "roofline", "frac"),
                    "leg_c3_launches_per_step": g(extra, "c3", "launches_per_step", "total"),
                    "leg_c3_aten_launches_per_step": g(extra, "c3", "launches_per_step", "aten_and_runtime"),
                    "leg_c3_ss_ms_per_step": g(extra, "c3_ss", "ms_per_step"),
                    "leg_c3_ss_frac_of_peak": g(extra, "c3_ss", "roofline", "frac"),
                    "leg_c3_ss_launches_per_step": g(extra, "c3_ss", "launches_per_step", "total")}
            out["config"].update({k: (round(v, 4) if isinstance(v, float) else v) for k, v in flat.items()})
            out["roofline"]["whole_step_frac"] = round(
                per_rank * (NC + NC + NF) * 2 * (MAC_FWD + MAC_DGRAD + MAC_WGRAD) / (elapsed / a.steps) / 1e12 / PEAK_FP32_MFMA_TFLOPS, 4)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        emit()
    os.close(json_fd)
    if dist.is_initialized():
        if world > 1:
            # the line is out: a communicator that a failed experiment of the side leg left in a bad state must not keep the
            # process (and the launcher waiting for it) alive
            threading.Timer(30.0, lambda: os._exit(status[0] or 7)).start()     # teardown hung: say so in the status
        dist.destroy_process_group()
        if world > 1:
            sys.stderr.flush()
            os._exit(status[0])
    if status[0]:
        raise SystemExit(status[0])


if __name__ == "__main__":
    main()
