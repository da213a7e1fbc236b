"""Chaos-free training parity at the true C2 size.

The PSNR-parity clause of the north star cannot be closed by comparing two free-running fp32 Adam trajectories: they
decorrelate at the rate round-off is amplified (profiles/r03_psnr_*).  What CAN be asserted is the map one training step
applies — (weights, Adam moments, batch, random streams) -> (loss, gradient, new weights) — at points ALONG a real trajectory:
the HIP path trains the synthetic DTU-like scene at the true C2 shapes (scripts/psnr_parity.py::scene: 512x640 views, 4096-ray
batches, 64 + 128 samples, D=8/W=256, the reference's pytest=True RNG hook), and at steps {0, 2, 17, 50, 100, 150, 1000} the state is
snapshotted and ONE oracle step is run on the CPU from exactly that state (O.query -> O.composite on both levels -> mse + mse ->
autograd -> O.adam_step; reference: run_nerf.py:764-789, run_nerf.py:281, run_nerf_helpers.py:9-10).

Teacher forcing INSIDE the step.  The step map is piecewise smooth; its discontinuities are (a) the hidden ReLU patterns, (b) the
resampled fine depths (no gradient, R:397) and (c) the sign of the LAST sample's sigma — its interval is 1e10 wide (R:281), so
alpha_last jumps from 0 to 1 across sigma_last = 0 and the derivative between 0 and ~1e-8 is ~1e10.  The oracle is therefore
evaluated on the kernel's branch: at the kernel's depths (its own resampled depths are compared separately), with the kernel's ReLU
sign bits applied instead of its own (z > 0) (`masks` of O.query), and with the kernel's sigma_last substituted for the rays whose
own sigma_last is on the other side of 0 or inside the 1e-7 spike zone — every such substitution / pattern difference is counted and
must sit within round-off of the discontinuity.

The step is evaluated twice: in float64 (g_exact: the exact gradient of the reference's step map on that branch; the oracle's own
torch code run through stock ATen float64 kernels ON THE GPU — exactness does not depend on where it is computed, and it takes 2 s
instead of 30 s per snapshot) and in fp32 ON THE CPU (g_ref32: the reference arithmetic itself — ATen / MKL sgemm sum ~10^6 signed
products per weight in fp32).

Stated bounds:
  loss                       |d| <= 1e-6 relative, vs g_ref32's loss, vs the float64 loss, and vs the oracle running FREE
                             (O.render_rays_pytest: its own depths, ReLU patterns and tail signs) 1e-5
  gradient, per tensor       |g_HIP - g_exact| <= 1e-5 * A_max, where A = sum over the ray-samples of |dZ| |h| per element (float64)
                             is the mass the element's sum is formed from and A_max its largest value in the tensor: every
                             gradient element is a sum of ~10^6 signed products that increasingly cancel as training converges,
                             so the rounding error of ANY fp32 evaluation scales with A, not with the result.  Without
                             cancellation (A_max = max|g|: the early steps) this IS 1e-5 * max|g|; the cancellation factor
                             A_max / max|g| (1 ... 30 here), the plain |d| / max|g| figure and the same two figures for the
                             reference's own fp32 arithmetic (g_ref32 - g_exact) are reported per tensor.
                             d loss / d raw (the compositing + loss backward alone) is compared the same way.
  ReLU pattern differences   only where the oracle's own |z| < 1e-5, fewer than 1e-5 of all units (measured: 2.4e-6 at step 0)
  sigma-branch substitutions only where the kernel's |sigma| < 1e-5 (relu(sigma) of R:284 at every sample: a kink; at the LAST
                             sample a jump — reported separately as tail substitutions)
  Adam on the SAME gradient  new weights |d| <= 2e-7 (+ 2 ulp), moments 1e-6 relative to their max
  new weights end to end     |d| <= 2e-6 + the first-order propagation of the measured gradient difference through Adam's
                             normalisation (an element whose gradient history is ~0 moves by ~lr whatever its sign), and the
                             fraction of elements beyond the plain 2e-6 is reported and <= 1e-3
  fine depths                the oracle's own resampled depths vs the kernel's: median <= 2e-6 * far, 99 % <= 2e-5 * far, max <=
                             1e-3 * far (inverse-CDF interpolation divides by a CDF gap that may be as small as 1e-5, H:246-247:
                             a 1e-7 CDF difference moves such a sample by 1e-2 of its bin; test_gpu_parity pins the indices)
"""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch

from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SNAPSHOTS = (0, 2, 17, 50, 100, 150, 1000)     # round 5: + one late snapshot


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def _masks_from_stash(stash, M, D=8, W=256):
    """The ReLU patterns the kernel's backward uses, from its training stash: [M, W] bool per trunk layer + [M, W/2] for the
    view branch, on the CPU.  (_stash_blocks_dev decodes the sign-bit words and asserts they equal (h > 0) of the stored
    activations.)"""
    import test_gpu_parity as G
    blk = G._stash_blocks_dev(stash, M, D, W, True)
    masks = [(blk[f"h{l}"] > 0).cpu() for l in range(D)] + [(blk["hv"] > 0).cpu()]
    del blk
    torch.cuda.empty_cache()
    return masks


def _adam_first_order_bound(dg, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """|d w| allowed for a gradient difference of at most `dg` (scalar) per element, to first order in dg, through
    w -= lr/bc1 * m' / (sqrt(v'/bc2) + eps) with m' = b1 m + (1-b1) g, v' = b2 v + (1-b2) g^2."""
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    m1 = b1 * m + (1 - b1) * g
    v1 = b2 * v + (1 - b2) * g * g
    den = (v1 / bc2).sqrt() + eps
    d_m = (1 - b1) * dg
    d_v = (1 - b2) * 2 * g.abs() * dg + (1 - b2) * dg * dg
    d_den = d_v / bc2 / (2 * (v1 / bc2).sqrt() + eps)
    return lr / bc1 * (d_m / den + m1.abs() * d_den / (den * den))


EPS32 = 2.0 ** -24      # unit round-off of fp32


def _oracle_step(dtype, w0, names, rays, tgt, z_c, z_f, masks_c, masks_f, raw_k, ncfg, want_abs=False, device=None, loss_fn=None):
    """`device`: where the oracle's torch ops are evaluated.  None = the CPU (the reference arithmetic: what g_ref32 must be).  The
    float64 evaluation — whose only role is to be EXACT — may run the same oracle code on the GPU through stock ATen float64 kernels
    (30 s -> 2 s per snapshot); results come back on the CPU."""
    if device is not None:
        mv = lambda t: t.to(device)   # noqa: E731
        with torch.device(device):
            out = _oracle_step(dtype, [{k: mv(v) for k, v in d.items()} for d in w0], names, mv(rays), mv(tgt), mv(z_c), mv(z_f),
                               [mv(m) for m in masks_c], [mv(m) for m in masks_f], tuple(mv(r) for r in raw_k), ncfg, want_abs,
                               loss_fn=None if loss_fn is None else loss_fn.to(device))
        torch.cuda.synchronize()
        res = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in out.items()}
        res["d_raw"] = {k: v.cpu() for k, v in out["d_raw"].items()}
        del out
        torch.cuda.empty_cache()
        return res
    return _oracle_step_impl(dtype, w0, names, rays, tgt, z_c, z_f, masks_c, masks_f, raw_k, ncfg, want_abs, loss_fn)


def _oracle_step_cpu32(*common, loss_fn_factory=None):
    """The reference arithmetic itself: the oracle's step in fp32 ON THE CPU.  One full suite run of round 6 saw this evaluation —
    and only it: the HIP step and the float64 evaluation of the same inputs were bit-identical to every other run — come back with
    NaNs in four sample points on one box of the pool (stock ATen addmm on 32 threads; not reproducible on that test alone or in
    two further suite runs).  A non-finite CHECKER value is therefore recomputed once, single-threaded, and the row says so
    (`cpu_ref32_retried`); a second non-finite result fails the test as before."""
    mk = loss_fn_factory if loss_fn_factory is not None else (lambda: None)
    res = _oracle_step(torch.float32, *common, loss_fn=mk())
    res["retried"] = False
    if not (np.isfinite(res["loss"]) and bool(torch.isfinite(res["grad"]).all())):
        n = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            res = _oracle_step(torch.float32, *common, loss_fn=mk())
        finally:
            torch.set_num_threads(n)
        res["retried"] = True
    return res


def _oracle_step_impl(dtype, w0, names, rays, tgt, z_c, z_f, masks_c, masks_f, raw_k, ncfg, want_abs=False, loss_fn=None):
    """The oracle's training step on the KERNEL'S BRANCH (module docstring) in `dtype`: O.query -> O.composite on both levels ->
    mse + mse -> autograd.  raw_k = the kernel's (coarse, fine) raw outputs (fp32, CPU) for the tail-branch substitution.
    -> dict(loss, grad flat [dtype], flips, tail, A) with A = sum over samples of |dZ| |h| per gradient element (float64 runs)."""
    osd = [{k: v.to(dtype).clone().requires_grad_(True) for k, v in d.items()} for d in w0]
    r = rays.to(dtype)
    o_, d_, vd = r[:, 0:3], r[:, 3:6], r[:, 8:11]
    tg = tgt.to(dtype)
    recs, orig_lin = [], O._lin

    def lin(sd, name, x):
        y = orig_lin(sd, name, x)
        rec = {"k": 0 if sd is osd[0] else 1, "name": name, "x": x.detach()}
        y.register_hook(lambda g, rec=rec: rec.__setitem__("g", g.detach()))
        recs.append(rec)
        return y
    if want_abs:
        O._lin = lin
    try:
        flips, tail, loss, d_raw = [], [], 0.0, {}
        for k, z, mk in ((0, z_c, masks_c), (1, z_f, masks_f)):
            zz = z.to(dtype)
            raw = O.query(osd[k], o_[:, None, :] + d_[:, None, :] * zz[:, :, None], vd, ncfg, mk, flips)
            # the kernel's branch of relu(sigma) (R:284) at EVERY sample — a kink for the inner samples, a jump for the last one
            # (1e10-wide interval): where the oracle's sigma is on the other side of 0 (or, last sample, either value is inside
            # the ~1e-8-wide zone where d alpha / d sigma ~ 1e10) the kernel's value is substituted
            sk, so = raw_k[k][..., 3].to(dtype), raw[..., 3]
            sub = (so > 0) != (sk > 0)
            sub[:, -1] |= ((so[:, -1] > 0) & (so[:, -1] < 1e-7)) | ((sk[:, -1] > 0) & (sk[:, -1] < 1e-7))
            tail.append((int(sub[:, -1].sum()), int(sub.sum()), float(sk[sub].abs().max()) if bool(sub.any()) else 0.0))
            # (value = the kernel's, derivative d/d sigma = 1 into the oracle's own graph: the sample keeps feeding the MLP backward)
            raw = torch.cat([raw[..., :3], torch.where(sub, so + (sk - so).detach(), so)[..., None]], -1)
            raw.register_hook(lambda g, k=k: d_raw.__setitem__(k, g.detach()))
            comp = O.composite(raw, zz, d_)
            loss = loss + (O.mse(comp[0], tg) if loss_fn is None else loss_fn(k, comp, tg, dtype))
        plist = [osd[k][n] for k in (0, 1) for n in names[k]]
        grads = torch.autograd.grad(loss, plist, allow_unused=True)
    finally:
        O._lin = orig_lin
    flat = torch.cat([(torch.zeros_like(p) if g is None else g).reshape(-1) for p, g in zip(plist, grads)])
    out = {"loss": float(loss), "grad": flat, "flips": flips, "tail": tail, "d_raw": d_raw}
    if want_abs:
        A = [{n: torch.zeros_like(osd[k][n]) for n in names[k]} for k in (0, 1)]
        for rec in recs:
            g, x = rec["g"].abs(), rec["x"].abs()
            A[rec["k"]][rec["name"] + ".weight"] += g.t() @ x
            A[rec["k"]][rec["name"] + ".bias"] += g.sum(0)
        out["A"] = torch.cat([A[k][n].reshape(-1) for k in (0, 1) for n in names[k]])
    return out


def test_c2_teacher_forced_training_steps(dev, monkeypatch):
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import importlib
    import argparse
    import json
    P = importlib.import_module("psnr_parity")
    from consistentnerf_amd import ops, run_nerf as R
    P.set_size("c2")
    seed = 1                         # the seed whose free-running trajectory left the oracle's at step 2 in round 3
    K, bank, target, _, _, sds = P.scene(seed)
    args = argparse.Namespace(
        multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=128, netdepth=8, netwidth=256,
        netdepth_fine=8, netwidth_fine=256, netchunk=1024 * 64, lrate=P.LRATE, basedir=tempfile.mkdtemp(), expname="tf",
        ft_path=None, no_reload=True, perturb=1.0, N_samples=64, white_bkgd=False, raw_noise_std=0.0,
        dataset_type="dtu", no_ndc=True, lindisp=False)
    kw, _, _, grad_vars, opt = R.create_nerf(args)
    nets = [kw["network_fn"], kw["network_fine"]]
    for n_, sd in zip(nets, sds):
        n_.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    rr_kw = {k: v for k, v in kw.items() if k not in ("near", "far", "ndc", "use_viewdirs")}
    bank_d, target_d = bank.to(dev), target.to(dev)
    names = [[n for n, _ in m.named_parameters()] for m in nets]
    sizes = [[p.numel() for _, p in m.named_parameters()] for m in nets]
    seen = {}
    orig_pair = ops.mlp_backward_pair

    def spy(*a, **k):
        seen["args"] = a
        return orig_pair(*a, **k)
    monkeypatch.setattr(ops, "mlp_backward_pair", spy)

    ncfg, rcfg = O.NetCfg(8, 256, output_ch=5), O.RenderCfg(64, 128, 1.0)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    report, regime_rays = [], 0
    for i in range(max(SNAPSHOTS) + 1):
        lo, hi = P.batch_bounds(i, bank.shape[0])
        rb, tg = bank_d[lo:hi], target_d[lo:hi]
        snap = i in SNAPSHOTS
        if snap:
            torch.cuda.synchronize()
            w0 = [{k: v.detach().cpu().clone() for k, v in m.state_dict().items()} for m in nets]
            m0, v0, p0 = opt.exp_avg.cpu().clone(), opt.exp_avg_sq.cpu().clone(), opt.flat_param.cpu().clone()
            lr_i, step_i = opt.param_groups[0]["lr"], opt._step + 1
        out = R.render_rays(rb, retraw=True, pytest=True, _debug=snap, **rr_kw)
        opt.zero_grad()
        loss = R.img2mse(out["rgb_map"], tg) + R.img2mse(out["rgb0"], tg)
        seen.pop("args", None)
        loss.backward()
        if snap:
            assert "args" in seen, "the step did not take the merged coarse+fine backward"
            fs, fp, fg, fB, fS, fst, fgr, cs, cp, cg, cB, cS, cst, cgr = seen.pop("args")
            g_hip = opt.flat_grad.cpu().clone()
            masks_f = _masks_from_stash(fst, fB * fS)
            masks_c = _masks_from_stash(cst, cB * cS)
            d_raw_hip = (cg.detach().cpu().reshape(cB, cS, -1), fg.detach().cpu().reshape(fB, fS, -1))
            del fst, cst, fg, cg
            z_f, z_c = out["_z_vals"].cpu(), out["_z_coarse"].cpu()
            raw_k = (out["_raw_coarse"].detach().cpu(), out["raw"].detach().cpu())
        opt.step()
        for g_ in opt.param_groups:                # R:784-788
            g_["lr"] = P.LRATE * (0.1 ** (i / (P.LRATE_DECAY * 1000)))
        if not snap:
            continue
        torch.cuda.synchronize()
        loss_hip, p1_hip = float(loss), opt.flat_param.cpu().clone()
        m1_hip, v1_hip = opt.exp_avg.cpu().clone(), opt.exp_avg_sq.cpu().clone()
        rays_c, tgt_c = bank[lo:hi], target[lo:hi]
        common = (w0, names, rays_c, tgt_c, z_c, z_f, masks_c, masks_f, raw_k, ncfg)
        ex = _oracle_step(torch.float64, *common, want_abs=True, device=dev)
        r32 = _oracle_step_cpu32(*common)
        del masks_c, masks_f
        with torch.no_grad():      # the oracle running free: its own depths, ReLU patterns, tail signs
            osd = [{k: v.clone() for k, v in d.items()} for d in w0]
            fr = O.render_rays_pytest(rays_c, osd[0], osd[1], ncfg, rcfg, retraw=False)
            loss_free = float(O.mse(fr["rgb_map"], tgt_c) + O.mse(fr["rgb0"], tgt_c))
            dz = (fr["z_vals"] - z_f).abs().flatten()
        sig_last = raw_k[1][:, -1, 3]
        row = {"step": i, "cpu_ref32_retried": r32["retried"], "loss_hip": loss_hip, "loss_oracle_f32": r32["loss"], "loss_oracle_f64": ex["loss"],
               "loss_oracle_free_running": loss_free,
               "loss_rel_f32": abs(loss_hip - r32["loss"]) / abs(r32["loss"]),
               "loss_rel_f64": abs(loss_hip - ex["loss"]) / abs(ex["loss"]),
               "loss_rel_free": abs(loss_hip - loss_free) / abs(loss_free),
               "z_fine_own_vs_kernel_median": float(dz.median()),
               "z_fine_own_vs_kernel_p99": float(dz.kthvalue(int(dz.numel() * 0.99)).values),
               "z_fine_own_vs_kernel_max": float(dz.max())}
        tot_units = 4096 * (64 + 192) * (8 * 256 + 128)
        for tag, res in (("f64", ex), ("f32", r32)):
            row[f"relu_flips_{tag}"] = int(sum(n for n, _ in res["flips"]))
            row[f"relu_flip_max_abs_z_{tag}"] = max(z for _, z in res["flips"])
            row[f"tail_substitutions_{tag}"] = int(sum(t[0] for t in res["tail"]))
            row[f"sigma_sign_substitutions_{tag}"] = int(sum(t[1] for t in res["tail"]))
            row[f"tail_substitution_max_abs_sigma_{tag}"] = max(t[2] for t in res["tail"])
        row["relu_flip_frac"] = max(row["relu_flips_f64"], row["relu_flips_f32"]) / tot_units
        row["rays_sigma_last_lt_1e-2"] = int((sig_last.abs() < 1e-2).sum())
        row["rays_sigma_last_lt_1e-3"] = int((sig_last.abs() < 1e-3).sum())
        regime_rays += row["rays_sigma_last_lt_1e-2"]
        # the compositing + loss backward alone: d loss / d raw of both levels (what the MLP backward is fed)
        for k, tag in ((0, "coarse"), (1, "fine")):
            e_, h_, r_ = ex["d_raw"][k], d_raw_hip[k].double(), r32["d_raw"][k].double()
            sc = float(e_.abs().max())
            dh = (h_ - e_).abs()
            row[f"d_raw_{tag}_hip_vs_exact_rel_max"] = float(dh.max()) / sc
            row[f"d_raw_{tag}_ref32_vs_exact_rel_max"] = float((r_ - e_).abs().max()) / sc
            row[f"d_raw_{tag}_hip_vs_exact_rel_l2"] = float(dh.norm() / e_.norm())
            row[f"d_raw_{tag}_ref32_vs_exact_rel_l2"] = float((r_ - e_).norm() / e_.norm())
            fl = int(dh.reshape(-1).argmax())
            ray, smp, ch = fl // (dh.shape[1] * 4), (fl // 4) % dh.shape[1], fl % 4
            rk = raw_k[k][ray]
            row[f"d_raw_{tag}_worst_at"] = {"ray": ray, "sample": smp, "channel": ch, "sigma_there": float(rk[smp, 3]),
                                            "sigma_last": float(rk[-1, 3]), "exact": float(e_[ray, smp, ch]),
                                            "hip": float(h_[ray, smp, ch]), "ref32": float(r_[ray, smp, ch])}
        # gradient, per tensor: round-off units of the sum (K) and the plain relative-to-max figure
        g_ex, g_32, A = ex["grad"], r32["grad"], ex["A"]
        off, per_tensor, dg_flat = 0, {}, torch.zeros_like(g_hip)
        row["K_hip_worst"], row["K_ref32_worst"], row["grad_hip_vs_exact_rel_max_worst"] = 0.0, 0.0, 0.0
        row["grad_ref32_vs_exact_rel_max_worst"], row["grad_hip_vs_ref32_rel_max_worst"], bad = 0.0, 0.0, []
        for k in (0, 1):
            for nme, n in zip(names[k], sizes[k]):
                a, b, e, aa = g_hip[off:off + n].double(), g_32[off:off + n].double(), g_ex[off:off + n], A[off:off + n]
                dg_flat[off:off + n] = float((a - b).abs().max())
                off += n
                scale = float(e.abs().max())
                if scale == 0:
                    assert float(a.abs().max()) == 0 and float(b.abs().max()) == 0
                    continue
                amax = float(aa.max())
                d_ex, r_ex, d_32 = float((a - e).abs().max()), float((b - e).abs().max()), float((a - b).abs().max())
                full = ("coarse." if k == 0 else "fine.") + nme
                per_tensor[full] = {"hip_vs_exact_over_Amax": d_ex / amax, "ref32_vs_exact_over_Amax": r_ex / amax,
                                    "hip_vs_exact_over_max_g": d_ex / scale, "ref32_vs_exact_over_max_g": r_ex / scale,
                                    "hip_vs_ref32_over_max_g": d_32 / scale, "cancellation_Amax_over_max_g": amax / scale}
                row["K_hip_worst"] = max(row["K_hip_worst"], d_ex / amax)
                row["K_ref32_worst"] = max(row["K_ref32_worst"], r_ex / amax)
                row["grad_hip_vs_exact_rel_max_worst"] = max(row["grad_hip_vs_exact_rel_max_worst"], d_ex / scale)
                row["grad_ref32_vs_exact_rel_max_worst"] = max(row["grad_ref32_vs_exact_rel_max_worst"], r_ex / scale)
                row["grad_hip_vs_ref32_rel_max_worst"] = max(row["grad_hip_vs_ref32_rel_max_worst"], d_32 / scale)
                if d_ex > 1e-5 * amax:
                    bad.append((full, d_ex / amax, d_ex / scale))
        assert off == g_hip.numel()
        row["grad_bad"], row["grad_per_tensor"] = bad, per_tensor
        # Adam on the SAME gradient (the kernel's): isolates cnerf_adam_step along the trajectory
        p_same, m_same, v_same = p0.clone(), m0.clone(), v0.clone()
        O.adam_step(p_same, g_hip, m_same, v_same, step_i, lr_i)
        row["adam_same_grad_dw"] = float((p_same - p1_hip).abs().max())
        row["adam_same_grad_dm_rel"] = float((m_same - m1_hip).abs().max() / m_same.abs().max())
        row["adam_same_grad_dv_rel"] = float((v_same - v1_hip).abs().max() / v_same.abs().max())
        # end to end: the reference-arithmetic gradient through the oracle's Adam
        p_or, m_or, v_or = p0.clone(), m0.clone(), v0.clone()
        O.adam_step(p_or, g_32, m_or, v_or, step_i, lr_i)
        dw = (p_or - p1_hip).abs()
        bound = 2e-6 + 2.0 * _adam_first_order_bound(dg_flat, g_32, m0, v0, step_i, lr_i)
        row["dw_max"] = float(dw.max())
        row["dw_frac_beyond_2e-6"] = float((dw > 2e-6).float().mean())
        row["dw_beyond_conditioning_bound"] = int((dw > bound).sum())
        report.append(row)
        print("  " + " ".join(f"{k}={v:.3e}" if isinstance(v, float) else f"{k}={v}" for k, v in row.items()
                              if k != "grad_per_tensor" and k != "grad_bad"), flush=True)
        del ex, r32, fr, osd
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "teacher_forced_c2.json"), "w") as f:
        json.dump({"what": "tests/test_gpu_training_parity.py::test_c2_teacher_forced_training_steps", "seed": seed,
                   "snapshots": list(SNAPSHOTS), "rows": report}, f, indent=1)
    for row in report:
        s_ = f"step {row['step']}: "
        assert row["loss_rel_f32"] <= 1e-6 and row["loss_rel_f64"] <= 1e-6, s_ + "loss on the kernel's branch"
        assert row["loss_rel_free"] <= 1e-5, s_ + "loss vs the free-running oracle"
        assert (row["z_fine_own_vs_kernel_median"] <= 2e-6 * P.FAR and row["z_fine_own_vs_kernel_p99"] <= 2e-5 * P.FAR
                and row["z_fine_own_vs_kernel_max"] <= 1e-3 * P.FAR), s_ + "fine depths"
        for tag, zb in (("f64", 1e-4), ("f32", 1e-5)):     # (float64 pre-activations sit one fp32 round-off of a 256-term dot
            assert row[f"relu_flip_max_abs_z_{tag}"] < zb, s_ + "ReLU pattern differs away from zero"      # product away)
            assert row[f"tail_substitution_max_abs_sigma_{tag}"] < 1e-5, s_ + "sigma_last branch differs away from zero"
        assert row["relu_flip_frac"] < 1e-5, s_ + "too many ReLU pattern differences"
        assert not row["grad_bad"], s_ + f"gradient [tensor, |d| / A_max, |d| / max|g|]: {row['grad_bad']}"
        for lv in ("coarse", "fine"):
            assert row[f"d_raw_{lv}_hip_vs_exact_rel_l2"] <= 2e-5, s_ + f"d loss / d raw ({lv})"
        assert row["adam_same_grad_dw"] <= 2e-7 + 2 * 6e-8 * 4.0, s_ + "Adam kernel on the same gradient"
        assert row["adam_same_grad_dm_rel"] <= 1e-6 and row["adam_same_grad_dv_rel"] <= 1e-6, s_ + "Adam moments"
        assert row["dw_beyond_conditioning_bound"] == 0 and row["dw_frac_beyond_2e-6"] <= 1e-3, s_ + "updated weights"
    assert regime_rays > 0, "no snapshot had rays with |sigma_last| < 1e-2 (the R:281 regime the divergence probe blames)"


# ------------------------------------------------------------------------------------------------ the ConsistentNeRF step
class _C3Loss:
    """The loss of one level of the ConsistentNeRF step in the ORACLE's functions (V:1645-1648 masked rgb, V:1678-1726 patch term,
    V:1737 masked depth; accumulated in the reference's order): what run_nerf_view.render_loss folds into the compositing launches."""

    def __init__(self, mask, prior, mono, far, coef=0.2, rgb_w=1.0, depth_w=0.1, patch_w=0.001):
        self.mask, self.prior, self.mono, self.far = mask, prior, mono, far
        self.coef, self.rgb_w, self.depth_w, self.patch_w = coef, rgb_w, depth_w, patch_w

    def to(self, device):
        return _C3Loss(self.mask.to(device), self.prior.to(device), self.mono.to(device), self.far, self.coef, self.rgb_w,
                       self.depth_w, self.patch_w)

    def __call__(self, k, comp, tg, dtype):
        rgb, depth = comp[0], comp[4]
        level = self.rgb_w * O.masked_rgb_loss(rgb, tg, self.mask, self.coef)
        level = level + self.patch_w * O.patch_depth_loss(depth, self.mono.to(dtype), 4, 256)
        return level + self.depth_w * O.masked_depth_loss(depth, self.prior.to(dtype), self.mask, self.far)


C3_SNAPSHOTS = (0, 50)


def test_c3_teacher_forced_step(dev, monkeypatch):
    """the chaos-free step parity of the C2 test above for ConsistentNeRF's OWN step.  The HIP path trains the
    C3 rig of bench.py (three 378x504 views, hard masks from the cross-view warp, 4096 random + 1024 patch rays per step, 64 + 128
    samples, D=8/W=256, pytest streams) through the product surface — raybank.sample_patch_rays (one launch), run_nerf_view.render_loss
    (masked rgb + depth on both levels + the monocular patch term folded into compositing), run_nerf.backward, FusedAdam with the
    value clip 0.1 (V:1983) — and at steps {0, 50} ONE oracle step is run from the snapshotted state on the kernel's branch (its
    depths, ReLU sign bits, sigma signs), in float64 (exact) and in fp32 on the CPU (the reference arithmetic): O.query -> O.composite
    -> O.masked_rgb_loss / O.masked_depth_loss / O.patch_depth_loss -> autograd -> O.adam_step(clip=0.1).

    Same bounds as the C2 test: loss 1e-6 relative; every gradient tensor within 1e-5 * A_max of the exact one (A = the absolute mass
    of the element's sum; the reference's own fp32 figure reported beside it); d loss / d raw — here carrying the depth-loss and the
    patch-term gradients into `raw` (g_depth of V:1737 at 5120 rays had no oracle comparison before) — 2e-5 rel-L2; pattern
    differences only at the discontinuities; the Adam kernel with the clip on the same gradient 2e-7; the set of clipped elements
    identical up to elements within 1e-5 * A of the clip value."""
    sys.path.insert(0, ROOT)
    import json
    import bench
    from consistentnerf_amd import ops, raybank as RB, run_nerf as R, run_nerf_view as V
    sc = bench.c3_scene(dev)
    H, W, K, far, kw, opt = sc["H"], sc["W"], sc["K"], sc["far"], sc["kw"], sc["opt"]
    nets = [kw["network_fn"], kw["network_fine"]]
    names = [[n for n, _ in m.named_parameters()] for m in nets]
    sizes = [[p.numel() for _, p in m.named_parameters()] for m in nets]
    seen = {}
    orig_pair = ops.mlp_backward_pair

    def spy(*a, **k):
        seen["args"] = a
        return orig_pair(*a, **k)
    monkeypatch.setattr(ops, "mlp_backward_pair", spy)
    ncfg = O.NetCfg(8, 256, output_ch=5)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    torch.manual_seed(5)
    np.random.seed(5)
    report = []
    for i in range(max(C3_SNAPSHOTS) + 1):
        v = i % 3
        snap = i in C3_SNAPSHOTS
        starts = RB.draw_patch_starts(H, W, 4, 16)
        rays, target, sel, (d_prior, m, mono_s) = RB.sample_patch_rays(
            sc["img_t"][v], sc["poses"][v], H, W, K, 4096, starts, extras=(sc["dep_t"][v], sc["msk_t"][v], sc["mono_t"][v]),
            render_kwargs=kw)
        if snap:
            torch.cuda.synchronize()
            w0 = [{k: t.detach().cpu().clone() for k, t in mdl.state_dict().items()} for mdl in nets]
            m0, v0, p0 = opt.exp_avg.cpu().clone(), opt.exp_avg_sq.cpu().clone(), opt.flat_param.cpu().clone()
            lr_i, step_i = opt.param_groups[0]["lr"], opt._step + 1
        loss, terms, rgb, disp, acc, depth, extras = V.render_loss(
            H, W, K, target, mask=m, depth_prior=d_prior, chunk=32768, rays=rays, hardmask_coef=0.2, depth_w=0.1, mono=mono_s,
            patch_num=4, patch_size=16, patch_w=0.001, retraw=True, pytest=True, _debug=snap, **kw)
        opt.zero_grad()
        seen.pop("args", None)
        R.backward(loss)
        if snap:
            assert "args" in seen, "the step did not take the merged coarse+fine backward"
            fs, fp, fg, fB, fS, fst, fgr, cs, cp, cg, cB, cS, cst, cgr = seen.pop("args")
            g_hip = opt.flat_grad.cpu().clone()
            masks_f, masks_c = _masks_from_stash(fst, fB * fS), _masks_from_stash(cst, cB * cS)
            d_raw_hip = (cg.detach().cpu().reshape(cB, cS, -1), fg.detach().cpu().reshape(fB, fS, -1))
            del fst, cst, fg, cg
            z_f, z_c = extras["_z_vals"].cpu(), extras["_z_coarse"].cpu()
            raw_k = (extras["_raw_coarse"].detach().cpu(), extras["raw"].detach().cpu())
            rows_c = rays._cnerf_packed.rows.cpu()
        opt.step()
        for g_ in opt.param_groups:
            g_["lr"] = 5e-4 * (0.1 ** (i / 250000))
        if not snap:
            continue
        torch.cuda.synchronize()
        loss_hip, p1_hip = float(loss), opt.flat_param.cpu().clone()
        lf = _C3Loss(m.cpu(), d_prior.cpu(), mono_s.cpu(), far)
        common = (w0, names, rows_c, target.cpu(), z_c, z_f, masks_c, masks_f, raw_k, ncfg)
        ex = _oracle_step(torch.float64, *common, want_abs=True, device=dev, loss_fn=lf)
        r32 = _oracle_step_cpu32(*common, loss_fn_factory=lambda: lf)
        del masks_c, masks_f
        row = {"step": i, "cpu_ref32_retried": r32["retried"], "loss_hip": loss_hip, "loss_oracle_f32": r32["loss"], "loss_oracle_f64": ex["loss"],
               "loss_rel_f32": abs(loss_hip - r32["loss"]) / abs(r32["loss"]), "loss_rel_f64": abs(loss_hip - ex["loss"]) / abs(ex["loss"]),
               "terms_hip": {k: float(t) for k, t in terms.items()}}
        for tag, res in (("f64", ex), ("f32", r32)):
            row[f"relu_flips_{tag}"] = int(sum(n for n, _ in res["flips"]))
            row[f"relu_flip_max_abs_z_{tag}"] = max(z for _, z in res["flips"])
            row[f"sigma_sign_substitutions_{tag}"] = int(sum(t[1] for t in res["tail"]))
            row[f"tail_substitution_max_abs_sigma_{tag}"] = max(t[2] for t in res["tail"])
        row["relu_flip_frac"] = max(row["relu_flips_f64"], row["relu_flips_f32"]) / (5120 * 256 * (8 * 256 + 128))
        for k, tag in ((0, "coarse"), (1, "fine")):
            e_, h_, r_ = ex["d_raw"][k], d_raw_hip[k].double(), r32["d_raw"][k].double()
            row[f"d_raw_{tag}_hip_vs_exact_rel_l2"] = float((h_ - e_).norm() / e_.norm())
            row[f"d_raw_{tag}_ref32_vs_exact_rel_l2"] = float((r_ - e_).norm() / e_.norm())
        g_ex, g_32, A = ex["grad"], r32["grad"], ex["A"]
        off, bad, dg_flat = 0, [], torch.zeros_like(g_hip)
        row["K_hip_worst"] = row["K_ref32_worst"] = row["grad_hip_vs_exact_rel_max_worst"] = row["grad_ref32_vs_exact_rel_max_worst"] = 0.0
        for k in (0, 1):
            for nme, n in zip(names[k], sizes[k]):
                a, b, e, aa = g_hip[off:off + n].double(), g_32[off:off + n].double(), g_ex[off:off + n], A[off:off + n]
                dg_flat[off:off + n] = float((a - b).abs().max())
                off += n
                scale = float(e.abs().max())
                if scale == 0:
                    assert float(a.abs().max()) == 0 and float(b.abs().max()) == 0
                    continue
                amax = float(aa.max())
                d_ex, r_ex = float((a - e).abs().max()), float((b - e).abs().max())
                row["K_hip_worst"] = max(row["K_hip_worst"], d_ex / amax)
                row["K_ref32_worst"] = max(row["K_ref32_worst"], r_ex / amax)
                row["grad_hip_vs_exact_rel_max_worst"] = max(row["grad_hip_vs_exact_rel_max_worst"], d_ex / scale)
                row["grad_ref32_vs_exact_rel_max_worst"] = max(row["grad_ref32_vs_exact_rel_max_worst"], r_ex / scale)
                if d_ex > 1e-5 * amax:
                    bad.append((("coarse." if k == 0 else "fine.") + nme, d_ex / amax, d_ex / scale))
        assert off == g_hip.numel()
        row["grad_bad"] = bad
        # the value clip (V:1983): which elements it touches, kernel vs exact gradient
        clipped_hip, clipped_ex = g_hip.abs() > 0.1, g_ex.abs() > 0.1
        diff = clipped_hip != clipped_ex
        row["clipped_elements_hip"], row["clipped_elements_exact"] = int(clipped_hip.sum()), int(clipped_ex.sum())
        row["clip_set_differences"] = int(diff.sum())
        row["clip_set_difference_max_distance"] = float((g_ex[diff].abs() - 0.1).abs().max()) if bool(diff.any()) else 0.0
        p_same, m_same, v_same = p0.clone(), m0.clone(), v0.clone()
        O.adam_step(p_same, g_hip, m_same, v_same, step_i, lr_i, clip=0.1)
        row["adam_same_grad_dw"] = float((p_same - p1_hip).abs().max())
        p_or, m_or, v_or = p0.clone(), m0.clone(), v0.clone()
        O.adam_step(p_or, g_32, m_or, v_or, step_i, lr_i, clip=0.1)
        dw = (p_or - p1_hip).abs()
        bound = 2e-6 + 2.0 * _adam_first_order_bound(dg_flat, g_32.clamp(-0.1, 0.1), m0, v0, step_i, lr_i)
        row["dw_max"], row["dw_frac_beyond_2e-6"] = float(dw.max()), float((dw > 2e-6).float().mean())
        row["dw_beyond_conditioning_bound"] = int((dw > bound).sum())
        report.append(row)
        print("  " + " ".join(f"{k}={t:.3e}" if isinstance(t, float) else f"{k}={t}" for k, t in row.items() if k != "grad_bad"), flush=True)
        del ex, r32
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "teacher_forced_c3.json"), "w") as f:
        json.dump({"what": "tests/test_gpu_training_parity.py::test_c3_teacher_forced_step", "snapshots": list(C3_SNAPSHOTS),
                   "rows": report}, f, indent=1)
    for row in report:
        s_ = f"step {row['step']}: "
        assert row["loss_rel_f32"] <= 1e-6 and row["loss_rel_f64"] <= 1e-6, s_ + "loss on the kernel's branch"
        for tag, zb in (("f64", 1e-4), ("f32", 1e-5)):
            assert row[f"relu_flip_max_abs_z_{tag}"] < zb, s_ + "ReLU pattern differs away from zero"
            assert row[f"tail_substitution_max_abs_sigma_{tag}"] < 1e-5, s_ + "sigma branch differs away from zero"
        assert row["relu_flip_frac"] < 1e-5, s_ + "too many ReLU pattern differences"
        assert not row["grad_bad"], s_ + f"gradient [tensor, |d| / A_max, |d| / max|g|]: {row['grad_bad']}"
        for lv in ("coarse", "fine"):
            assert row[f"d_raw_{lv}_hip_vs_exact_rel_l2"] <= 2e-5, s_ + f"d loss / d raw ({lv})"
        assert row["clip_set_differences"] <= 4 and row["clip_set_difference_max_distance"] <= 1e-5, s_ + "clipped element set"
        assert row["adam_same_grad_dw"] <= 2e-7 + 2 * 6e-8 * 4.0, s_ + "Adam kernel (clip 0.1) on the same gradient"
        assert row["dw_beyond_conditioning_bound"] == 0 and row["dw_frac_beyond_2e-6"] <= 1e-3, s_ + "updated weights"


# ------------------------------------------------------------------------------------------------ the in-loop consistency step
class _SSLoss:
    """The loss of one `--ss_loss --with_depth_loss` step in the ORACLE's functions, on the combined batch of the one-render form
    (rows [0, N) the primary rays, [N, N + M) the warped rays): the second render's four terms (VT:930-938: O.mse on rgb / depth of the
    fine, then of the coarse level) and then the primary render's four coin-gated terms (VT:941-969: O.ss_primary_losses, pinned on the
    reference's fixture `ssloss_primary` in the CPU suite), accumulated in the reference's order.  _oracle_step calls it per level
    (coarse first): the coarse level's maps are kept and the whole loss is formed at the fine level."""

    def __init__(self, N, mask_bound, mask, prior, depth_ref_tgt, coins):
        self.N, self.mask_bound, self.mask, self.prior, self.dref, self.coins = N, mask_bound, mask, prior, depth_ref_tgt, coins
        self.c0 = None

    def to(self, device):
        return _SSLoss(self.N, self.mask_bound.to(device), self.mask.to(device), self.prior.to(device), self.dref.to(device), self.coins)

    def __call__(self, k, comp, tg, dtype):
        if k == 0:
            self.c0 = comp
            return 0.0
        N = self.N
        rgb, depth, rgb0, depth0 = comp[0], comp[4], self.c0[0], self.c0[4]
        dref = self.dref.to(dtype)
        loss = O.mse(rgb[N:], tg[N:])
        loss = loss + O.mse(depth[N:], dref)
        loss = loss + O.mse(rgb0[N:], tg[N:])
        loss = loss + O.mse(depth0[N:], dref)
        lp, _, _ = O.ss_primary_losses(rgb[:N], depth[:N], rgb0[:N], depth0[:N], tg[:N], self.prior.to(dtype), self.mask_bound, self.mask,
                                       True, list(self.coins))
        return loss + lp


C3SS_SNAPSHOTS = (0, 50)


def test_c3ss_teacher_forced_step(dev, monkeypatch):
    """the chaos-free step parity of the C2 / C3 tests for the IN-LOOP CONSISTENCY step (a15, VT:899-969) at the
    bench's `c3_ss` size.  The HIP path trains bench.c3_ss_step_fn's rig through the product surface — raybank.sample_patch_rays,
    run_nerf_view.ss_step_loss (ONE render of the 8192-row combined batch: 4096 primary rays + ~3700 warped rays + padding, the
    device-side live-row count, the two-segment folded loss), run_nerf.backward, FusedAdam with the value clip 0.1 — and at steps
    {0, 50} ONE oracle step is run from the snapshotted state on the kernel's branch:

        O.ss_block (VT:905-925; here its result is first compared with the step's own batch assembly: masks, compaction, threshold,
        targets bit for bit) -> O.query / O.composite of both levels on the live rows at the kernel's depths, ReLU sign bits and
        sigma signs -> the second render's four O.mse terms + O.ss_primary_losses with the step's coins -> autograd ->
        O.adam_step(clip = 0.1)

    in float64 (exact) and in fp32 on the CPU (the reference arithmetic).  Bounds of test_c3_teacher_forced_step: loss 1e-6 relative;
    every gradient tensor within 1e-5 * A_max of the exact one; d loss / d raw of the LIVE rows 2e-5 rel-L2 (and exactly zero on the
    padding rows); pattern differences only at the discontinuities; clipped element sets equal up to 1e-5; the Adam kernel 2e-7."""
    sys.path.insert(0, ROOT)
    import json
    import bench
    from consistentnerf_amd import ops, raybank as RB, run_nerf as R, run_nerf_view as V
    sc = bench.c3_scene(dev)
    H, W, K, kw, opt = sc["H"], sc["W"], sc["K"], sc["kw"], sc["opt"]
    nets = [kw["network_fn"], kw["network_fine"]]
    names = [[n for n, _ in m.named_parameters()] for m in nets]
    sizes = [[p.numel() for _, p in m.named_parameters()] for m in nets]
    seen = {}
    orig_pair = ops.mlp_backward_pair

    def spy(*a, **k):
        seen["args"], seen["kw"] = a, k
        return orig_pair(*a, **k)
    monkeypatch.setattr(ops, "mlp_backward_pair", spy)
    ncfg = O.NetCfg(8, 256, output_ch=5)
    torch.set_num_threads(max(1, min(32, len(os.sched_getaffinity(0)))))
    torch.manual_seed(7)
    np.random.seed(7)
    rs = np.random.RandomState(3)
    N, report = 4096, []
    for i in range(max(C3SS_SNAPSHOTS) + 1):
        v, r = i % 3, (i + 1 + int(rs.randint(0, 2))) % 3
        snap = i in C3SS_SNAPSHOTS
        rays, target, sel_px, (d_prior,) = RB.sample_patch_rays(sc["img_t"][v], sc["poses"][v], H, W, K, N, None,
                                                                extras=(sc["dep_ss_t"][v],), render_kwargs=kw)
        coins = [int(c) for c in rs.randint(0, 2, 4)]
        if snap:
            coins = [1, 1, 0, 1] if i == 0 else [0, 1, 1, 0]       # (both branches of every term across the two snapshots)
            torch.cuda.synchronize()
            w0 = [{k: t.detach().cpu().clone() for k, t in mdl.state_dict().items()} for mdl in nets]
            m0, v0, p0 = opt.exp_avg.cpu().clone(), opt.exp_avg_sq.cpu().clone(), opt.flat_param.cpu().clone()
            lr_i, step_i = opt.param_groups[0]["lr"], opt._step + 1
        loss, info = V.ss_step_loss(H, W, K, rays, target, d_prior, sc["poses"][r], sc["img_t"][r], sc["dep_ss_t"][r],
                                    dict(kw, pytest=True, _debug=snap), chunk=32768, occlusion_threshold=0.1, with_depth_loss=True,
                                    coins=coins, route="one_render")
        opt.zero_grad()
        seen.pop("args", None)
        R.backward(loss)
        if snap:
            bf3 = os.environ.get("CNERF_TRAIN_PRECISION", "fp32") == "bf16x3"       # (the opt-in arithmetic has no live-row gate)
            assert "args" in seen and (bf3 or seen["kw"].get("live") is info["live"]), "the step did not take the merged live backward"
            fs, fp, fg, fB, fS, fst, fgr, cs, cp, cg, cB, cS, cst, cgr = seen.pop("args")
            assert fB == cB == 2 * N
            live = int(info["live"].item())
            M = live - N
            g_hip = opt.flat_grad.cpu().clone()
            # (tile-major stash: the first live * S points are its first live * S * s_rows floats; the rest was never written)
            masks_f = _masks_from_stash(fst[:live * fS * (fst.numel() // (fB * fS))], live * fS)
            masks_c = _masks_from_stash(cst[:live * cS * (cst.numel() // (cB * cS))], live * cS)
            d_raw_all = (cg.detach().reshape(cB, cS, -1), fg.detach().reshape(fB, fS, -1))
            pad_d_raw = max(float(d[live:].abs().max()) for d in d_raw_all)
            d_raw_hip = tuple(d[:live].cpu() for d in d_raw_all)
            del fst, cst, fg, cg, d_raw_all
            both = lambda k: torch.cat([info["extras"][k], info["extras_ref"][k]], 0)[:live].detach().cpu()   # noqa: E731
            z_f, z_c = both("_z_vals"), both("_z_coarse")
            raw_k = (both("_raw_coarse"), both("raw"))
            rows_c, tgt2 = info["rows"][:live].cpu(), info["_target2"][:live].cpu()
            # the step's own batch assembly against the oracle's block (pinned on the reference's fixture)
            blk = O.ss_block(rays[0].cpu(), rays[1].cpu(), d_prior.cpu(), torch.from_numpy(sc["poses"][r]).float(),
                             torch.from_numpy(np.asarray(K, np.float32)), sc["img_t"][r].cpu().permute(2, 0, 1), sc["dep_ss_t"][r].cpu(), 0.1)
            hv = V.ss_host_view(info)
            ok_block = (M == int(blk["mask_bound"].sum()) and torch.equal(hv["mask_bound"].cpu(), blk["mask_bound"])
                        and torch.equal(hv["mask"].cpu(), blk["mask"]) and float(hv["threshold"]) == np.float32(blk["thr"])
                        and torch.equal(info["sel"].cpu(), blk["sel"]) and torch.equal(tgt2[N:], blk["rgb_target_ref"][0].t())
                        and torch.equal(info["_prior2"][N:live].cpu(), blk["rays_depth_ref"].reshape(-1)))
            rays_err = float((hv["batch_rays_ref"].cpu() - blk["rays_ref"]).abs().max())
        opt.step()
        if not snap:
            continue
        torch.cuda.synchronize()
        loss_hip, p1_hip = float(loss), opt.flat_param.cpu().clone()
        # (the oracle's loss takes the ORACLE's masks / targets: they were just checked against the step's own, bit for bit)
        mk = lambda: _SSLoss(N, blk["mask_bound"], blk["mask"], d_prior.cpu(), blk["rays_depth_ref"].reshape(-1), coins)  # noqa: E731
        common = (w0, names, rows_c, tgt2, z_c, z_f, masks_c, masks_f, raw_k, ncfg)
        ex = _oracle_step(torch.float64, *common, want_abs=True, device=dev, loss_fn=mk())
        r32 = _oracle_step_cpu32(*common, loss_fn_factory=mk)
        del masks_c, masks_f
        row = {"step": i, "cpu_ref32_retried": r32["retried"], "coins": coins, "live_rows": live, "M": M, "threshold": float(hv["threshold"]), "block_equals_oracle": bool(ok_block),
               "rays_ref_max_err": rays_err, "padding_d_raw_max": pad_d_raw,
               "loss_hip": loss_hip, "loss_oracle_f32": r32["loss"], "loss_oracle_f64": ex["loss"],
               "loss_rel_f32": abs(loss_hip - r32["loss"]) / abs(r32["loss"]), "loss_rel_f64": abs(loss_hip - ex["loss"]) / abs(ex["loss"]),
               "terms_hip": {k: float(t) for k, t in info["terms"].items()}}
        for tag, res in (("f64", ex), ("f32", r32)):
            row[f"relu_flips_{tag}"] = int(sum(n for n, _ in res["flips"]))
            row[f"relu_flip_max_abs_z_{tag}"] = max(z for _, z in res["flips"])
            row[f"sigma_sign_substitutions_{tag}"] = int(sum(t[1] for t in res["tail"]))
            row[f"tail_substitution_max_abs_sigma_{tag}"] = max(t[2] for t in res["tail"])
        row["relu_flip_frac"] = max(row["relu_flips_f64"], row["relu_flips_f32"]) / (live * 256 * (8 * 256 + 128))
        for k, tag in ((0, "coarse"), (1, "fine")):
            e_, h_, r_ = ex["d_raw"][k], d_raw_hip[k].double(), r32["d_raw"][k].double()
            row[f"d_raw_{tag}_hip_vs_exact_rel_l2"] = float((h_ - e_).norm() / e_.norm())
            row[f"d_raw_{tag}_ref32_vs_exact_rel_l2"] = float((r_ - e_).norm() / e_.norm())
        g_ex, g_32, A = ex["grad"], r32["grad"], ex["A"]
        off, bad, dg_flat = 0, [], torch.zeros_like(g_hip)
        row["K_hip_worst"] = row["K_ref32_worst"] = row["grad_hip_vs_exact_rel_max_worst"] = row["grad_ref32_vs_exact_rel_max_worst"] = 0.0
        for k in (0, 1):
            for nme, n in zip(names[k], sizes[k]):
                a, b, e, aa = g_hip[off:off + n].double(), g_32[off:off + n].double(), g_ex[off:off + n], A[off:off + n]
                dg_flat[off:off + n] = float((a - b).abs().max())
                off += n
                scale = float(e.abs().max())
                if scale == 0:
                    assert float(a.abs().max()) == 0 and float(b.abs().max()) == 0
                    continue
                amax = float(aa.max())
                d_ex, r_ex = float((a - e).abs().max()), float((b - e).abs().max())
                row["K_hip_worst"] = max(row["K_hip_worst"], d_ex / amax)
                row["K_ref32_worst"] = max(row["K_ref32_worst"], r_ex / amax)
                row["grad_hip_vs_exact_rel_max_worst"] = max(row["grad_hip_vs_exact_rel_max_worst"], d_ex / scale)
                row["grad_ref32_vs_exact_rel_max_worst"] = max(row["grad_ref32_vs_exact_rel_max_worst"], r_ex / scale)
                if d_ex > 1e-5 * amax:
                    bad.append((("coarse." if k == 0 else "fine.") + nme, d_ex / amax, d_ex / scale))
        assert off == g_hip.numel()
        row["grad_bad"] = bad
        clipped_hip, clipped_ex = g_hip.abs() > 0.1, g_ex.abs() > 0.1
        diff = clipped_hip != clipped_ex
        row["clipped_elements_hip"], row["clipped_elements_exact"] = int(clipped_hip.sum()), int(clipped_ex.sum())
        row["clip_set_differences"] = int(diff.sum())
        row["clip_set_difference_max_distance"] = float((g_ex[diff].abs() - 0.1).abs().max()) if bool(diff.any()) else 0.0
        p_same, m_same, v_same = p0.clone(), m0.clone(), v0.clone()
        O.adam_step(p_same, g_hip, m_same, v_same, step_i, lr_i, clip=0.1)
        row["adam_same_grad_dw"] = float((p_same - p1_hip).abs().max())
        p_or, m_or, v_or = p0.clone(), m0.clone(), v0.clone()
        O.adam_step(p_or, g_32, m_or, v_or, step_i, lr_i, clip=0.1)
        dw = (p_or - p1_hip).abs()
        bound = 2e-6 + 2.0 * _adam_first_order_bound(dg_flat, g_32.clamp(-0.1, 0.1), m0, v0, step_i, lr_i)
        row["dw_max"], row["dw_frac_beyond_2e-6"] = float(dw.max()), float((dw > 2e-6).float().mean())
        row["dw_beyond_conditioning_bound"] = int((dw > bound).sum())
        report.append(row)
        print("  " + " ".join(f"{k}={t:.3e}" if isinstance(t, float) else f"{k}={t}" for k, t in row.items() if k != "grad_bad"), flush=True)
        del ex, r32
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "teacher_forced_c3ss.json"), "w") as f:
        json.dump({"what": "tests/test_gpu_training_parity.py::test_c3ss_teacher_forced_step", "snapshots": list(C3SS_SNAPSHOTS),
                   "rows": report}, f, indent=1)
    for row in report:
        s_ = f"step {row['step']}: "
        assert row["block_equals_oracle"] and row["rays_ref_max_err"] <= 1e-6 * 4.0, s_ + "batch assembly vs O.ss_block"
        assert row["padding_d_raw_max"] == 0.0, s_ + "padding rows must get a zero gradient"
        assert row["loss_rel_f32"] <= 1e-6 and row["loss_rel_f64"] <= 1e-6, s_ + "loss on the kernel's branch"
        for tag, zb in (("f64", 1e-4), ("f32", 1e-5)):
            assert row[f"relu_flip_max_abs_z_{tag}"] < zb, s_ + "ReLU pattern differs away from zero"
            assert row[f"tail_substitution_max_abs_sigma_{tag}"] < 1e-5, s_ + "sigma branch differs away from zero"
        assert row["relu_flip_frac"] < 1e-5, s_ + "too many ReLU pattern differences"
        assert not row["grad_bad"], s_ + f"gradient [tensor, |d| / A_max, |d| / max|g|]: {row['grad_bad']}"
        for lv in ("coarse", "fine"):
            assert row[f"d_raw_{lv}_hip_vs_exact_rel_l2"] <= 2e-5, s_ + f"d loss / d raw ({lv})"
        assert row["clip_set_differences"] <= 4 and row["clip_set_difference_max_distance"] <= 1e-5, s_ + "clipped element set"
        assert row["adam_same_grad_dw"] <= 2e-7 + 2 * 6e-8 * 4.0, s_ + "Adam kernel (clip 0.1) on the same gradient"
        assert row["dw_beyond_conditioning_bound"] == 0 and row["dw_frac_beyond_2e-6"] <= 1e-3, s_ + "updated weights"
