#!/usr/bin/env python3
"""Why a seed's HIP and oracle trajectories part early at the true C2 batch size (profiles/r03_psnr_c2_seed1_divergence.txt).
   python scripts/psnr_divergence_probe.py <step> [grads]        (MI355X; PROBE_SEED=1)
Trains the HIP path for <step> steps (scripts/psnr_parity.py's C2-size scene, seed, batches, RNG hook), then evaluates THAT step's
forward with the HIP path and with the CPU oracle on the SAME weights and batch: losses, per-ray differences, fine-sample
differences, and the distribution of the last sample's sigma — R:281 gives the last interval a width of 1e10, so the last
sample's alpha is 0 or 1 by the SIGN of its sigma, and a ray whose sigma_last sits within the trajectory difference of zero flips
between acc ~ 0.06 and acc = 1.  With `grads`: also the gradients of both sides on those weights."""
import sys, os, json, argparse, tempfile, numpy as np, torch
sys.path.insert(0, "scripts"); sys.path.insert(0, "."); sys.path.insert(0, "tests/golden")
import psnr_parity as P
from consistentnerf_amd import run_nerf as R
from oracle import nerf_oracle as O
P.set_size("c2")
SEED = int(os.environ.get("PROBE_SEED", "1"))
STEP = int(sys.argv[1]) if len(sys.argv) > 1 else 17
dev = torch.device("cuda:0")
K, bank, target, test_rays, test_rgb, sds = P.scene(SEED)
args = argparse.Namespace(multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=128, netdepth=8, netwidth=256,
    netdepth_fine=8, netwidth_fine=256, netchunk=1024 * 64, lrate=P.LRATE, basedir=tempfile.mkdtemp(), expname="p", ft_path=None,
    no_reload=True, perturb=1.0, N_samples=64, white_bkgd=False, raw_noise_std=0.0, dataset_type="dtu", no_ndc=True, lindisp=False)
kw, kw_test, _, grad_vars, opt = R.create_nerf(args)
kw["network_fn"].load_state_dict({k: torch.from_numpy(v) for k, v in sds[0].items()})
kw["network_fine"].load_state_dict({k: torch.from_numpy(v) for k, v in sds[1].items()})
kw.update(near=P.NEAR, far=P.FAR)
bank_d, target_d = bank.to(dev), target.to(dev)
for i in range(STEP + 1):
    lo, hi = P.batch_bounds(i, bank.shape[0])
    rb, tg = bank_d[lo:hi], target_d[lo:hi]
    if i == STEP:
        break
    rgb, disp, acc, ex = R.render(P.H, P.W, K, chunk=32768, rays=torch.stack([rb[:, 0:3], rb[:, 3:6]]), retraw=True, pytest=True, **kw)
    opt.zero_grad()
    loss = R.img2mse(rgb, tg) + R.img2mse(ex["rgb0"], tg)
    loss.backward(); opt.step()
    for g_ in opt.param_groups:
        g_["lr"] = P.LRATE * (0.1 ** (i / (P.LRATE_DECAY * 1000)))
# step STEP: forward only, HIP vs oracle on the SAME weights
with torch.no_grad():
    kk = {k: v for k, v in kw.items() if k not in ("near", "far", "ndc", "lindisp", "use_viewdirs")}
    out = R.render_rays(rb, _with_depth=True, _debug=True, retraw=True, pytest=True, lindisp=False, **kk)
sd_c = {k: v.detach().cpu() for k, v in kw["network_fn"].state_dict().items()}
sd_f = {k: v.detach().cpu() for k, v in kw["network_fine"].state_dict().items()}
net, cfg = O.NetCfg(8, 256, output_ch=5), O.RenderCfg(64, 128, 1.0)
with torch.no_grad():
    ref = O.render_rays_pytest(rb.cpu(), sd_c, sd_f, net, cfg)
tgc = tg.cpu()
lh = float(((out["rgb_map"].cpu() - tgc) ** 2).mean() + ((out["rgb0"].cpu() - tgc) ** 2).mean())
lo_ = float(((ref["rgb_map"] - tgc) ** 2).mean() + ((ref["rgb0"] - tgc) ** 2).mean())
print("step", STEP, "loss hip", lh, "oracle (same weights)", lo_)
for k in ("rgb0", "depth0", "rgb_map", "depth_map"):
    d = (out[k].cpu() - ref[k]).abs()
    d = d.reshape(d.shape[0], -1).max(1).values
    print(k, "max|d|", float(d.max()), "rays with |d| > 1e-3:", int((d > 1e-3).sum()), "> 1e-2:", int((d > 1e-2).sum()))
dz = (out["_z_vals"].cpu() - ref["z_vals"]).abs().max(1).values
print("z_vals: rays with |d| > 1e-4:", int((dz > 1e-4).sum()), "max", float(dz.max()))
bad = torch.nonzero(((out["rgb0"].cpu() - ref["rgb0"]).abs().max(1).values > 1e-3)).flatten()[:5]
raw0_h = None
for b in bad.tolist():
    print("ray", b, "rgb0 hip", out["rgb0"][b].cpu().numpy(), "oracle", ref["rgb0"][b].numpy(), "acc0", float(out["acc0"][b]), float(ref["acc0"][b]))

idx = torch.nonzero(dz > 1e-4).flatten()
print("rays with z diff:", idx[:10].tolist(), "of", len(idx))
# coarse level of those rays: weights / sigma (oracle side recomputed here)
with torch.no_grad():
    r_c = rb.cpu()
    zc = out["_z_coarse"].cpu()
    pts = r_c[:, None, 0:3] + r_c[:, None, 3:6] * zc[..., None]
    raw_c = O.query(sd_c, pts, r_c[:, 8:11], net)
    rgb0, disp0, acc0, w0, depth0 = O.composite(raw_c, zc, r_c[:, 3:6])
wh = out["_weights"].cpu() if "_weights" in out else None
sig = raw_c[..., 3]
print("sigma(all rays): min %.3e max %.3e; fraction of samples with 0 < sigma*dist < 1e-5: %.4f" % (float(sig.min()), float(sig.max()),
      float(((sig > 0) & (sig * 0.04 < 1e-5)).float().mean())))
for b in idx[:4].tolist():
    print("ray", b, "oracle coarse weights: sum %.3e max %.3e; sigma min %.3e max %.3e; acc0 hip %.3e oracle %.3e" % (
        float(w0[b].sum()), float(w0[b].max()), float(sig[b].min()), float(sig[b].max()), float(out["acc0"][b]), float(acc0[b])))
good = torch.nonzero(dz < 1e-6).flatten()[:2].tolist()
for b in good:
    print("ok ray", b, "oracle coarse weights: sum %.3e max %.3e; sigma min %.3e max %.3e" % (float(w0[b].sum()), float(w0[b].max()), float(sig[b].min()), float(sig[b].max())))

sl = raw_c[:, -1, 3]
rf = out["raw"].cpu()[:, -1, 3]
for nm, v in (("coarse sigma_last", sl), ("fine sigma_last", rf)):
    print(nm, "fraction > 0: %.3f;  |sigma_last| < 1e-3: %d rays, < 1e-2: %d rays, < 1e-1: %d rays of %d; median |.| %.3e" % (
        float((v > 0).float().mean()), int((v.abs() < 1e-3).sum()), int((v.abs() < 1e-2).sum()), int((v.abs() < 1e-1).sum()), v.numel(), float(v.abs().median())))
if "grads" not in sys.argv:
    sys.exit(0)

# gradients at this step: HIP (flat_grad) vs oracle autograd on the same weights and batch
rgb, disp, acc, ex = R.render(P.H, P.W, K, chunk=32768, rays=torch.stack([rb[:, 0:3], rb[:, 3:6]]), retraw=True, pytest=True, **kw)
opt.zero_grad()
loss = R.img2mse(rgb, tg) + R.img2mse(ex["rgb0"], tg)
loss.backward()
names = [n for n, _ in kw["network_fn"].named_parameters()]
osd = [{k: v.clone().requires_grad_(k not in ("temp_rgb", "temp_depth", "depth_scale")) for k, v in sd.items()} for sd in (sd_c, sd_f)]
ref = O.render_rays_pytest(rb.cpu(), osd[0], osd[1], net, cfg)
lo2 = O.mse(ref["rgb_map"], tgc) + O.mse(ref["rgb0"], tgc)
params = [p for d in osd for p in d.values() if p.requires_grad]
grads = torch.autograd.grad(lo2, params, allow_unused=True)
it = iter(grads)
worst = []
for tag, mdl, d in (("coarse", kw["network_fn"], osd[0]), ("fine", kw["network_fine"], osd[1])):
    for k, p in mdl.named_parameters():
        if not d[k].requires_grad:
            continue
        g_o = next(it)
        g_h = p.grad.detach().cpu() if p.grad is not None else torch.zeros_like(p).cpu()
        if g_o is None:
            g_o = torch.zeros_like(g_h)
        sc = float(g_o.abs().max())
        dd = float((g_h - g_o).abs().max())
        worst.append((dd / max(sc, 1e-30), tag, k, sc, dd))
worst.sort(reverse=True)
for w in worst[:8]:
    print("grad rel-max diff %.3e  %s.%s  (max|g| %.3e, max|d| %.3e)" % w)
