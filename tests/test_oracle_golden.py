"""Pins oracle/nerf_oracle.py against the golden vectors captured from the reference
(tests/golden/make_golden.py).  CPU only.  Where the oracle repeats the reference's ATen op
sequence the comparison is bit-exact; gradients go through a differently-shaped autograd graph and
get a 1e-6-relative bound."""
import numpy as np
import pytest
import torch

import _inputs as I
from conftest import golden
from oracle import nerf_oracle as O

torch.set_num_threads(4)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def eq(a, b, name=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    assert a.shape == b.shape, (name, a.shape, b.shape)
    assert np.array_equal(a, b, equal_nan=True), f"{name}: max|d|={np.nanmax(np.abs(a.astype(np.float64)-b))}"


def close(a, b, rtol=2e-6, atol=1e-7, name=""):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol, err_msg=name, equal_nan=True)


def test_embed():
    g = golden("embed")
    eq(O.embed(T(g["x"]), 10), g["L10"])
    eq(O.embed(T(g["x"]), 4), g["L4"])


MLP_CASES = [("D8W256_vd", 8, 256, True, 5), ("D4W128_vd", 4, 128, True, 4),
             ("D4W128_novd", 4, 128, False, 5), ("D8W128_vd", 8, 128, True, 5)]


def check_grads(g, prefix_full, prefix_sum, sd, rtol=2e-5):
    for k, p in sd.items():
        gr = p.grad if p.grad is not None else torch.zeros_like(p)
        if prefix_full + k in g:
            ref = g[prefix_full + k]
            scale = max(np.abs(ref).max(), 1e-30)
            assert np.abs(gr.numpy() - ref).max() <= rtol * scale + 1e-9, k
        else:
            ref = g[prefix_sum + k + ".sub"]
            scale = max(np.abs(ref).max(), 1e-30)
            assert np.abs(gr.reshape(-1)[::61].numpy() - ref).max() <= rtol * scale + 1e-9, k
            a = g[prefix_sum + k + ".abssum"]
            assert abs(gr.double().abs().sum().item() - a) <= 1e-5 * max(a, 1e-30) + 1e-9, k


@pytest.mark.parametrize("tag,D,W,vd,och", MLP_CASES)
def test_query(tag, D, W, vd, och):
    g = golden("mlp_" + tag)
    sd = O.as_tensors(I.nerf_state_dict(D, W, 10, 4, och, vd, seed=11), requires_grad=True)
    cfg = O.NetCfg(D=D, W=W, use_viewdirs=vd, output_ch=och)
    raw = O.query(sd, T(g["pts"]), T(g["dirs"]) if vd else None, cfg)
    # GEMM blocking differs with the reference's netchunk / thread count -> ulp-level differences
    close(raw, g["raw"], rtol=1e-5, atol=3e-6, name="raw")
    (raw * T(g["G"])).sum().backward()
    check_grads(g, "grad.", "gs.", sd)


R2O = [("S64", 64, False, 0.0), ("S192", 192, False, 0.0), ("S64_white", 64, True, 0.0),
       ("S192_white_noise", 192, True, 1.0)]


@pytest.mark.parametrize("tag,S,white,noise", R2O)
def test_composite(tag, S, white, noise):
    g = golden("raw2outputs_" + tag)
    raw, z, d = I.raw2outputs_inputs(32, S, seed=S + int(white))
    rawt = T(raw).requires_grad_(True)
    nz = O.pytest_uniform((32, S)) * noise if noise > 0 else None
    rgb, disp, acc, w, depth = O.composite(rawt, T(z), T(d), nz, white)
    for a, k in ((rgb, "rgb_map"), (disp, "disp_map"), (acc, "acc_map"), (w, "weights"), (depth, "depth_map")):
        eq(a, g[k], k)
    if noise == 0:
        assert np.isnan(g["disp_map"][:2]).all()      # acc==0 rows: reference returns NaN
    loss = (rgb * T(g["g_rgb"])).sum() + (depth * T(g["g_depth"])).sum() + (acc * T(g["g_acc"])).sum()
    (d_raw,) = torch.autograd.grad(loss, rawt, retain_graph=True)
    close(d_raw, g["d_raw"], rtol=1e-5, atol=1e-7)
    (dd,) = torch.autograd.grad((disp[2:] * T(g["g_disp"][2:])).sum(), rawt)
    close(dd, g["d_raw_disp"], rtol=1e-5, atol=1e-7)


@pytest.mark.parametrize("tag", ["det", "rand"])
def test_sample_pdf(tag):
    g = golden("sample_pdf_" + tag)
    bins, weights = I.sample_pdf_inputs(256, 64, seed=7)
    samples, inds = O.sample_pdf(T(bins), T(weights), T(g["u"]))
    eq(inds, g["inds"], "inds")            # the bit-exact target
    eq(samples, g["samples"], "samples")
    if tag == "rand":
        eq(O.pytest_uniform((256, 128)), g["u"], "u stream")


def _bulk_u(tag, B=4096, Nf=128):
    """the u of the reference's call: linspace(0, 1, Nf) broadcast (det, H:224-226) or its pytest stream (H:227-229)"""
    if tag == "det":
        return torch.from_numpy(np.broadcast_to(np.linspace(0., 1., Nf), (B, Nf)).astype(np.float32).copy())
    return O.pytest_uniform((B, Nf))


@pytest.mark.parametrize("tag", ["det", "rand"])
def test_sample_pdf_bulk(tag):
    """C2-size capture (4096 x 128) of the reference's own sample_pdf call inside render_rays, weights produced by its D=8/W=256
    coarse pass: the oracle reproduces every index (bit-exact target) and the per-row sample sums."""
    g = golden("sample_pdf_bulk")
    bins, weights = T(g[tag + "_bins"]), T(g[tag + "_weights"])
    samples, inds = O.sample_pdf(bins, weights, _bulk_u(tag))
    safe = np.unpackbits(g[tag + "_safe"])[:inds.numel()].reshape(inds.shape).astype(bool)
    assert int(safe.sum()) == int(g[tag + "_n_safe"])
    mism = inds.numpy() != g[tag + "_inds"].astype(np.int64)
    assert not (mism & safe).any(), "oracle indices differ from the reference away from CDF ties"
    assert mism.sum() == 0, f"{mism.sum()} index mismatches at ties (same ATen build: none expected)"
    close(samples.double().sum(-1).float(), g[tag + "_samples_sum"], rtol=1e-6, atol=1e-5)
    # the coarse depths behind `bins` are a deterministic elementwise function of the rays: the fixture's bins must be the
    # midpoints of the oracle's coarse_z on the same rays
    rays = T(I.ray_batch(4096, seed=5, near=2.125, far=4.67))
    t_rand = O.pytest_uniform((4096, 64)) if tag == "rand" else None
    z = O.coarse_z(rays[:, 6:7], rays[:, 7:8], 64, False, t_rand)
    eq(0.5 * (z[:, 1:] + z[:, :-1]), g[tag + "_bins"], "bins")


RR = [("C1", 4, 128, 64, 0, 1.0, True, 0.0, False, 64), ("C1_noise", 4, 128, 64, 0, 1.0, True, 1.0, False, 32),
      ("C2", 8, 256, 64, 128, 1.0, False, 0.0, False, 64), ("C2_det", 8, 256, 64, 128, 0.0, False, 0.0, False, 16),
      ("small_lindisp", 4, 128, 32, 32, 1.0, False, 0.0, True, 32)]


@pytest.mark.parametrize("tag,D,W,Nc,Nf,perturb,white,noise,lindisp,B", RR)
def test_render_rays(tag, D, W, Nc, Nf, perturb, white, noise, lindisp, B):
    g = golden("render_rays_" + tag)
    och = 5 if Nf > 0 else 4
    sdc = O.as_tensors(I.nerf_state_dict(D, W, 10, 4, och, True, seed=21), True)
    sdf = O.as_tensors(I.nerf_state_dict(D, W, 10, 4, och, True, seed=22), True) if Nf > 0 else None
    net = O.NetCfg(D=D, W=W, output_ch=och)
    cfg = O.RenderCfg(Nc, Nf, perturb, lindisp, white, noise)
    out = O.render_rays_pytest(T(I.ray_batch(B, seed=3)), sdc, sdf, net, cfg)
    keys = ["rgb_map", "disp_map", "acc_map", "depth_map", "raw"] + (
        ["rgb0", "disp0", "acc0", "depth0", "z_std"] if Nf > 0 else [])
    for k in keys:
        close(out[k], g[k], rtol=2e-5, atol=5e-6, name=k)
    far = 6.0
    loss = O.mse(out["rgb_map"], T(g["target"])) + O.mse(out["depth_map"] / far, T(g["prior"]) / far)
    if Nf > 0:
        loss = loss + O.mse(out["rgb0"], T(g["target"])) + O.mse(out["depth0"] / far, T(g["prior"]) / far)
    close(loss, g["loss"], rtol=1e-5, name="loss")
    loss.backward()
    check_grads(g, "gc.", "gc.", sdc)
    if sdf is not None:
        check_grads(g, "gf.", "gf.", sdf)


def test_rays_and_full_render():
    g = golden("render_full_tiny")
    K, c2w = g["K"], T(g["c2w"])
    ro, rd = O.get_rays(16, 16, K, c2w)
    eq(ro, g["rays_o"]); eq(rd, g["rays_d"])
    ron, rdn = O.get_rays_np(16, 16, K, g["c2w"])
    eq(ron, g["rays_o_np"]); eq(rdn, g["rays_d_np"])
    no, nd = O.ndc_rays(16, 16, float(K[0][0]), 1.0, ro, rd)
    eq(no, g["ndc_o"]); eq(nd, g["ndc_d"])
    sdc = O.as_tensors(I.nerf_state_dict(4, 128, 10, 4, 5, True, seed=31))
    sdf = O.as_tensors(I.nerf_state_dict(4, 128, 10, 4, 5, True, seed=32))
    net = O.NetCfg(D=4, W=128, output_ch=5)
    for ndc in (False, True):
        near, far = (0.0, 1.0) if ndc else (2.0, 6.0)
        rays = O.build_ray_batch(ro, rd, near, far, True, ndc, 16, 16, float(K[0][0]))
        with torch.no_grad():
            out = O.render_rays_pytest(rays, sdc, sdf, net, O.RenderCfg(16, 16, 0.0))
        sfx = "_ndc" if ndc else ""
        for k, gk in (("rgb_map", "rgb"), ("disp_map", "disp"), ("acc_map", "acc"), ("depth_map", "depth"),
                      ("rgb0", "rgb0"), ("depth0", "depth0"), ("z_std", "z_std")):
            close(out[k].reshape(g[gk + sfx].shape), g[gk + sfx], rtol=2e-5, atol=5e-6, name=gk + sfx)


def test_warp():
    g = golden("warp")
    K = T(g["K"]); P = T(g["P"]); w2c = T(g["w2c_ref"])
    c2w = torch.eye(4); c2w[:3, :4] = T(g["poses"][1])
    img = T(g["images"][1]).permute(2, 0, 1); dep = T(g["depths"][1])
    for tag, flip, masked in (("V", True, False), ("VT", False, True)):
        rgb, d, Xc, ro, rd, inb = O.get_ref_rays(w2c, c2w, K, P, img, dep, flip, masked)
        eq(rgb, g[tag + ".rgb_ref"][0]); eq(d, g[tag + ".depth_ref"][0, 0])
        eq(Xc, g[tag + ".Xc"][0] if not masked else g[tag + ".Xc"])
        eq(ro, g[tag + ".rays_o"]); eq(rd, g[tag + ".rays_d"]); eq(inb, g[tag + ".mask"][0])
    Xc, x, y, inb = O.warp_points(P, w2c, K, 32, 40, True)
    eq(y, g["label.y"][0]); eq(x, g["label.x"][0]); eq(inb, g["label.mask"][0]); eq(Xc[:, 2], g["label.z"][0])


def test_hard_masks():
    g = golden("hardmask_tiny")
    masks, thr = O.hard_masks(96, 128, g["K"], g["poses"], g["depths"], list(g["i_train"]))
    eq(masks, g["masks"])
    assert np.array_equal(thr, g["thr"], equal_nan=True)
    assert not g["masks"][3].any() and (g["thr"][:, 3] > 0.1).any()


def test_masked_losses():
    g = golden("losses_mask")
    far, c = float(g["far"]), float(g["coef"])
    for tag, m in (("mixed", g["mask"]), ("allone", np.ones_like(g["mask"]))):
        r = T(g["rgb"]).requires_grad_(True); d = T(g["depth"]).requires_grad_(True)
        lr = O.masked_rgb_loss(r, T(g["target"]), T(m), c)
        ld = O.masked_depth_loss(d, T(g["prior"]), T(m), far)
        eq(lr, g[tag + ".l_rgb"]); eq(ld, g[tag + ".l_depth"])
        (lr + ld).backward()
        eq(r.grad, g[tag + ".d_rgb"]); eq(d.grad, g[tag + ".d_depth"])
    eq(O.psnr_from_mse(O.mse(T(g["rgb"]), T(g["target"]))), g["psnr"])


def test_train_10_steps():
    """Optimiser wiring: Adam(0.9,0.999,1e-8) + per-step exponential lr (R:768-788)."""
    g = golden("train_10steps_C1")
    sd = O.as_tensors(I.nerf_state_dict(4, 128, 10, 4, 4, True, seed=41), True)
    net = O.NetCfg(D=4, W=128, output_ch=4)
    cfg = O.RenderCfg(64, 0, 1.0, False, True, 0.0)
    m = {k: torch.zeros_like(v) for k, v in sd.items()}
    v = {k: torch.zeros_like(p) for k, p in sd.items()}
    lr = 5e-4
    for i in range(10):
        rays = T(I.ray_batch(256, seed=100 + i))
        target = T(np.random.RandomState(200 + i).uniform(size=(256, 3)).astype(np.float32))
        out = O.render_rays_pytest(rays, sd, None, net, cfg)
        loss = O.mse(out["rgb_map"], target)
        assert abs(loss.item() - g["losses"][i]) <= 2e-6 * g["losses"][i], (i, loss.item(), g["losses"][i])
        grads = torch.autograd.grad(loss, [p for p in sd.values()], allow_unused=True)
        with torch.no_grad():
            for (k, p), gr in zip(sd.items(), grads):
                if gr is None:
                    continue   # temp_rgb / temp_depth / depth_scale and the unused view branch bits
                O.adam_step(p, gr, m[k], v[k], i + 1, lr)
        lr = O.lr_at(5e-4, i, 250)
    for k, p in sd.items():
        ref = g["final." + k]
        assert np.abs(p.detach().numpy() - ref).max() <= 5e-6, k
    assert float(g["test_perturb"]) == 0.0


def test_train_10_steps_view_variant():
    """The ConsistentNeRF loop's wiring: masked rgb + depth losses on both levels (V:1645-1648, 1737, 1786-1788, 1865),
    clip_grad_value_(0.1) before Adam (V:1983), lr decay — against the reference's own 10 steps (`train_10steps_V`)."""
    g = golden("train_10steps_V")
    sds = [O.as_tensors(I.nerf_state_dict(4, 128, 10, 4, 5, True, seed=s, gain=0.6), True) for s in (51, 52)]
    net, cfg = O.NetCfg(D=4, W=128, output_ch=5), O.RenderCfg(32, 32, 1.0)
    params = [p for sd in sds for p in sd.values()]
    m = [torch.zeros_like(p) for p in params]
    v = [torch.zeros_like(p) for p in params]
    near, far, lr = 2.0, 6.0, 5e-4
    for i in range(10):
        rays = T(I.ray_batch(256, seed=300 + i, near=near, far=far))
        rs = np.random.RandomState(400 + i)
        target = T(rs.uniform(size=(256, 3)).astype(np.float32))
        prior = T(rs.uniform(near, far, size=(256,)).astype(np.float32))
        mask = T((rs.uniform(size=(256,)) < 0.6).astype(np.float32))
        out = O.render_rays_pytest(rays, sds[0], sds[1], net, cfg)
        terms = (O.masked_rgb_loss(out["rgb_map"], target, mask, 0.2), O.masked_depth_loss(out["depth_map"], prior, mask, far),
                 O.masked_rgb_loss(out["rgb0"], target, mask, 0.2), O.masked_depth_loss(out["depth0"], prior, mask, far))
        for t, k in zip(terms, ("img_loss", "depth_loss", "img_loss0", "depth_loss0")):
            assert abs(t.item() - g[k][i]) <= 5e-5 * abs(g[k][i]), (i, k, t.item(), g[k][i])
        loss = terms[0] + terms[1] + terms[2] + terms[3]
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        flat = torch.cat([(gr if gr is not None else torch.zeros_like(p)).reshape(-1) for p, gr in zip(params, grads)])
        assert int((flat.abs() > 0.1).sum()) == int(g["n_clipped"][i]), (i, int((flat.abs() > 0.1).sum()), g["n_clipped"][i])
        if i == 0:
            assert np.array_equal(torch.nonzero(flat.abs() > 0.1).reshape(-1).numpy(), g["clipped_idx"])
        with torch.no_grad():
            for p, gr, mm, vv in zip(params, grads, m, v):
                if gr is not None:
                    O.adam_step(p, gr, mm, vv, i + 1, lr, clip=0.1)
        lr = O.lr_at(5e-4, i, 250)
    assert abs(lr - float(g["lr_final"])) < 1e-12
    idx = torch.from_numpy(g["clipped_idx"])
    flat_p = torch.cat([p.detach().reshape(-1) for p in params])
    flat_m, flat_v = torch.cat([x.reshape(-1) for x in m]), torch.cat([x.reshape(-1) for x in v])
    # what the clip changes is far above the tolerance: the fixture's own no-clip control differs by 1e-4 (weights),
    # 24 % (first moment) and >50 % (second moment) at these positions
    close(flat_p[idx], g["final_at_clipped"], rtol=0, atol=2e-5)
    close(flat_m[idx], g["exp_avg_at_clipped"], rtol=2e-3, atol=1e-7)
    close(flat_v[idx], g["exp_avg_sq_at_clipped"], rtol=2e-3, atol=1e-12)
    assert np.abs(g["final_at_clipped"] - g["final_at_clipped_noclip"]).max() > 2e-4
    for tag, sd in (("c", sds[0]), ("f", sds[1])):
        for k, p in sd.items():
            assert np.abs(p.detach().reshape(-1)[::7].numpy() - g[f"final.{tag}.{k}.sub"]).max() <= 2e-5, (tag, k)


def test_ss_block_golden():
    """VT:905-925 (the in-loop consistency block in front of the second render) against the reference's own statements (`ssloss`):
    masks and gathered targets exact, the exit value of the doubled threshold, the reference rays 1e-6 — for a threshold that
    passes at once (a) and one that needs doublings (b)."""
    g = golden("ssloss")
    Hh, Ww = g["images"].shape[1:3]
    ro, rd = O.get_rays_np(Hh, Ww, g["K"], g["poses"][0][:3, :4])
    for tag, thr in (("a", 0.1), ("b", 1e-4)):
        sel, r = g[tag + ".sel"], int(g[tag + ".ref_index"])
        out = O.ss_block(T(ro.reshape(-1, 3)[sel]), T(rd.reshape(-1, 3)[sel]), T(g["depths"][0].reshape(-1)[sel]), T(g["poses"][r]),
                         T(g["K"]), T(g["images"][r]).permute(2, 0, 1), T(g["depths"][r]), thr)
        assert np.array_equal(out["mask_bound"].numpy(), g[tag + ".mask_bound"])
        assert np.array_equal(out["mask"].numpy(), g[tag + ".mask"])
        assert abs(out["thr_next"] - float(g[tag + ".thr_next"])) <= 1e-12 * float(g[tag + ".thr_next"])
        if tag == "b":
            assert out["thr"] > thr          # this case needs the doubling
        assert np.array_equal(out["rgb_target_ref"].numpy(), g[tag + ".rgb_target_ref"])
        assert np.array_equal(out["rays_depth_ref"].numpy(), g[tag + ".rays_depth_ref"])
        close(out["rays_ref"], g[tag + ".rays_ref"], rtol=0, atol=1e-6)
        mb, mk = g[tag + ".mask_bound"].reshape(-1), g[tag + ".mask"].reshape(-1)
        want = np.zeros(mb.shape[0], np.float32)
        want[mb] = mk
        assert np.array_equal(out["sel"].numpy(), want)


def test_ss_primary_losses():
    """VT:941-969: the coin-flip masked consumers of the in-loop consistency masks."""
    g = golden("ssloss_primary")
    for wd in (True, False):
        for coins in ((1, 1, 1, 1), (0, 0, 0, 0), (1, 0, 0, 1), (0, 1, 1, 0)):
            tag = f"{'d' if wd else 'n'}{''.join(map(str, coins))}."
            leaf = {k: T(g[k]).requires_grad_(True) for k in ("rgb", "rgb0", "depth_pred", "depth0")}
            seq = coins if wd else (coins[0], coins[2])
            loss, il, il0 = O.ss_primary_losses(leaf["rgb"], leaf["depth_pred"], leaf["rgb0"], leaf["depth0"], T(g["target_s"]),
                                                T(g["depth_cas_s"]), T(g["mask_bound"]), T(g["mask"]), wd, seq)
            eq(loss, g[tag + "loss"]); eq(il, g[tag + "img_loss"]); eq(il0, g[tag + "img_loss0"])
            eq(O.psnr_from_mse(il), g[tag + "psnr"]); eq(O.psnr_from_mse(il0), g[tag + "psnr0"])
            loss.backward()
            for k, t in leaf.items():
                eq(t.grad if t.grad is not None else torch.zeros_like(t), g[tag + "d_" + k], tag + k)


def test_ray_bank_and_samplers_golden():
    """the oracle's ray bank / batching / --no_batching sampler against the statements of the
    reference's train() replayed around its own get_rays_np / get_rays (tests/golden/make_golden.py::fx_raybank)."""
    g = golden("raybank")
    Hh, Ww, focal = int(g["hwf"][0]), int(g["hwf"][1]), float(g["hwf"][2])
    K = I.intrinsics(Hh, Ww, focal)
    bank = O.ray_bank(g["images"], g["poses"], Hh, Ww, K, g["i_train"])
    assert np.array_equal(bank, g["unshuffled"])
    perm = O.numpy_shuffle_perm(bank.shape[0], 5)
    assert np.array_equal(bank[perm], g["bank0"])
    bb = O.BankBatches(g["bank0"].copy())
    for it in range(4):
        rays, tgt = bb.next(50, g["rand_idx"])
        assert np.array_equal(rays, g[f"rays{it}"]) and np.array_equal(tgt, g[f"tgt{it}"]), it
    for tag, frac in (("full", None), ("crop", 0.5)):
        assert np.array_equal(O.crop_coords(Hh, Ww, frac), g[f"nb_{tag}_coords"].astype(np.int64))
        rays, tgt = O.sample_image_rays(g["images"][2], g["poses"][2], Hh, Ww, K, g[f"nb_{tag}_inds"], frac)
        assert np.array_equal(rays, g[f"nb_{tag}_rays"]) and np.array_equal(tgt, g[f"nb_{tag}_tgt"])


def test_patch_sampler_and_depth_term_golden():
    """f-2 / f-5: the oracle's patch sampler and monocular-depth patch term against the reference's own statements
    (V:1472-1509, V:1681-1719) executed on seeded inputs."""
    g = golden("patch")
    Hh, Ww = (int(v) for v in g["hw"])
    for tag in ("full", "crop"):
        pre = tuple(int(v) for v in g["crop_dhw"]) if tag == "crop" else None
        np.random.seed(21)
        starts = O.draw_patch_starts(Hh, Ww, 4, 16, pre)
        pc = O.patch_coords(starts, 16)
        assert np.array_equal(pc, g[f"{tag}_patch_idxs"])
        coords = O.crop_coords(Hh, Ww, 0.9 if pre else None)
        inds = np.random.choice(coords.shape[0], size=[37], replace=False)      # the draw that follows at V:1507
        assert np.array_equal(inds, g[f"{tag}_select_inds"])
        assert np.array_equal(np.concatenate([pc, coords[inds]], 0), g[f"{tag}_select_coords"])
    for tag in ("a", "b"):
        dp = torch.from_numpy(g[f"term_{tag}_depth"]).requires_grad_(True)
        loss = O.patch_depth_loss(dp, torch.from_numpy(g[f"term_{tag}_mono"]), 4, 256)
        gr, = torch.autograd.grad(loss, dp)
        assert torch.equal(loss.detach(), torch.from_numpy(g[f"term_{tag}_loss"]))
        assert np.array_equal(gr.numpy(), g[f"term_{tag}_grad"], equal_nan=True)


def test_render_rays_trained_network():
    """The oracle on a TRAINED network (fixture `render_rays_trained`: the reference trained the C2 networks 200 steps and rendered
    1024 held-out rays free-running, make_golden.py::fx_trained): 256 of those rays, both perturb settings.  A trained network
    does not amplify depth differences, so the free-running oracle matches the reference's maps to 1e-5."""
    g = golden("render_rays_trained")
    sdc = {k[2:]: T(g[k]) for k in g if k.startswith("c.")}
    sdf = {k[2:]: T(g[k]) for k in g if k.startswith("f.")}
    rays, far = T(g["rays"])[::4], float(g["near_far"][1])
    net = O.NetCfg(8, 256, output_ch=5)
    for tag, perturb in (("p1.", 1.0), ("p0.", 0.0)):
        with torch.no_grad():
            # (the pytest streams are drawn for the fixture's 1024 rays and subsampled like the rays)
            t_rand = O.pytest_uniform((1024, 64))[::4] if perturb > 0 else None
            u = O.pytest_uniform((1024, 128))[::4] if perturb > 0 else torch.from_numpy(
                np.broadcast_to(np.linspace(0., 1., 128), (256, 128)).astype(np.float32).copy())
            out = O.render_rays(rays, sdc, sdf, net, O.RenderCfg(64, 128, perturb), t_rand, u)
        for k, tol in (("rgb0", 1e-5), ("depth0", 1e-5 * far), ("rgb_map", 1e-5), ("acc_map", 1e-5), ("depth_map", 1e-5 * far),
                       ("z_std", 1e-5 * far)):
            d = float((out[k] - T(g[tag + k])[::4]).abs().max())
            assert d <= tol, (tag, k, d)


def test_alt_losses_golden():
    """img2mse_softLpmask (V:58) and the `--use_noise` level schedule (V:80-100, V:1420) against the reference's own objects
    (fixture `altlosses`): value and gradient for three exponents incl. exact-zero residuals; 60 scheduler steps, two settings."""
    g = golden("altlosses")
    for coef in (2.0, 1.0, 0.5):
        for tag, (xk, yk) in (("rgb", ("x3", "y3")), ("depth", ("x1", "y1"))):
            x = T(g[xk]).requires_grad_(True)
            loss = O.mse_soft_lp(x, T(g[yk]), coef)
            loss.backward()
            eq(loss, g[f"{tag}.c{coef}.loss"])
            eq(x.grad, g[f"{tag}.c{coef}.d_x"], f"{tag} c={coef}")
    for total, base, floor in ((200000, 0.05, 0.05), (50, 0.2, 0.05)):
        want = g[f"sched.{total}"]
        got = np.array([O.noise_level(total, k + 1, base, floor) for k in range(60)])
        assert np.array_equal(got, want), (total, got[:5], want[:5])
