"""GPU parity tests (run on the MI355X box: `pytest -m gpu`).  Every test drives the HIP path through the
C ABI (ctypes -> libcnerf_hip.so) and compares with (a) the committed golden vectors captured from the
reference and/or (b) the CPU oracle on the same seeded inputs.

Stated tolerances (fp32 path, v_mfma_f32_32x32x2_f32 + OCML sin/cos/exp vs ATen/Sleef on the CPU):
  raw network outputs      |d| <= 3e-5 * max(1, max|raw|)
  rgb_map / acc_map        |d| <= 2e-5          depth_map  |d| <= 2e-5 * far
  weights                  |d| <= 2e-5          z (coarse) bit-exact;  z (fine) |d| <= 2e-5 * far
  sample_pdf indices       bit-exact wherever min_k |u - cdf_k| > 1e-5 (SURVEY hard part 3), mismatch rate reported
  warp pixels / masks      bit-exact wherever the projected pixel is > 1e-3 px away from a rounding tie
  gradients                |d| <= 1e-5 * max|g| per tensor against an fp64 replay from the kernel's own ReLU masks, AND
                           against the CPU oracle (pinned on the reference captures) differentiating the kernel's own ReLU
                           branch (its sign bits passed in; pattern differences counted, only within 1e-5 of zero);
                           vs the reference capture itself 2e-1 rel-max (one near-zero ReLU may flip: loose by construction)
  end to end (fine level)  |d rgb| <= 5e-3, PSNR-equivalent >= 50 dB vs the capture (conditioning of the
                           2^9-frequency encoding, see test_render_rays_golden); 2e-5 when the oracle is evaluated at
                           the kernel's own sample depths
"""
import os

import numpy as np
import pytest
import torch

import _inputs as I
from conftest import golden
from oracle import nerf_oracle as O

pytestmark = pytest.mark.gpu

# `CNERF_TRAIN_PRECISION=bf16x3 pytest -m gpu` runs this whole suite with the opt-in bf16x3 training arithmetic as the default of
# every network it covers (run_nerf.training_precision) — at the SAME tolerances.  The few tests that pin the exact-fp32 kernels
# by NAME, or bit-identity between the Python surface and the exact-fp32 single-call C path, only make sense for the default.
BF3 = os.environ.get("CNERF_TRAIN_PRECISION", "fp32") == "bf16x3"
fp32_only = pytest.mark.skipif(BF3, reason="pins the exact-fp32 kernels by name / bit-identity with the fp32 single-call C path")


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    from consistentnerf_amd import ops
    ok, name, cus, lds = ops.device_info(0)
    print(f"device: {name} CUs={cus} LDS/CU={lds}")
    assert ok, f"not a gfx950 device: {name}"
    return torch.device("cuda:0")


def T(a, dev=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dev) if dev is not None else t


def maxdiff(a, b):
    a = np.atleast_1d(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a))
    b = np.atleast_1d(b.detach().cpu().numpy() if isinstance(b, torch.Tensor) else np.asarray(b))
    assert a.shape == b.shape, (a.shape, b.shape)
    both_nan = np.isnan(a) & np.isnan(b)
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    d[both_nan] = 0
    assert not np.isnan(d).any(), "NaN mismatch"
    return float(d.max()) if d.size else 0.0


def check(a, b, atol, name):
    d = maxdiff(a, b)
    print(f"  {name}: max|d|={d:.3e} (tol {atol:.1e})")
    assert d <= atol, f"{name}: max|d|={d:.3e} > {atol:.1e}"


def make_model(D, W, vd, och, seed, dev):
    from consistentnerf_amd.run_nerf_helpers import NeRF
    sd = I.nerf_state_dict(D, W, 10, 4, och, vd, seed)
    m = NeRF(D=D, W=W, input_ch=63, output_ch=och, skips=[4], input_ch_views=27 if vd else 0, use_viewdirs=vd)
    m.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    return m.to(dev), sd


def check_param_grads(model, g, prefix_full, prefix_sum, rtol=5e-2, l2tol=5e-3, bias_tol=None):
    """Gradients vs the reference capture.  NOT a tight bound by construction: one ReLU whose pre-activation
    is within fp32 round-off of 0 gets a different mask on the CPU and on the GPU, which perturbs every
    upstream weight gradient by ~1/M of its scale.  The tight (1e-5) check of the backward kernels is
    test_mlp_backward_exact_from_stash, which replays the backward in fp64 from the kernel's own masks."""
    rows, bad = [], []
    for k, p in model.named_parameters():
        gr = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu()
        if prefix_full + k in g:
            ref = g[prefix_full + k]
            got = gr.numpy()
        else:
            ref = g[prefix_sum + k + ".sub"]
            got = gr.reshape(-1)[::61].numpy()
        scale = max(float(np.abs(ref).max()), 1e-12)
        dmax = float(np.abs(got - ref).max()) / scale
        dl2 = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-12))
        rows.append((k, dmax, dl2))
        # bias_tol: a bias gradient is ONE column sum over every ray-sample of signed values; where it cancels, max|g| is far below
        # the mass the fp32 oracle itself rounds against (its sequential CPU sum is the larger error: K_ref32 of
        # tests/test_gpu_training_parity.py) — the seeded sweeps bound those tensors separately
        lim = (bias_tol, bias_tol) if (bias_tol is not None and p.dim() == 1) else (rtol, l2tol)
        if dmax > lim[0] or dl2 > lim[1]:
            bad.append(k)
    for k, dmax, dl2 in rows:
        print(f"    grad {k:28s} rel-max {dmax:.2e}  rel-L2 {dl2:.2e}")
    assert not bad, f"gradient mismatch in {bad}"


def stash_blocks(stash, M, D, W, vd, in_chp=64):
    """training stash of cnerf_mlp_fwd (logically [Mp][rows]) -> dict of [M, cols] float64 CPU tensors (common.hpp).
    Padding points [M, Mp) must be zero rows (the wgrad DMA relies on it).  The last block holds the ReLU sign-bit
    words the backward masks with; they are decoded and checked against the stored activations (bit == h > 0)."""
    Mp = (M + 31) // 32 * 32
    rows = stash.numel() // Mp
    # tile-major storage (common.hpp): tiles of 32 points x 8 columns -> logical [Mp][rows]
    full = stash.reshape(Mp // 32, rows // 8, 32, 8).permute(0, 2, 1, 3).reshape(Mp, rows)
    assert float(full[M:].abs().max()) == 0.0 if Mp > M else True
    s = full[:M].double().cpu()
    out, r = {}, 0

    def take(name, n):
        nonlocal r
        out[name] = s[:, r:r + n]
        r += n
    take("enc", in_chp)
    for l in range(D):
        take(f"h{l}", W)
    if vd:
        take("feat", W); take("denc", 32); take("hv", W // 2)
    nt = W // 32
    md, mdv = (nt + 1) // 2, (nt // 2 + 1) // 2
    nmask = (D * 2 * md + (2 * mdv if vd else 0) + 7) // 8 * 8
    assert r + nmask == s.shape[1], (r, nmask, s.shape)
    words = full[:M, r:].contiguous().view(torch.int32).cpu().numpy().astype(np.int64) & 0xffffffff

    def decode(col0, tiles, m_d):
        """-> [M, 32*tiles] bool in feature order n = 32t + 8(r>>2) + 4hh + (r&3); word layout of relu_bits
        (mlp_common.hpp): MSB first, even registers of the dword's tiles, then the odd ones."""
        bits = np.zeros((M, 32 * tiles), dtype=bool)
        for hh in range(2):
            for d in range(m_d):
                wd = words[:, col0 + hh * m_d + d]
                pos = 31
                for par in range(2):
                    for t in range(2 * d, min(2 * d + 2, tiles)):
                        for rr in range(par, 16, 2):
                            bits[:, 32 * t + 8 * (rr >> 2) + 4 * hh + (rr & 3)] = (wd >> pos) & 1
                            pos -= 1
        return bits
    for l in range(D):
        assert np.array_equal(decode(l * 2 * md, nt, md), out[f"h{l}"].numpy() > 0), f"sign bits of layer {l}"
    if vd:
        assert np.array_equal(decode(D * 2 * md, max(nt // 2, 1), mdv), out["hv"].numpy() > 0), "sign bits (view branch)"
    return out


def relu_masks(stash, M, D, W, vd):
    """The ReLU patterns of a training forward, from its stash: [M, W] bool per trunk layer (+ [M, W/2] for the view branch) on
    the CPU — what O.mlp_forward(..., masks=) applies instead of its own (z > 0).  stash_blocks() checks the sign-bit words the
    dgrad kernel reads against (h > 0) of the stored activations, so these ARE the kernel's derivative patterns."""
    blk = stash_blocks(stash, M, D, W, vd)
    return [blk[f"h{l}"] > 0 for l in range(D)] + ([blk["hv"] > 0] if vd else [])


def check_flips(flips, tol=1e-5):
    """Units whose ReLU pattern differs between the kernel and the oracle must sit within round-off of zero."""
    n, zmax = sum(f[0] for f in flips), max([f[1] for f in flips] + [0.0])
    print(f"  ReLU pattern differences kernel vs oracle: {n} units, max |z| there {zmax:.2e} (bound {tol:.0e})")
    assert zmax < tol, f"a ReLU unit with |z| = {zmax:.3e} took different branches"


# ------------------------------------------------------------------------------------------------
def test_embed(dev):
    from consistentnerf_amd import ops
    g = golden("embed")
    check(ops.embed(T(g["x"], dev), 10), g["L10"], 2e-6, "embed L10")
    check(ops.embed(T(g["x"], dev), 4), g["L4"], 2e-6, "embed L4")


@pytest.mark.parametrize("lindisp", [False, True])
@pytest.mark.parametrize("perturb", [False, True])
def test_coarse_z_bit_exact(dev, lindisp, perturb):
    from consistentnerf_amd import ops
    rays = T(I.ray_batch(300, seed=9), dev)
    tr = O.pytest_uniform((300, 64)) if perturb else None
    z = ops.coarse_z(rays, 64, tr.to(dev) if perturb else None, lindisp)
    ref = O.coarse_z(rays[:, 6:7].cpu(), rays[:, 7:8].cpu(), 64, lindisp, tr)
    assert torch.equal(z.cpu(), ref), f"max diff {maxdiff(z, ref)}"


R2O = [("S64", 64, False, 0.0), ("S192", 192, False, 0.0), ("S64_white", 64, True, 0.0),
       ("S192_white_noise", 192, True, 1.0)]


@pytest.mark.parametrize("tag,S,white,noise", R2O)
def test_composite_golden(dev, tag, S, white, noise):
    from consistentnerf_amd.run_nerf import raw2outputs
    g = golden("raw2outputs_" + tag)
    raw, z, d = I.raw2outputs_inputs(32, S, seed=S + int(white))
    rawt = T(raw, dev).requires_grad_(True)
    rgb, disp, acc, w, depth = raw2outputs(rawt, T(z, dev), T(d, dev), noise, white, pytest=True)
    check(rgb, g["rgb_map"], 2e-5, "rgb_map")
    check(acc, g["acc_map"], 2e-5, "acc_map")
    check(depth, g["depth_map"], 2e-5 * 6, "depth_map")
    check(w, g["weights"], 2e-5, "weights")
    dg, dr = disp.detach().cpu().numpy(), g["disp_map"]
    assert np.array_equal(np.isnan(dg), np.isnan(dr)), "disp NaN pattern (acc==0 rays)"
    ok = ~np.isnan(dr)
    assert np.abs(dg[ok] - dr[ok]).max() <= 2e-5 * max(1.0, np.abs(dr[ok]).max())
    loss = (rgb * T(g["g_rgb"], dev)).sum() + (depth * T(g["g_depth"], dev)).sum() + (acc * T(g["g_acc"], dev)).sum()
    (d_raw,) = torch.autograd.grad(loss, rawt, retain_graph=True)
    check(d_raw, g["d_raw"], 2e-4 * max(1.0, np.abs(g["d_raw"]).max()), "d_raw")
    (dd,) = torch.autograd.grad((disp[2:] * T(g["g_disp"][2:], dev)).sum(), rawt)
    check(dd, g["d_raw_disp"], 2e-4 * max(1.0, np.abs(g["d_raw_disp"]).max()), "d_raw (disp)")


@pytest.mark.parametrize("tag", ["det", "rand"])
def test_sample_pdf_indices(dev, tag):
    from consistentnerf_amd import ops
    g = golden("sample_pdf_" + tag)
    bins, weights = I.sample_pdf_inputs(256, 64, seed=7)
    samples, inds = ops.sample_pdf(T(bins, dev), T(weights, dev), T(g["u"], dev), want_inds=True)
    inds = inds.cpu().numpy()
    safe = g["margin"] > 1e-5
    mism = inds != g["inds"]
    print(f"  index mismatches: {mism.sum()} of {mism.size} ({(mism & safe).sum()} with margin > 1e-5)")
    assert not (mism & safe).any(), "sample_pdf indices must be bit-exact away from CDF ties"
    # rounds 1-2 summed the pdf normaliser in fp64 (correctly rounded) and missed the reference's index at CDF ties (det: 32 of
    # 32768, every one at the u = 1.0 sample of a row); since round 3 the kernel associates that sum exactly as torch.sum does
    # on the CPU (sampling.hip::aten_row_sum) and the indices are bit-exact INCLUDING the ties
    assert mism.sum() == 0, mism.sum()
    # at a CDF tie (u == cdf_k to within round-off, e.g. u = 1.0 in det mode) the reference itself jumps by a whole
    # bin when the neighbouring CDF gap is < 1e-5 (H:246-247), so only non-tie entries are compared tightly
    sg = samples.cpu().numpy()
    d = np.abs(sg - g["samples"])
    print(f"  samples: max|d| safe={d[safe].max():.3e}  at ties={d[~safe].max() if (~safe).any() else 0:.3e}")
    assert d[safe].max() <= 2e-5 * 6
    assert np.all(sg >= bins.min(-1, keepdims=True) - 1e-6) and np.all(sg <= bins.max(-1, keepdims=True) + 1e-6)


@pytest.mark.parametrize("tag", ["det", "rand"])
def test_sample_pdf_bulk_indices(dev, tag):
    """Bulk index parity at the C2 batch size (SURVEY hard part 3; H:207-250): 4096 rows x 128 samples captured from the
    reference's own sample_pdf call inside render_rays, the weights produced by its D=8/W=256 coarse pass (render-like, not
    synthetic), both streams.  Asserted: ZERO index mismatches — INCLUDING the CDF ties.  The det stream (test-time renders:
    u = linspace incl. 1.0) ties on EVERY row at its last sample: there cdf[-1] is 1.0 up to one fp32 ulp of the pdf
    normaliser `torch.sum(weights)`, and the reference's index is 63 (3082 rows) or 62 (1014 rows) by that ulp.  With a
    correctly rounded (fp64) normaliser the kernel missed 1003 of those 4096 last-sample indices and 1 + 1 other ties
    (measured on the MI355X, profiles/r03_sample_pdf_bulk.json); associating the sum as ATen's CPU kernel does
    (sampling.hip::aten_row_sum) removes all of them.  The count is written to gpurun_out/ for profiles/."""
    import json
    from consistentnerf_amd import ops
    g = golden("sample_pdf_bulk")
    B, Nf = 4096, 128
    bins, weights = T(g[tag + "_bins"], dev), T(g[tag + "_weights"], dev)
    if tag == "det":
        u = T(np.broadcast_to(np.linspace(0., 1., Nf), (B, Nf)).astype(np.float32).copy(), dev)
    else:
        u = O.pytest_uniform((B, Nf)).to(dev)
    samples, inds = ops.sample_pdf(bins, weights, u, want_inds=True)
    inds = inds.cpu().numpy()
    ref = g[tag + "_inds"].astype(np.int64)
    safe = np.unpackbits(g[tag + "_safe"])[:ref.size].reshape(ref.shape).astype(bool)
    mism = inds != ref
    rows = int(mism.any(1).sum())
    last = int(mism[:, -1].sum())
    print(f"  bulk {tag}: {int(mism.sum())} index mismatches of {mism.size} ({int((mism & safe).sum())} with margin > 1e-5); "
          f"{rows} rows affected; {last} at the last sample (u = {float(u[0, -1]):.3f})")
    assert not (mism & safe).any(), "sample_pdf indices must be bit-exact away from CDF ties"
    assert mism.sum() == 0, "sample_pdf indices must be bit-exact at CDF ties too (ATen-ordered pdf normaliser)"
    # the fused resampler of the render path (cnerf_resample: z -> midpoints -> sample_pdf -> sort): the coarse depths behind
    # the fixture's bins are regenerated from the rays (cnerf_coarse_z is bit-exact, test_coarse_z_bit_exact)
    rays = T(I.ray_batch(B, seed=5, near=2.125, far=4.67), dev)
    z = ops.coarse_z(rays, 64, O.pytest_uniform((B, 64)).to(dev) if tag == "rand" else None, False)
    assert torch.equal(0.5 * (z[:, 1:] + z[:, :-1]), bins)
    pad = torch.zeros(B, 1, device=dev)
    z_f, z_std, s2, i2 = ops.resample(z, torch.cat([pad, weights, pad], 1), u, want_samples=True)
    assert np.array_equal(i2.cpu().numpy(), ref) and torch.equal(s2, samples)
    # values: per-row sums against the reference (a tie moves one sample by at most one bin width)
    ssum = samples.double().sum(-1).float().cpu().numpy()
    d = np.abs(ssum - g[tag + "_samples_sum"])
    clean = ~mism.any(1)
    assert d[clean].max() <= 128 * 2e-5 * 4.67
    os.makedirs("gpurun_out", exist_ok=True)
    with open(os.path.join("gpurun_out", f"sample_pdf_bulk_{tag}.json"), "w") as f:
        json.dump({"stream": tag, "rows": B, "samples_per_row": Nf, "index_mismatches": int(mism.sum()),
                   "mismatches_with_margin_gt_1e-5": int((mism & safe).sum()), "rows_affected": rows,
                   "mismatches_at_last_sample": last, "samples_within_1e-5_of_a_cdf_entry": int((~safe).sum()),
                   "rate": float(mism.mean()), "max_row_sum_diff_clean_rows": float(d[clean].max())}, f)


def test_resample_vs_oracle(dev):
    from consistentnerf_amd import ops
    rs = np.random.RandomState(3)
    B, Nc, Nf = 257, 64, 128
    z = np.sort(rs.uniform(2, 6, size=(B, Nc)), -1).astype(np.float32)
    w = (rs.uniform(size=(B, Nc)) ** 6).astype(np.float32)
    u = O.pytest_uniform((B, Nf))
    zf, zstd, samples, inds = ops.resample(T(z, dev), T(w, dev), u.to(dev), want_samples=True)
    zt, wt = T(z), T(w)
    ref_s, ref_i = O.sample_pdf(0.5 * (zt[:, 1:] + zt[:, :-1]), wt[:, 1:-1], u)
    ref_z, _ = torch.sort(torch.cat([zt, ref_s], -1), -1)
    check(samples, ref_s, 2e-5 * 6, "z_samples")
    check(zf, ref_z, 2e-5 * 6, "z_fine (sorted)")
    assert (zf[:, 1:] >= zf[:, :-1]).all(), "sortedness"
    check(zstd, torch.std(ref_s, dim=-1, unbiased=False), 2e-5, "z_std")
    mism = (inds.cpu() != ref_i).float().mean().item()
    print(f"  index mismatch rate vs oracle: {mism:.2e}")
    assert mism < 1e-3


MLP_CASES = [("D8W256_vd", 8, 256, True, 5), ("D4W128_vd", 4, 128, True, 4), ("D4W128_novd", 4, 128, False, 5),
             ("D8W128_vd", 8, 128, True, 5)]


@pytest.mark.parametrize("tag,D,W,vd,och", MLP_CASES)
def test_mlp_golden(dev, tag, D, W, vd, och):
    """a4+a5+a6 forward and backward against the reference capture (explicit-points mode)."""
    from consistentnerf_amd.run_nerf import run_network
    from consistentnerf_amd.run_nerf_helpers import get_embedder
    g = golden("mlp_" + tag)
    model, _ = make_model(D, W, vd, och, 11, dev)
    e, _ = get_embedder(10, 0)
    ed = get_embedder(4, 0)[0] if vd else None
    raw = run_network(T(g["pts"], dev), T(g["dirs"], dev) if vd else None, model, e, ed)
    scale = max(1.0, float(np.abs(g["raw"]).max()))
    check(raw, g["raw"], 3e-5 * scale, "raw")
    M = raw.shape[0] * raw.shape[1]
    masks = relu_masks(raw.grad_fn.stash, M, D, W, vd)
    (raw * T(g["G"], dev)).sum().backward()
    # (a) vs the reference capture: loose on purpose (a near-zero ReLU may take a different branch than on the CPU, see
    # check_param_grads)
    check_param_grads(model, g, "grad.", "gs.", rtol=2e-1, l2tol=1e-1)
    # (b) TIGHT, with the ReLU patterns held equal: the oracle — pinned on this very capture by
    # tests/test_oracle_golden.py — differentiates the branch the kernel took (its sign bits are passed in), and every unit
    # where the two patterns differ must lie within round-off of zero
    sd = O.as_tensors(I.nerf_state_dict(D, W, 10, 4, och, vd, 11), True)
    cfg = O.NetCfg(D, W, use_viewdirs=vd, output_ch=och)
    pts, flips = T(g["pts"]), []
    ref = O.query(sd, pts, T(g["dirs"]) if vd else None, cfg, masks, flips)
    check_flips(flips)
    (ref * T(g["G"])).sum().backward()
    tight = {"t." + k: (p.grad if p.grad is not None else torch.zeros_like(p)).numpy() for k, p in sd.items()}
    print("  parameter gradients vs the oracle on the kernel's ReLU branch:")
    check_param_grads(model, tight, "t.", "t.", rtol=1e-5, l2tol=1e-5)    # measured 3e-6


# (the envelope's far corners too: their capture-style gradient check is loose by construction, this one is not)
EXACT_CASES = MLP_CASES + [("D16W128_novd_och8", 16, 128, False, 8), ("D1W256_vd", 1, 256, True, 4),
                           ("D9W256_novd_och4", 9, 256, False, 4), ("D2W64_vd", 2, 64, True, 4)]


@pytest.mark.parametrize("tag,D,W,vd,och", EXACT_CASES)
@pytest.mark.parametrize("M_rays,S", [(24, 16), (37, 13)])
def test_mlp_backward_exact_from_stash(dev, tag, D, W, vd, och, M_rays, S):
    """Tight check of dgrad + wgrad + split reduction + stash layout: the backward is replayed in fp64 from the
    activations the forward kernel itself stashed (so ReLU masks are identical by construction) and must agree
    to 1e-5 of each tensor's max.  Also checks the stash against the oracle's activations.  M = 481 is not a
    multiple of 32 (ragged last wave, masked padding columns in wgrad)."""
    from consistentnerf_amd import ops
    from consistentnerf_amd.run_nerf import _packed
    model, sd = make_model(D, W, vd, och, 11, dev)
    rs = np.random.RandomState(5)
    pts = rs.uniform(-3, 3, size=(M_rays, S, 3)).astype(np.float32)
    dirs = rs.normal(size=(M_rays, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    M = M_rays * S
    spec = model.spec()
    packed = _packed(model)
    raw, stash = ops.mlp_forward(spec, packed, M_rays, S, pts=T(pts.reshape(-1, 3), dev),
                                 dirs=T(dirs, dev) if vd else None, want_stash=True)
    G = rs.normal(size=(M, spec.raw_ch)).astype(np.float32)
    grads = ops.mlp_backward(spec, packed, T(G, dev), M_rays, S, stash)
    names = [n for n, _ in I.nerf_param_shapes(D, W, 63, 27 if vd else 0, och, vd)][3:]
    got = {n: g_.double().cpu() for n, g_ in zip(names, grads)}
    blk = stash_blocks(stash, M, D, W, vd)
    # stash == oracle activations (fp32 tolerance)
    x = O.embed(T(pts.reshape(-1, 3)), 10)
    check(blk["enc"][:, :63].float(), x, 2e-6, "stash gamma(x)")
    assert float(blk["enc"][:, 63].abs().max()) == 0.0
    Wd = {k: T(v).double() for k, v in sd.items()}
    Gd = T(G).double()
    # trunk activations vs an fp64 forward: values to fp32 round-off; ReLU masks may only disagree where the fp64
    # pre-activation is itself within round-off of zero (this is what makes capture-vs-kernel gradients jumpy)
    h64, flips = x.double(), 0
    for l in range(D):
        inp = torch.cat([x.double(), h64], 1) if (D > 5 and l == 5) else h64
        z64 = inp @ Wd[f"pts_linears.{l}.weight"].t() + Wd[f"pts_linears.{l}.bias"]
        hk = blk[f"h{l}"]
        assert float((hk - z64.clamp(min=0)).abs().max()) <= 2e-5 * max(1.0, float(z64.abs().max())), f"h{l}"
        dis = (hk > 0) != (z64 > 0)
        flips += int(dis.sum())
        assert not dis.any() or float(z64[dis].abs().max()) < 1e-5, f"layer {l}: ReLU mask differs away from zero"
        h64 = z64.clamp(min=0)
    print(f"  ReLU masks differing from the fp64 forward (all at |z| < 1e-5): {flips}")
    ref = {}
    if vd:
        d_rgb, d_sig = Gd[:, :3], Gd[:, 3:4]
        dZv = (d_rgb @ Wd["rgb_linear.weight"]) * (blk["hv"] > 0)
        ref["rgb_linear.weight"], ref["rgb_linear.bias"] = d_rgb.t() @ blk["hv"], d_rgb.sum(0)
        vin = torch.cat([blk["feat"], blk["denc"][:, :27]], 1)
        ref["views_linears.0.weight"], ref["views_linears.0.bias"] = dZv.t() @ vin, dZv.sum(0)
        dF = dZv @ Wd["views_linears.0.weight"][:, :W]
        hl = blk[f"h{D-1}"]
        ref["feature_linear.weight"], ref["feature_linear.bias"] = dF.t() @ hl, dF.sum(0)
        ref["alpha_linear.weight"], ref["alpha_linear.bias"] = d_sig.t() @ hl, d_sig.sum(0)
        dH = dF @ Wd["feature_linear.weight"] + d_sig @ Wd["alpha_linear.weight"]
    else:
        hl = blk[f"h{D-1}"]
        ref["output_linear.weight"], ref["output_linear.bias"] = Gd.t() @ hl, Gd.sum(0)
        dH = Gd @ Wd["output_linear.weight"]
    dZ = dH * (blk[f"h{D-1}"] > 0)
    for l in range(D - 1, 0, -1):
        skip_layer = (D > 5 and l == 5)
        inp = torch.cat([blk["enc"][:, :63], blk[f"h{l-1}"]], 1) if skip_layer else blk[f"h{l-1}"]
        ref[f"pts_linears.{l}.weight"], ref[f"pts_linears.{l}.bias"] = dZ.t() @ inp, dZ.sum(0)
        Wl = Wd[f"pts_linears.{l}.weight"]
        dZ = (dZ @ (Wl[:, 63:] if skip_layer else Wl)) * (blk[f"h{l-1}"] > 0)
    ref["pts_linears.0.weight"], ref["pts_linears.0.bias"] = dZ.t() @ blk["enc"][:, :63], dZ.sum(0)
    worst = 0.0
    for n in names:
        if n not in ref:                       # views_linears of a no-viewdirs net: no gradient
            assert float(got[n].abs().max()) == 0.0, n
            continue
        r = ref[n].reshape(got[n].shape)
        d = float((got[n] - r).abs().max() / max(float(r.abs().max()), 1e-30))
        worst = max(worst, d)
        assert d <= 1e-5, f"{n}: rel max diff {d:.3e}"
    print(f"  exact backward: worst rel max diff {worst:.3e}")


RR = [("C1", 4, 128, 64, 0, 1.0, True, 0.0, False, 64), ("C1_noise", 4, 128, 64, 0, 1.0, True, 1.0, False, 32),
      ("C2", 8, 256, 64, 128, 1.0, False, 0.0, False, 64), ("C2_det", 8, 256, 64, 128, 0.0, False, 0.0, False, 16),
      ("small_lindisp", 4, 128, 32, 32, 1.0, False, 0.0, True, 32)]


def _kwargs(coarse, fine, Nc, Nf, perturb, white, noise, lindisp):
    from consistentnerf_amd.run_nerf import run_network
    from consistentnerf_amd.run_nerf_helpers import get_embedder
    e, _ = get_embedder(10, 0)
    ed, _ = get_embedder(4, 0)
    q = lambda inputs, viewdirs, fn: run_network(inputs, viewdirs, fn, embed_fn=e, embeddirs_fn=ed)  # noqa: E731
    return dict(network_query_fn=q, perturb=perturb, N_importance=Nf, network_fine=fine, N_samples=Nc,
                network_fn=coarse, white_bkgd=white, raw_noise_std=noise, lindisp=lindisp)


@pytest.mark.parametrize("tag,D,W,Nc,Nf,perturb,white,noise,lindisp,B", RR)
def test_render_rays_golden(dev, tag, D, W, Nc, Nf, perturb, white, noise, lindisp, B):
    """a3 end to end (incl. a7-a9) against the reference capture, forward and parameter gradients."""
    from consistentnerf_amd import run_nerf_view as V
    g = golden("render_rays_" + tag)
    och = 5 if Nf > 0 else 4
    coarse, _ = make_model(D, W, True, och, 21, dev)
    fine = make_model(D, W, True, och, 22, dev)[0] if Nf > 0 else None
    rays = T(I.ray_batch(B, seed=3), dev)
    ret = V.render_rays(rays, retraw=True, pytest=True, _debug=True,
                        **_kwargs(coarse, fine, Nc, Nf, perturb, white, noise, lindisp))
    dbg = {k: ret.pop(k) for k in ("_z_coarse", "_z_vals", "_weights")}
    ret.pop("_raw_coarse", None)
    assert set(ret) == {k for k in g if not k.startswith(("gc.", "gf.")) and k not in ("target", "prior", "loss")}
    far = 6.0
    # (1) coarse level: identical sample depths on both sides (coarse z is bit-exact) -> tight bounds
    lvl0 = "0" if Nf > 0 else "_map"
    check(ret["rgb" + lvl0], g["rgb" + lvl0], 2e-5, "rgb (coarse level)")
    check(ret["acc" + lvl0], g["acc" + lvl0], 2e-5, "acc (coarse level)")
    check(ret["depth" + lvl0], g["depth" + lvl0], 2e-5 * far, "depth (coarse level)")
    if Nf > 0:
        # (2) fine level, teacher-forced: the CPU oracle evaluated at the GPU's own fine depths -> tight bounds.
        sdc = O.as_tensors(I.nerf_state_dict(D, W, 10, 4, och, True, seed=21), True)
        sdf = O.as_tensors(I.nerf_state_dict(D, W, 10, 4, och, True, seed=22), True)
        node_f = ret["raw"].grad_fn
        node_c = node_f.pair.coarse()
        flips = []
        ref = O.render_rays_pytest(rays.cpu(), sdc, sdf, O.NetCfg(D, W, output_ch=och),
                                   O.RenderCfg(Nc, Nf, perturb, lindisp, white, noise),
                                   z_fine=dbg["_z_vals"].cpu(), flips=flips,
                                   masks_coarse=relu_masks(node_c.stash, B * Nc, D, W, True),
                                   masks_fine=relu_masks(node_f.stash, B * (Nc + Nf), D, W, True))
        check_flips(flips)
        tf_target, tf_prior = T(g["target"]), T(g["prior"])
        tf_loss = (O.mse(ref["rgb_map"], tf_target) + O.mse(ref["depth_map"] / far, tf_prior / far) +
                   O.mse(ref["rgb0"], tf_target) + O.mse(ref["depth0"] / far, tf_prior / far))
        tf_loss.backward()
        tf_grads = {"gc." + k: (p.grad if p.grad is not None else torch.zeros_like(p)).numpy() for k, p in sdc.items()}
        tf_grads.update({"gf." + k: (p.grad if p.grad is not None else torch.zeros_like(p)).numpy()
                         for k, p in sdf.items()})
        check(ret["rgb_map"], ref["rgb_map"], 2e-5, "rgb_map (oracle at the kernel's depths)")
        check(ret["acc_map"], ref["acc_map"], 2e-5, "acc_map (oracle at the kernel's depths)")
        check(ret["depth_map"], ref["depth_map"], 2e-5 * far, "depth_map (oracle at the kernel's depths)")
        check(ret["raw"], ref["raw"], 3e-5 * max(1.0, float(ref["raw"].abs().max())), "raw (oracle at the kernel's depths)")
        # (3) the fine depths themselves: resampling from 1e-6-different coarse weights moves samples by
        # ~1e-6 relative (and by a whole flat bin at exact CDF ties, see test_sample_pdf_indices)
        zk, zr = dbg["_z_vals"].cpu(), None
        with torch.no_grad():
            zr = O.render_rays_pytest(rays.cpu(), sdc, sdf, O.NetCfg(D, W, output_ch=och),
                                      O.RenderCfg(Nc, Nf, perturb, lindisp, white, noise))["z_vals"]
        dz = (zk - zr).abs()
        print(f"  fine depths vs oracle: median {dz.median():.2e}  99.9% {dz.flatten().kthvalue(int(dz.numel()*0.999)).values:.2e}  max {dz.max():.2e}")
        assert dz.median() <= 2e-6 * far
        check(ret["z_std"], g["z_std"], 1e-3 * far, "z_std")
    # (4) end to end vs the reference capture.  The 2^9-frequency encoding and the He-gain random nets of the
    # fixtures amplify the ~1e-6 depth differences of (3) by ~1e3, so this bound is about conditioning, not
    # kernel error (which (1),(2) pin): |d rgb| <= 5e-3 and PSNR-equivalent >= 50 dB.
    check(ret["rgb_map"], g["rgb_map"], 5e-3, "rgb_map vs capture")
    check(ret["acc_map"], g["acc_map"], 5e-3, "acc_map vs capture")
    check(ret["depth_map"], g["depth_map"], 5e-3 * far, "depth_map vs capture")
    mse_rgb = float(((ret["rgb_map"].detach().cpu() - T(g["rgb_map"])) ** 2).mean())
    print(f"  PSNR-equivalent of the rgb difference: {-10*np.log10(max(mse_rgb, 1e-20)):.1f} dB")
    assert mse_rgb <= 1e-5
    target, prior = T(g["target"], dev), T(g["prior"], dev)
    loss = V.img2mse(ret["rgb_map"], target) + V.img2mse(ret["depth_map"] / far, prior / far)
    if Nf > 0:
        loss = loss + V.img2mse(ret["rgb0"], target) + V.img2mse(ret["depth0"] / far, prior / far)
    check(loss, g["loss"], 1e-4, "loss")
    loss.backward()
    if Nf > 0:
        # gradients against the oracle differentiated AT THE KERNEL'S DEPTHS (same sample set on both sides)
        # TIGHT: same sample set AND same ReLU branch on both sides
        print("  parameter gradients vs oracle at the kernel's depths, on the kernel's ReLU branch:")
        check_param_grads(coarse, tf_grads, "gc.", "gc.", rtol=1e-5, l2tol=1e-5)       # measured <= 3e-6
        check_param_grads(fine, tf_grads, "gf.", "gf.", rtol=1e-5, l2tol=1e-5)
        print("  parameter gradients vs the reference capture (different fine sample set at CDF ties):")
        check_param_grads(coarse, g, "gc.", "gc.", rtol=2e-1, l2tol=1e-1)
        check_param_grads(fine, g, "gf.", "gf.", rtol=2e-1, l2tol=1e-1)
    else:
        check_param_grads(coarse, g, "gc.", "gc.", rtol=2e-3, l2tol=1e-3)


def _render_sweep_cases(n=10, seed=20260930):
    """Seeded draws of render_rays configurations: (D, W, Nc, Nf, perturb, white_bkgd, raw_noise_std, lindisp, B, weight seed)."""
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        out.append((int(rs.choice([2, 4, 8])), int(rs.choice([64, 128, 256])), int(rs.choice([4, 16, 33, 64])),
                    int(rs.choice([1, 8, 32, 100, 128])), float(rs.choice([0.0, 1.0])), bool(rs.randint(0, 2)),
                    float(rs.choice([0.0, 1.0])), bool(rs.randint(0, 2)), int(rs.choice([1, 5, 32, 45, 96])), 300 + 2 * i))
    return out


@pytest.mark.parametrize("D,W,Nc,Nf,perturb,white,noise,lindisp,B,wseed", _render_sweep_cases())
def test_random_render_rays_vs_oracle(dev, D, W, Nc, Nf, perturb, white, noise, lindisp, B, wseed):
    """Seeded sweep of a3 (coarse pass, compositing, hierarchical resampling, fine pass) over sample counts that are not multiples
    of anything, single rays, both backgrounds, density noise and inverse-depth spacing: the coarse level against the oracle at
    2e-5 (its depths are bit-exact), the fine level and the parameter gradients against the oracle evaluated at the kernel's own
    fine depths on the kernel's ReLU branch (2e-5 / 1e-5 max per tensor)."""
    from consistentnerf_amd import run_nerf_view as V
    far = 6.0
    coarse, sdc_np = make_model(D, W, True, 5, wseed, dev)
    fine, sdf_np = make_model(D, W, True, 5, wseed + 1, dev)
    rays = T(I.ray_batch(B, seed=wseed), dev)
    ret = V.render_rays(rays, retraw=True, pytest=True, _debug=True, **_kwargs(coarse, fine, Nc, Nf, perturb, white, noise, lindisp))
    node_f = ret["raw"].grad_fn
    node_c = node_f.pair.coarse()
    sdc, sdf = O.as_tensors(sdc_np, True), O.as_tensors(sdf_np, True)
    flips = []
    ref = O.render_rays_pytest(rays.cpu(), sdc, sdf, O.NetCfg(D, W, output_ch=5), O.RenderCfg(Nc, Nf, perturb, lindisp, white, noise),
                               z_fine=ret["_z_vals"].cpu(), flips=flips,
                               masks_coarse=relu_masks(node_c.stash, B * Nc, D, W, True),
                               masks_fine=relu_masks(node_f.stash, B * (Nc + Nf), D, W, True))
    check_flips(flips)
    for k, tol in (("rgb0", 2e-5), ("acc0", 2e-5), ("depth0", 2e-5 * far), ("rgb_map", 2e-5), ("acc_map", 2e-5),
                   ("depth_map", 2e-5 * far)):
        check(ret[k], ref[k].detach(), tol, k)
    check(ret["raw"], ref["raw"].detach(), 3e-5 * max(1.0, float(ref["raw"].detach().abs().max())), "raw")
    rs = np.random.RandomState(wseed)
    tgt = T(rs.uniform(size=(B, 3)).astype(np.float32))
    loss = V.img2mse(ret["rgb_map"], tgt.to(dev)) + V.img2mse(ret["rgb0"], tgt.to(dev)) + V.img2mse(ret["depth_map"] / far, tgt[:, 0].to(dev))
    loss.backward()
    rl = O.mse(ref["rgb_map"], tgt) + O.mse(ref["rgb0"], tgt) + O.mse(ref["depth_map"] / far, tgt[:, 0])
    assert abs(float(loss.detach()) - float(rl.detach())) <= 2e-6 * max(1e-3, abs(float(rl.detach())))
    rl.backward()
    tg = {"gc." + k: (p_.grad if p_.grad is not None else torch.zeros_like(p_)).numpy() for k, p_ in sdc.items()}
    tg.update({"gf." + k: (p_.grad if p_.grad is not None else torch.zeros_like(p_)).numpy() for k, p_ in sdf.items()})
    check_param_grads(coarse, tg, "gc.", "gc.", rtol=1e-5, l2tol=1e-5, bias_tol=1e-4)
    check_param_grads(fine, tg, "gf.", "gf.", rtol=1e-5, l2tol=1e-5, bias_tol=1e-4)


def _trained_models(g, dev=None):
    from consistentnerf_amd.run_nerf_helpers import NeRF
    out = []
    for tag in ("c.", "f."):
        m = NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
        m.load_state_dict({k[len(tag):]: T(g[k]) for k in g if k.startswith(tag)}, strict=True)
        out.append(m.to(dev) if dev is not None else m)
    return out


def test_render_rays_trained_network_free_running(dev):
    """a3 END TO END, FREE-RUNNING, on a well-conditioned network.  Fixture `render_rays_trained` = the
    reference itself trained the C2 networks 200 steps on the analytic scene (make_golden.py::fx_trained; the weights are inputs
    here) and rendered 1024 held-out rays through its own render_rays, with perturb = 1 (pytest streams) and test-time (perturb =
    0).  The HIP path renders the same rays from the same weights with NOTHING teacher-forced — its own coarse pass, its own
    resampled depths, its own fine pass — and must match the reference's maps to 1e-5 (rgb, acc), 1e-5 * far (depth).  (On the
    random-init fixtures this comparison is bounded at 5e-3 by conditioning; a trained network does not amplify.)"""
    from consistentnerf_amd import run_nerf_view as V
    g = golden("render_rays_trained")
    coarse, fine = _trained_models(g, dev)
    rays = T(g["rays"], dev)
    far = float(g["near_far"][1])
    for tag, perturb in (("p1.", 1.0), ("p0.", 0.0)):
        with torch.no_grad():
            ret = V.render_rays(rays, pytest=True, **_kwargs(coarse, fine, 64, 128, perturb, False, 0.0, False))
        print(f"  perturb = {perturb}:")
        for k, tol in (("rgb0", 5e-6), ("acc0", 5e-6), ("depth0", 5e-6 * far), ("rgb_map", 1e-5), ("acc_map", 1e-5),
                       ("depth_map", 1e-5 * far), ("z_std", 3e-5 * far)):       # measured 1e-6 / 5e-7 / 5e-5
            check(ret[k], g[tag + k], tol, k)
        mse = float(((ret["rgb_map"].cpu() - T(g[tag + "rgb_map"])) ** 2).mean())
        print(f"  PSNR-equivalent of the rgb difference: {-10 * np.log10(max(mse, 1e-30)):.1f} dB")
        assert mse <= 1e-10            # >= 100 dB


def test_render_full_image_and_rays(dev):
    """a1: render(c2w=...) incl. get_rays, viewdirs-before-NDC, ndc_rays; R and V surfaces."""
    from consistentnerf_amd import run_nerf as R, run_nerf_view as V
    from consistentnerf_amd.run_nerf_helpers import get_rays, ndc_rays
    g = golden("render_full_tiny")
    K, c2w = g["K"], g["c2w"]
    ro, rd = get_rays(16, 16, K, T(c2w, dev))
    check(ro, g["rays_o"], 0.0, "rays_o"); check(rd, g["rays_d"], 1e-6, "rays_d")
    no, nd = ndc_rays(16, 16, float(K[0][0]), 1.0, T(g["rays_o"], dev), T(g["rays_d"], dev))
    check(no, g["ndc_o"], 1e-5, "ndc_o"); check(nd, g["ndc_d"], 1e-5, "ndc_d")
    coarse, _ = make_model(4, 128, True, 5, 31, dev)
    fine, _ = make_model(4, 128, True, 5, 32, dev)
    for ndc in (False, True):
        kw = _kwargs(coarse, fine, 16, 16, 0.0, False, 0.0, False)
        if ndc:
            kw.pop("lindisp")
        near, far = (0.0, 1.0) if ndc else (2.0, 6.0)
        with torch.no_grad():
            rgb, disp, acc, depth, extras = V.render(16, 16, K, chunk=100, c2w=T(c2w, dev), ndc=ndc, near=near,
                                                     far=far, use_viewdirs=True, pytest=True, **kw)
            rgb_r, disp_r, acc_r, extras_r = R.render(16, 16, K, chunk=100, c2w=T(c2w, dev), ndc=ndc, near=near,
                                                      far=far, use_viewdirs=True, pytest=True, **kw)
        sfx = "_ndc" if ndc else ""
        tol = 5e-3   # end to end through the resampled fine level: conditioning bound, see test_render_rays_golden
        check(rgb, g["rgb" + sfx], tol, "rgb" + sfx)
        check(acc, g["acc" + sfx], tol, "acc" + sfx)
        check(depth, g["depth" + sfx], tol * far, "depth" + sfx)
        # the coarse level evaluates identical (bit-exact) sample depths on both sides: kernel-level tolerance there.  (Under
        # NDC the ray origins / directions themselves carry the 1e-5 of ndc_rays above, which the encoding amplifies.)
        check(extras["rgb0"], g["rgb0" + sfx], tol if ndc else 2e-5, "rgb0" + sfx)
        assert rgb.shape == (16, 16, 3) and depth.shape == (16, 16)
        assert torch.equal(rgb, rgb_r) and "depth_map" not in extras_r and "depth0" not in extras_r
        assert set(extras_r) == {"rgb0", "disp0", "acc0", "z_std"}


def test_warp_golden(dev):
    from consistentnerf_amd import run_nerf_view as V
    g = golden("warp")
    K = T(g["K"], dev)[None]
    P = T(g["P"], dev)[None, :, None, :]
    w2c = T(g["w2c_ref"], dev)[None]
    c2w = torch.eye(4, device=dev); c2w[:3, :4] = T(g["poses"][1], dev); c2w = c2w[None]
    img = T(g["images"][1], dev).permute(2, 0, 1)[None]
    dep = T(g["depths"][1], dev)[None]
    for tag in ("V", "VT"):
        rgb, d, Xc, ro, rd, mask = V.get_ref_rays(w2c, c2w, K, P, img, dep, variant=tag)
        assert np.array_equal(mask.cpu().numpy(), g[tag + ".mask"]), f"{tag} in-bounds mask"
        check(Xc, g[tag + ".Xc"], 1e-5, tag + ".Xc")
        check(rgb, g[tag + ".rgb_ref"], 0.0, tag + ".rgb_ref")
        check(d, g[tag + ".depth_ref"], 0.0, tag + ".depth_ref")
        check(ro, g[tag + ".rays_o"], 1e-6, tag + ".rays_o")
        check(rd, g[tag + ".rays_d"], 1e-5, tag + ".rays_d")
    y, x, m, z = V.get_test_label(w2c, c2w, K, P, img)
    check(y, g["label.y"], 0.0, "label.y"); check(x, g["label.x"], 0.0, "label.x")
    assert np.array_equal(m.cpu().numpy(), g["label.mask"])
    check(z, g["label.z"], 1e-5, "label.z")


def test_hard_masks_golden(dev):
    from consistentnerf_amd import run_nerf_view as V
    g = golden("hardmask_tiny")
    masks, thr = V.compute_hard_masks(96, 128, g["K"], g["poses"], g["depths"], list(g["i_train"]), 0.1, 5120,
                                      device=dev, return_thresholds=True)
    diff = (masks != g["masks"])
    print(f"  mask pixels differing: {diff.sum()} of {diff.size}")
    assert diff.sum() == 0, "hard masks must match the reference capture"
    for (t, r, c, th) in g["thr"]:
        got = thr[(int(t), int(r))][int(c)]
        assert (np.isnan(th) and np.isnan(got)) or np.float32(th) == got, (t, r, c, th, got)
    assert not masks[3].any()


def test_masked_losses_golden(dev):
    from consistentnerf_amd import run_nerf_view as V
    g = golden("losses_mask")
    far, c = float(g["far"]), float(g["coef"])
    for tag, m in (("mixed", g["mask"]), ("allone", np.ones_like(g["mask"]))):
        r = T(g["rgb"], dev).requires_grad_(True)
        d = T(g["depth"], dev).requires_grad_(True)
        lr, ld = V.hardmask_losses(r, T(g["target"], dev), T(m, dev), c, d, T(g["prior"], dev), far)
        check(lr, g[tag + ".l_rgb"], 2e-7, tag + ".l_rgb"); check(ld, g[tag + ".l_depth"], 2e-7, tag + ".l_depth")
        (lr + ld).backward()
        check(r.grad, g[tag + ".d_rgb"], 1e-9, tag + ".d_rgb"); check(d.grad, g[tag + ".d_depth"], 1e-9, tag + ".d_depth")
    lr, _ = V.hardmask_losses(T(g["rgb"], dev), T(g["target"], dev), None)
    check(V.mse2psnr(lr), g["psnr"], 1e-5, "psnr (unmasked)")


def test_train_10_steps_golden(dev):
    """create_nerf -> render -> loss -> backward -> FusedAdam + lr decay, 10 steps at C1 shapes."""
    import argparse
    import tempfile
    from consistentnerf_amd import run_nerf as R
    g = golden("train_10steps_C1")
    with tempfile.TemporaryDirectory() as tmp:
        args = argparse.Namespace(
            multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=0, netdepth=4, netwidth=128,
            netdepth_fine=4, netwidth_fine=128, netchunk=1024 * 64, lrate=5e-4, basedir=tmp, expname="exp",
            ft_path=None, no_reload=True, perturb=1.0, N_samples=64, white_bkgd=True, raw_noise_std=0.0,
            dataset_type="blender", no_ndc=False, lindisp=False)
        kw_train, kw_test, start, grad_vars, optimizer = R.create_nerf(args)
    sd = I.nerf_state_dict(4, 128, 10, 4, 4, True, seed=41)
    kw_train["network_fn"].load_state_dict({k: T(v) for k, v in sd.items()})
    kw_train.update(near=2.0, far=6.0)
    K = I.intrinsics(100, 100, 138.0)
    global_step = start
    for i in range(10):
        rays = T(I.ray_batch(256, seed=100 + i), dev)
        target = T(np.random.RandomState(200 + i).uniform(size=(256, 3)).astype(np.float32), dev)
        batch_rays = torch.stack([rays[:, 0:3], rays[:, 3:6]], 0)
        rgb, disp, acc, extras = R.render(100, 100, K, chunk=32768, rays=batch_rays, retraw=True, pytest=True,
                                          **kw_train)
        optimizer.zero_grad()
        loss = R.img2mse(rgb, target)
        rel = abs(loss.item() - g["losses"][i]) / g["losses"][i]
        print(f"  step {i}: loss {loss.item():.6f} ref {g['losses'][i]:.6f} rel {rel:.2e}")
        assert rel < 2e-4, (i, loss.item(), g["losses"][i])
        loss.backward()
        optimizer.step()
        new_lrate = args.lrate * (0.1 ** (global_step / (250 * 1000)))
        for pg in optimizer.param_groups:
            pg["lr"] = new_lrate
        global_step += 1
    worst = 0.0
    for k, v in kw_train["network_fn"].state_dict().items():
        worst = max(worst, float(np.abs(v.cpu().numpy() - g["final." + k]).max()))
    print(f"  final weights: max|d| = {worst:.3e}")
    assert worst < 2e-4   # 10 Adam steps of lr 5e-4: a sign flip of a ~0 gradient moves a weight by 1e-3 at most
    assert kw_test["perturb"] is False and kw_test["raw_noise_std"] == 0.


def test_checkpoint_round_trip(dev):
    """save_checkpoint (R:836-845 keys) -> create_nerf reload (R:226-243): 3 steps, save, reload into fresh objects,
    2 more steps == 5 uninterrupted steps bit for bit (weights, Adam moments, step count, lr); the optimizer state
    has torch.optim.Adam's structure."""
    import argparse
    import tempfile
    from consistentnerf_amd import run_nerf as R

    def mk(tmp, no_reload):
        return argparse.Namespace(
            multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=16, netdepth=2, netwidth=64,
            netdepth_fine=2, netwidth_fine=64, netchunk=1024 * 64, lrate=5e-4, basedir=tmp, expname="exp",
            ft_path=None, no_reload=no_reload, perturb=0.0, N_samples=16, white_bkgd=False, raw_noise_std=0.0,
            dataset_type="blender", no_ndc=False, lindisp=False)
    K = I.intrinsics(100, 100, 138.0)

    def steps(kw, opt, args, lo, hi):
        for i in range(lo, hi):
            rays = T(I.ray_batch(64, seed=300 + i), dev)
            target = T(np.random.RandomState(400 + i).uniform(size=(64, 3)).astype(np.float32), dev)
            rgb, _, _, extras = R.render(100, 100, K, chunk=32768, rays=torch.stack([rays[:, 0:3], rays[:, 3:6]], 0),
                                         retraw=True, near=2.0, far=6.0, **kw)
            opt.zero_grad()
            (R.img2mse(rgb, target) + R.img2mse(extras['rgb0'], target)).backward()
            opt.step()
            for pg in opt.param_groups:
                pg["lr"] = args.lrate * (0.1 ** (i / 250000))

    def weights(kw):
        return torch.cat([p.detach().reshape(-1) for m in (kw['network_fn'], kw['network_fine']) for p in m.parameters()])
    with tempfile.TemporaryDirectory() as tmp:
        torch.manual_seed(7)
        a = mk(tmp, True)
        kw, _, start, _, opt = R.create_nerf(a)
        init = {n: {k: v.clone() for k, v in kw[n].state_dict().items()} for n in ('network_fn', 'network_fine')}
        steps(kw, opt, a, 0, 5)
        ref_w, ref_m, ref_v = weights(kw).clone(), opt.exp_avg.clone(), opt.exp_avg_sq.clone()
        # interrupted run from the same initial weights
        kw1, _, _, _, opt1 = R.create_nerf(a)
        for n in init:
            kw1[n].load_state_dict(init[n])
        steps(kw1, opt1, a, 0, 3)
        path = R.save_checkpoint(tmp, "exp", 3, kw1, opt1)
        ck = torch.load(path, map_location="cpu", weights_only=False)
        assert set(ck) == {'global_step', 'network_fn_state_dict', 'network_fine_state_dict', 'optimizer_state_dict'}
        osd = ck['optimizer_state_dict']
        assert set(osd) == {'state', 'param_groups'} and set(osd['state'][0]) == {'step', 'exp_avg', 'exp_avg_sq'}
        cpu_twin = [torch.nn.Parameter(p.detach().cpu().clone()) for p in opt1.params]
        torch.optim.Adam(cpu_twin, lr=1e-3).load_state_dict(osd)        # the reference's optimizer accepts it
        kw2, _, start2, _, opt2 = R.create_nerf(mk(tmp, False))
        assert start2 == 3 and opt2._step == 3 and abs(opt2.param_groups[0]['lr'] - opt1.param_groups[0]['lr']) < 1e-12
        steps(kw2, opt2, a, 3, 5)
        assert torch.equal(weights(kw2), ref_w) and torch.equal(opt2.exp_avg, ref_m) and torch.equal(opt2.exp_avg_sq, ref_v)


# ------------------------------------------------------------------------------------------------
# full-size (BASELINE config C2: 4096 rays, 64+128 samples, D=8/W=256) property tests
def _c2(dev, B=4096):
    coarse, _ = make_model(8, 256, True, 5, 21, dev)
    fine, _ = make_model(8, 256, True, 5, 22, dev)
    rays = T(I.ray_batch(B, seed=5, near=2.125, far=4.67), dev)
    return coarse, fine, rays


def test_c2_properties(dev):
    from consistentnerf_amd import run_nerf_view as V
    coarse, fine, rays = _c2(dev)
    kw = _kwargs(coarse, fine, 64, 128, 1.0, False, 0.0, False)
    torch.manual_seed(0)
    with torch.no_grad():
        full = V.render_rays(rays, retraw=True, pytest=True, **kw)
        # chunk invariance (R:79-80): 4096 rays at once == 4 chunks of 1024 (pytest RNG is per-call, so compare
        # on the deterministic test-time path)
        kwt = dict(kw, perturb=0.0)
        a = V.render_rays(rays, **kwt)
        parts = [V.render_rays(rays[i:i + 1024], **kwt) for i in range(0, 4096, 1024)]
    for k in ("rgb_map", "depth_map", "acc_map", "rgb0"):
        assert torch.equal(a[k], torch.cat([p[k] for p in parts])), f"chunk invariance {k}"
    for k, v in full.items():
        if k.startswith("disp"):   # disp is NaN exactly where acc == 0 (reference behaviour, R:302)
            acc = full["acc_map" if k == "disp_map" else "acc0"]
            assert torch.equal(torch.isnan(v), acc == 0), k
        else:
            assert torch.isfinite(v).all(), k
    assert (full["acc_map"] <= 1 + 1e-5).all() and (full["acc_map"] >= 0).all()
    assert (full["rgb_map"] >= -1e-6).all() and (full["rgb_map"] <= 1 + 1e-5).all()
    assert (full["depth_map"] <= 4.67 * 1.0001).all()
    # oracle on a 128-ray slice of the same batch (CPU finishes in seconds)
    sl = slice(1000, 1128)
    sdc = O.as_tensors(I.nerf_state_dict(8, 256, 10, 4, 5, True, seed=21))
    sdf = O.as_tensors(I.nerf_state_dict(8, 256, 10, 4, 5, True, seed=22))
    with torch.no_grad():
        got = V.render_rays(rays[sl], pytest=True, **kw)
        ref = O.render_rays_pytest(rays[sl].cpu(), sdc, sdf, O.NetCfg(8, 256, output_ch=5), O.RenderCfg(64, 128, 1.0))
    check(got["rgb0"], ref["rgb0"], 2e-5, "C2 rgb0 vs oracle")
    check(got["depth0"], ref["depth0"], 2e-5 * 4.67, "C2 depth0 vs oracle")
    check(got["rgb_map"], ref["rgb_map"], 5e-3, "C2 rgb_map vs oracle (end to end)")
    check(got["depth_map"], ref["depth_map"], 5e-3 * 4.67, "C2 depth_map vs oracle (end to end)")


def test_c2_gradient_linearity(dev):
    """Backward is linear in the upstream gradient: grad(2L) == 2 grad(L), and two half-batches accumulate to
    the full batch (size-independent checks of dgrad + wgrad + split reduction at full size)."""
    from consistentnerf_amd import run_nerf_view as V
    coarse, fine, rays = _c2(dev, 2048)
    kw = _kwargs(coarse, fine, 64, 128, 0.0, False, 0.0, False)
    tgt = torch.rand(2048, 3, device=dev)

    def grads(scale, sl):
        for m in (coarse, fine):
            m.zero_grad(set_to_none=True)
        out = V.render_rays(rays[sl], **kw)
        loss = scale * (((out["rgb_map"] - tgt[sl]) ** 2).sum() + ((out["rgb0"] - tgt[sl]) ** 2).sum())
        loss.backward()
        return torch.cat([p.grad.reshape(-1) for m in (coarse, fine) for p in m.kernel_tensors()
                          if p.grad is not None])
    g1 = grads(1.0, slice(0, 2048))
    g2 = grads(2.0, slice(0, 2048))
    ga = grads(1.0, slice(0, 1024)) + grads(1.0, slice(1024, 2048))
    s = g1.abs().max().item()
    assert (g2 - 2 * g1).abs().max().item() <= 1e-5 * s
    assert (ga - g1).abs().max().item() <= 2e-4 * s
    assert torch.isfinite(g1).all() and s > 0
    # the split-partial reduction runs in a fixed order: the same inputs give the same bits, run to run
    assert torch.equal(grads(1.0, slice(0, 2048)), g1), "weight gradients are not bit-reproducible"


def test_direct_accumulation_into_flat_grads(dev):
    """FusedAdam owns the flat gradient buffer and _MlpFn.backward accumulates into its views without returning
    per-tensor gradients to autograd: two backward passes must add up, zero_grad must clear, and the values must equal
    the tensor route taken by plain nn.Parameters."""
    from consistentnerf_amd import run_nerf_view as V
    from consistentnerf_amd.optim import FusedAdam
    coarse, fine, rays = _c2(dev, 256)
    kw = _kwargs(coarse, fine, 64, 128, 0.0, False, 0.0, False)
    tgt = torch.rand(256, 3, device=dev)

    def run():
        out = V.render_rays(rays, **kw)
        (((out["rgb_map"] - tgt) ** 2).sum() + ((out["rgb0"] - tgt) ** 2).sum()).backward()
    for m in (coarse, fine):
        m.zero_grad(set_to_none=True)
    run()                                                    # tensor route (AccumulateGrad)
    ref = torch.cat([p.grad.reshape(-1) for m in (coarse, fine) for p in m.kernel_tensors()]).clone()
    opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
    opt.zero_grad()
    run()
    got = torch.cat([p.grad.reshape(-1) for m in (coarse, fine) for p in m.kernel_tensors()])
    assert torch.equal(got, ref)
    run()
    got2 = torch.cat([p.grad.reshape(-1) for m in (coarse, fine) for p in m.kernel_tensors()])
    assert (got2 - 2 * ref).abs().max().item() <= 1e-6 * ref.abs().max().item()
    # zero_grad() DROPS the gradient (torch's set_to_none=True default; no fill launch): the next backward overwrites it
    opt.zero_grad()
    assert float(opt.flat_grad.abs().max()) > 0.0
    run()
    got3 = torch.cat([p.grad.reshape(-1) for m in (coarse, fine) for p in m.kernel_tensors()])
    assert torch.equal(got3, ref)
    # only the coarse network gets a gradient: its views are overwritten, the fine network's dropped ones read as zeros by the
    # time the optimizer (or anything else that goes through materialize_grad) looks at them
    opt.zero_grad()
    kw0 = _kwargs(coarse, None, 64, 0, 0.0, False, 0.0, False)
    ((V.render_rays(rays, **kw0)["rgb_map"] - tgt) ** 2).sum().backward()
    opt.materialize_grad()
    lo, hi = opt.slice_of(list(fine.parameters()))
    assert float(opt.flat_grad[lo:hi].abs().max()) == 0.0
    lo, hi = opt.slice_of(list(coarse.parameters()))
    assert float(opt.flat_grad[lo:hi].abs().max()) > 0.0
    # the tensor route (autograd's AccumulateGrad adds into the view) zeroes a dropped view first
    opt.zero_grad()
    w = coarse.pts_linears[0].weight
    (w * 2.0).sum().backward()
    assert torch.equal(w.grad, torch.full_like(w, 2.0))
    opt.zero_grad(set_to_none=False)
    assert float(opt.flat_grad.abs().max()) == 0.0


def test_detached_gradients_accumulate_over_two_backwards_then_step(dev):
    """FusedAdam, `p.grad = None` (model.zero_grad()), TWO backwards, step().  The first backward leaves the
    gradient in a tensor of autograd's own; the second used to take the direct route into that foreign tensor with
    accumulate=0 (the first gradient lost) and Adam then stepped on the stale flat buffer.  Now: the direct route is taken only
    while .grad IS the flat view, both backwards add up, materialize_grad() copies the sum into the flat buffer, and the
    updated weights equal those of the plain two-backward accumulation into attached views."""
    from consistentnerf_amd import run_nerf_view as V
    from consistentnerf_amd.optim import FusedAdam

    def build():
        coarse, fine, rays = _c2(dev, 128)
        opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
        return coarse, fine, rays, opt
    tgt = torch.rand(128, 3, device=dev)

    def run(coarse, fine, rays, scale):
        kw = _kwargs(coarse, fine, 64, 128, 0.0, False, 0.0, False)
        out = V.render_rays(rays, **kw)
        (scale * (((out["rgb_map"] - tgt) ** 2).sum() + ((out["rgb0"] - tgt) ** 2).sum())).backward()
    # reference: views stay attached, zero_grad(), two backwards (direct route, accumulate), step
    c0, f0, rays, o0 = build()
    o0.zero_grad()
    run(c0, f0, rays, 2.0)
    o0.step()                                             # (a warm step: the flat views now hold a LIVE gradient of another scale)
    o0.zero_grad()
    run(c0, f0, rays, 1.0)
    run(c0, f0, rays, 0.5)
    g_ref = o0.flat_grad.clone()
    o0.step()
    # detached: p.grad = None on every parameter, the same two backwards, step
    c1, f1, rays1, o1 = build()
    o1.zero_grad()
    run(c1, f1, rays1, 2.0)
    o1.step()
    for m in (c1, f1):
        m.zero_grad(set_to_none=True)                     # the flat views keep the warm step's gradient: stale from here on
    assert all(p.grad is None for p in o1.params)
    run(c1, f1, rays1, 1.0)
    first = [p.grad.clone() for p in o1.params if p.grad is not None]
    assert first and all(p.grad is None or p.grad.data_ptr() != p._cnerf_view_ptr for p in o1.params)
    run(c1, f1, rays1, 0.5)
    second = [p.grad for p in o1.params if p.grad is not None]
    a_max = max(float(a.abs().max()) for a in first)
    assert max(float((b - 1.5 * a).abs().max()) for a, b in zip(first, second)) <= 2e-6 * a_max, "the second backward did not ADD"
    o1.materialize_grad()
    assert all(p.grad.data_ptr() == p._cnerf_view_ptr for p in o1.params)
    reached = torch.zeros_like(o1.flat_grad, dtype=torch.bool)
    for p, o in zip(o1.params, o1._offsets):
        if p.requires_grad and float(g_ref[o:o + p.numel()].abs().max()) > 0:
            reached[o:o + p.numel()] = True
    s = float(g_ref.abs().max())
    assert float((o1.flat_grad - g_ref)[reached].abs().max()) <= 2e-6 * s
    assert float(o1.flat_grad[~reached].abs().max()) == 0.0, "a parameter no backward reached steps on stale values"
    o1.step()
    assert float((o1.flat_param - o0.flat_param).abs().max()) <= 1e-6
    # and once more through zero_grad(): views re-attached, the direct route is back and overwrites
    o1.zero_grad()
    run(c1, f1, rays1, 1.0)
    o0.zero_grad()
    run(c0, f0, rays, 1.0)
    assert all(p.grad.data_ptr() == p._cnerf_view_ptr for p in o1.params)
    o0.materialize_grad(); o1.materialize_grad()
    assert float((o1.flat_grad - o0.flat_grad).abs().max()) <= 2e-5 * float(o0.flat_grad.abs().max())


# ------------------------------------------------------------------------------------------------
# secondary surface: NeRF.forward on pre-embedded inputs (H:107-130) and the corners of the compiled envelope
@pytest.mark.parametrize("D,W,vd,och", [(8, 256, True, 5), (4, 128, False, 5)])
def test_nerf_module_forward_on_embedded_inputs(dev, D, W, vd, och):
    model, sd = make_model(D, W, vd, och, 11, dev)
    rs = np.random.RandomState(8)
    M = 203
    pts = rs.uniform(-2, 2, size=(M, 3)).astype(np.float32)
    dirs = rs.normal(size=(M, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    xe = torch.cat([O.embed(T(pts), 10)] + ([O.embed(T(dirs), 4)] if vd else []), -1)
    out = model(xe.to(dev))
    sdt = O.as_tensors(sd, True)
    ref = O.mlp_forward(sdt, xe[:, :63], xe[:, 63:] if vd else None, O.NetCfg(D, W, use_viewdirs=vd, output_ch=och))
    check(out, ref.detach(), 3e-5 * max(1.0, float(ref.abs().max())), "NeRF.forward(embedded)")
    G = T(rs.normal(size=tuple(ref.shape)).astype(np.float32))
    (out * G.to(dev)).sum().backward()
    (ref * G).sum().backward()
    gref = {"g." + k: (p.grad if p.grad is not None else torch.zeros_like(p)).numpy() for k, p in sdt.items()}
    check_param_grads(model, gref, "g.", "g.", rtol=5e-2, l2tol=5e-3)


@pytest.mark.parametrize("D,W,vd,och,multires,i_embed", [(2, 64, True, 4, 10, 0), (6, 64, False, 4, 10, 0),
                                                          (8, 256, True, 5, 0, -1), (3, 128, True, 4, 4, 0),
                                                          (1, 256, True, 4, 10, 0), (16, 128, False, 8, 10, 0),
                                                          (9, 256, False, 4, 7, 0), (4, 64, True, 4, 1, 0)])
def test_envelope_corners_vs_oracle(dev, D, W, vd, och, multires, i_embed):
    """W=64 (one view-branch tile: wave 1 idles there), D=2, D=6 (skip right before the last layer), identity
    embedding (i_embed=-1: 3 input channels padded to 32) — forward + gradients against the CPU oracle."""
    from consistentnerf_amd.run_nerf import run_network
    from consistentnerf_amd.run_nerf_helpers import NeRF, get_embedder
    L = -1 if i_embed == -1 else multires
    Ld = -1 if i_embed == -1 else 4
    in_ch, dir_ch = O.embed_dim(L), (O.embed_dim(Ld) if vd else 0)
    rs = np.random.RandomState(3)
    model = NeRF(D=D, W=W, input_ch=in_ch, output_ch=och, skips=[4], input_ch_views=dir_ch, use_viewdirs=vd)
    with torch.no_grad():
        for p_ in model.parameters():
            if p_.dim() == 2:
                b = float(np.sqrt(6.0 / p_.shape[1]))
                p_.copy_(T(rs.uniform(-b, b, size=tuple(p_.shape)).astype(np.float32)))
            elif p_.numel() > 1:
                p_.copy_(T(rs.uniform(-0.1, 0.1, size=tuple(p_.shape)).astype(np.float32)))
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.to(dev)
    B, S = 19, 11
    pts = rs.uniform(-2, 2, size=(B, S, 3)).astype(np.float32)
    dirs = rs.normal(size=(B, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    e, _ = get_embedder(multires, i_embed)
    ed = get_embedder(4, i_embed)[0] if vd else None
    raw = run_network(T(pts, dev), T(dirs, dev) if vd else None, model, e, ed)
    cfg = O.NetCfg(D=D, W=W, multires=L, multires_views=Ld, use_viewdirs=vd, output_ch=och)
    ref = O.query(sd, T(pts), T(dirs) if vd else None, cfg)
    check(raw, ref.detach(), 3e-5 * max(1.0, float(ref.abs().max())), "raw")
    G = T(rs.normal(size=tuple(ref.shape)).astype(np.float32))
    (raw * G.to(dev)).sum().backward()
    (ref * G).sum().backward()
    gref = {"g." + k: (p.grad if p.grad is not None else torch.zeros_like(p)).numpy() for k, p in sd.items()}
    # one ReLU mask that differs from the CPU's (pre-activation within round-off of 0) moves every upstream gradient by
    # ~1/M of its scale, M = 209 points here: a 16-layer net collects a few of them (the tight check of these corners is
    # test_mlp_backward_exact_from_stash, which replays the backward from the kernel's own masks)
    deep = D >= 12
    check_param_grads(model, gref, "g.", "g.", rtol=1e-1 if deep else 5e-2, l2tol=3e-2 if deep else 5e-3)


def _sweep_cases(n=14, seed=20260929):
    """Seeded draws from the compiled envelope (D 1-10, W 64 | 128 | 256, view branch or not, 4 / 5 / 8 output channels, ragged
    point counts incl. single points and non-multiples of 32): (D, W, vd, och, B, S, weight seed)."""
    rs = np.random.RandomState(seed)
    out = []
    for i in range(n):
        vd = bool(rs.randint(0, 2))
        out.append((int(rs.randint(1, 11)), int(rs.choice([64, 128, 256])), vd, int(rs.choice([4, 5] if vd else [4, 5, 8])),
                    int(rs.choice([1, 2, 7, 19, 33, 64])), int(rs.choice([1, 3, 8, 17, 32])), 100 + i))
    return out


@pytest.mark.parametrize("D,W,vd,och,B,S,wseed", _sweep_cases())
def test_random_networks_vs_oracle(dev, D, W, vd, och, B, S, wseed):
    """Seeded sweep over the envelope: forward (3e-5 max|raw|) and — on the kernel's own ReLU branch, every pattern difference
    bounded at the kink — parameter gradients (1e-5 max per tensor) against the CPU oracle, explicit-points mode."""
    from consistentnerf_amd.run_nerf import run_network
    from consistentnerf_amd.run_nerf_helpers import get_embedder
    model, sd_np = make_model(D, W, vd, och, wseed, dev)
    rs = np.random.RandomState(wseed)
    pts = rs.uniform(-2, 2, size=(B, S, 3)).astype(np.float32)
    dirs = rs.normal(size=(B, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    e, _ = get_embedder(10, 0)
    ed = get_embedder(4, 0)[0] if vd else None
    raw = run_network(T(pts, dev), T(dirs, dev) if vd else None, model, e, ed)
    masks = relu_masks(raw.grad_fn.stash, B * S, D, W, vd)
    sd = O.as_tensors(sd_np, True)
    cfg = O.NetCfg(D, W, use_viewdirs=vd, output_ch=och)
    flips = []
    ref = O.query(sd, T(pts), T(dirs) if vd else None, cfg, masks, flips)
    check_flips(flips)
    check(raw, ref.detach(), 3e-5 * max(1.0, float(ref.detach().abs().max())), "raw")
    G = T(rs.normal(size=tuple(ref.shape)).astype(np.float32))
    (raw * G.to(dev)).sum().backward()
    (ref * G).sum().backward()
    tight = {"t." + k: (p_.grad if p_.grad is not None else torch.zeros_like(p_)).numpy() for k, p_ in sd.items()}
    check_param_grads(model, tight, "t.", "t.", rtol=1e-5, l2tol=1e-5)


# ------------------------------------------------------------------------------------------------
# training-ray supply
def test_ray_bank_and_samplers_golden(dev):
    """RayBank / sample_image_rays (device-side rays from cnerf_gen_rays) against the reference's train() statements
    replayed on the host (fixture `raybank`): same rows, same batches across an epoch end, same --no_batching picks.
    Colours and origins are copies (bit-exact); directions go through the ray kernel (1e-6)."""
    from consistentnerf_amd.raybank import RayBank, sample_image_rays
    g = golden("raybank")
    Hh, Ww, focal = int(g["hwf"][0]), int(g["hwf"][1]), float(g["hwf"][2])
    K = I.intrinsics(Hh, Ww, focal)
    n = g["bank0"].shape[0]
    bank = RayBank(g["images"], g["poses"], Hh, Ww, K, g["i_train"], device=dev, perm=O.numpy_shuffle_perm(n, 5))
    assert len(bank) == n
    check(bank.rays_rgb, g["bank0"], 1e-6, "shuffled bank")
    assert torch.equal(bank.rays_rgb[:, 2].cpu(), T(g["bank0"][:, 2]))
    also = RayBank(g["images"], g["poses"], Hh, Ww, K, g["i_train"], device=dev, seed=5)     # seed == that permutation
    assert torch.equal(also.rays_rgb, bank.rays_rgb)
    for it in range(4):
        rays, tgt = bank.next_batch(50, g["rand_idx"])
        check(rays, g[f"rays{it}"], 1e-6, f"batch {it} rays")
        assert torch.equal(tgt.cpu(), T(g[f"tgt{it}"]))
    assert bank.epochs == 1 and bank.i_batch == 50
    for tag, frac in (("full", None), ("crop", 0.5)):
        rays, tgt = sample_image_rays(T(g["images"][2], dev), g["poses"][2], Hh, Ww, K, len(g[f"nb_{tag}_inds"]), frac,
                                      g[f"nb_{tag}_inds"])
        check(rays, g[f"nb_{tag}_rays"], 1e-6, f"no_batching {tag} rays")
        assert torch.equal(tgt.cpu(), T(g[f"nb_{tag}_tgt"]))
    # device-side draws: a permutation of the same rows / distinct pixels
    rnd = RayBank(g["images"], g["poses"], Hh, Ww, K, g["i_train"], device=dev)
    assert torch.equal(torch.sort(rnd.rays_rgb.reshape(n, -1)[:, 8])[0], torch.sort(bank.rays_rgb.reshape(n, -1)[:, 8])[0])
    rays, tgt = sample_image_rays(T(g["images"][2], dev), g["poses"][2], Hh, Ww, K, 20)
    assert rays.shape == (2, 20, 3) and tgt.shape == (20, 3)


def test_exact_zero_preactivations(dev):
    """A layer whose weights and bias are all zero gives pre-activations that are exactly 0: ReLU'(0) = 0 (torch), the
    packed-math ReLU must not turn 0 * inf into a NaN, the sign bits must read 0 and no gradient may pass."""
    from consistentnerf_amd.run_nerf import run_network
    from consistentnerf_amd.run_nerf_helpers import NeRF, get_embedder
    rs = np.random.RandomState(12)
    model = NeRF(D=3, W=64, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
    with torch.no_grad():
        model.pts_linears[1].weight.zero_()
        model.pts_linears[1].bias.zero_()
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items()}
    model = model.to(dev)
    pts = rs.uniform(-1, 1, size=(7, 9, 3)).astype(np.float32)
    dirs = rs.normal(size=(7, 3)).astype(np.float32)
    e, ed = get_embedder(10, 0)[0], get_embedder(4, 0)[0]
    raw = run_network(T(pts, dev), T(dirs, dev), model, e, ed)
    assert torch.isfinite(raw).all()
    ref = O.query(sd, T(pts), T(dirs), O.NetCfg(D=3, W=64, use_viewdirs=True, output_ch=4))
    check(raw, ref.detach(), 3e-5 * max(1.0, float(ref.abs().max())), "raw")
    raw.square().sum().backward()
    ref.square().sum().backward()
    for k, p in model.named_parameters():
        g, gr = p.grad, sd[k].grad
        assert g is None or torch.isfinite(g).all(), k
        if k.startswith("pts_linears.0") or k.startswith("pts_linears.1"):
            assert g is None or float(g.abs().max()) == 0.0, f"{k}: gradient leaked through ReLU'(0)"
            assert gr is None or float(gr.abs().max()) == 0.0


@pytest.mark.parametrize("M", [1, 31, 32, 33, 65])
def test_ragged_batch_sizes(dev, M):
    """Batches that do not fill 32-point tiles (padding lanes address out of range): every point's output is
    bit-identical to the same point inside a large batch, and the parameter gradients of the ragged batch equal those
    of the same points zero-weighted inside the large batch (padding rows contribute exactly nothing)."""
    model, _ = make_model(8, 256, True, 5, 11, dev)
    rs = np.random.RandomState(77)
    big = 160
    pts = rs.uniform(-2, 2, size=(big, 3)).astype(np.float32)
    dirs = rs.normal(size=(big, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=-1, keepdims=True)
    xe = torch.cat([O.embed(T(pts), 10), O.embed(T(dirs), 4)], -1).to(dev)
    G = T(rs.normal(size=(big, 4)).astype(np.float32), dev)
    out_big = model(xe)
    out = model(xe[:M])
    assert torch.equal(out, out_big[:M])
    model.zero_grad(set_to_none=True)
    (out * G[:M]).sum().backward()
    g_small = torch.cat([p.grad.reshape(-1) for p in model.kernel_tensors()]).clone()
    model.zero_grad(set_to_none=True)
    Gz = G.clone()
    Gz[M:] = 0
    (model(xe) * Gz).sum().backward()
    g_big = torch.cat([p.grad.reshape(-1) for p in model.kernel_tensors()])
    assert torch.isfinite(g_small).all()
    assert (g_small - g_big).abs().max().item() <= 2e-5 * max(1e-12, g_big.abs().max().item())


def test_sharded_render_path_equals_render_path(dev):
    """§8e inference: a frame rendered as row blocks through render(rays=...) (what every rank of
    distributed.render_path_sharded does; world 1 here, plus a hand-made 3-way split with edge padding) equals
    render_path(c2w=...) bit for bit, NDC on and off."""
    from consistentnerf_amd import run_nerf as R, distributed as D
    from consistentnerf_amd.run_nerf_helpers import get_rays
    g = golden("render_full_tiny")
    K, c2w = g["K"], T(g["c2w"], dev)
    coarse, _ = make_model(4, 128, True, 5, 31, dev)
    fine, _ = make_model(4, 128, True, 5, 32, dev)
    H, W = 13, 16
    for ndc in (False, True):
        kw = _kwargs(coarse, fine, 16, 16, 0.0, False, 0.0, False)
        if ndc:
            kw.pop("lindisp")
        near, far = (0.0, 1.0) if ndc else (2.0, 6.0)
        kw.update(ndc=ndc, near=near, far=far, use_viewdirs=True)
        kw["network_query_fn"]._cnerf_stock = True      # (_kwargs' query IS create_nerf's stock one: same embedders)
        pose4 = torch.cat([c2w[:3, :4], torch.tensor([[0., 0., 0., 1.]], device=dev)], 0)
        rgbs, disps = R.render_path([pose4], (H, W, float(K[0][0])), K, 100, kw)
        rgbs_s, disps_s = D.render_path_sharded([pose4], (H, W, float(K[0][0])), K, 100, kw)
        assert np.array_equal(rgbs, rgbs_s) and np.array_equal(disps, disps_s, equal_nan=True)
        # the per-rank work of a 3-rank job, done serially: blocks 5+4+4 rows, each padded to 5
        ro, rd = get_rays(H, W, K, c2w[:3, :4])
        parts = []
        for r in range(3):
            lo, hi, idx = D.row_block(H, r, 3)
            assert len(idx) == 5
            idx = idx.to(dev)
            with torch.no_grad():
                out = R.render(H, W, K, chunk=100, rays=torch.stack([ro[idx], rd[idx]], 0), **kw)
            parts.append(out[0][:hi - lo])
        assert np.array_equal(torch.cat(parts, 0).cpu().numpy(), rgbs[0])
        # what the ranks of render_path_sharded do now: each renders ITS rows through the in-kernel camera path
        # (cam.first = lo * W; no ray tensor), 3 ranks done serially
        with torch.no_grad():
            assert R.camera_path_ok(c2w[:3, :4], kw)
        assert not R.camera_path_ok(c2w[:3, :4], kw)       # (grad mode + trainable parameters: an autograd graph is wanted)
        parts, dparts = [], []
        for r in range(3):
            lo, hi = D.shard_bounds(H, r, 3)
            with torch.no_grad():
                o = R.render_pixels(H, W, K, 100, c2w[:3, :4], lo * W, (hi - lo) * W, **kw)
            assert o["rgb_map"].shape == ((hi - lo) * W, 3)
            parts.append(o["rgb_map"])
            dparts.append(o["disp_map"])
        assert np.array_equal(torch.cat(parts, 0).reshape(H, W, 3).cpu().numpy(), rgbs[0])
        assert np.array_equal(torch.cat(dparts, 0).reshape(H, W).cpu().numpy(), disps[0], equal_nan=True)
    with pytest.raises(Exception):           # a training-mode call (autograd graph wanted) has no camera path
        R.render_pixels(H, W, K, 100, c2w[:3, :4], 0, W, **kw)


SHARDED_RENDER_WORKER = r"""
import os, sys, numpy as np, torch
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _inputs as I
from conftest import golden
from test_gpu_parity import make_model, _kwargs, T
from consistentnerf_amd import distributed as D, run_nerf as R, run_nerf_view as V
import torch.distributed as dist
dev = torch.device("cuda:0")                       # both ranks share the box's one GPU; the gather goes through gloo
rank, world, _ = D.init_from_env("gloo")
assert world == 2
g = golden("render_full_tiny")
K, c2w = g["K"], T(g["c2w"], dev)
coarse, _ = make_model(4, 128, True, 5, 31, dev)
fine, _ = make_model(4, 128, True, 5, 32, dev)
H, W = 13, 16                                      # 13 rows over 2 ranks: 7 + 6, rank 1's block is padded by one row
kw = _kwargs(coarse, fine, 16, 16, 0.0, False, 0.0, False)
kw.update(ndc=False, near=2.0, far=6.0, use_viewdirs=True)
kw["network_query_fn"]._cnerf_stock = True         # (_kwargs' query IS create_nerf's stock one: same embedders)
pose4 = torch.cat([c2w[:3, :4], torch.tensor([[0., 0., 0., 1.]], device=dev)], 0)
poses = [pose4, pose4.clone(), pose4.clone()]
poses[1][0, 3] += 0.2; poses[2][1, 3] -= 0.3
calls = []
orig = R.render_pixels
def spy(H_, W_, K_, chunk, c2w_, first, count, **k):
    calls.append((first, count)); return orig(H_, W_, K_, chunk, c2w_, first, count, **k)
R.render_pixels = spy
ft = []
res = D.render_path_sharded(poses, (H, W, float(K[0][0])), K, 100, kw, frame_times=ft)
res_v = D.render_path_sharded(poses[:1], (H, W, float(K[0][0])), K, 100, kw, want_acc=True)
lo, hi = D.shard_bounds(H)
assert calls == [(lo * W, (hi - lo) * W)] * 4, calls          # own rows only, through the in-kernel camera path
if rank == 0:
    rgbs, disps = R.render_path(poses, (H, W, float(K[0][0])), K, 100, kw)
    assert np.array_equal(res[0], rgbs) and np.array_equal(res[1], disps, equal_nan=True)
    rv, dv, av = V.render_path(poses[:1], (H, W, float(K[0][0])), K, 100, kw)
    assert np.array_equal(res_v[0], rv) and np.array_equal(res_v[2], av)
    assert len(ft) == 3
else:
    assert res is None and res_v is None
D.barrier()
if rank == 0:
    print("SHARDED_RENDER_OK", world, [round(t, 4) for t in ft])
dist.destroy_process_group()
"""


def test_two_ranks_sharded_render_path_on_one_gpu(dev, tmp_path):
    """distributed.render_path_sharded with WORLD SIZE 2 (both ranks on the box's one MI355X, gloo): every rank renders only
    its own rows through the in-kernel camera path (no [H*W, 11] ray tensor on any rank), ONE gather of the packed rgb|disp
    (|acc) block to rank 0 per frame, frame i's D2H under frame i+1 — and the frames equal render_path's bit for bit (3 poses;
    R and V return structures).  Precedent: RegNeRF/internal/models.py:311-322; reference R:140-178."""
    import subprocess
    import sys
    script = tmp_path / "sharded_render_worker.py"
    script.write_text(SHARDED_RENDER_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29595", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    env.pop("CNERF_FORCE_DIST", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29595", str(script), root], capture_output=True, text=True, env=env,
                       timeout=900)
    print(r.stdout[-1500:])
    assert r.returncode == 0 and "SHARDED_RENDER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_patch_sampler_golden(dev):
    """f-2 (V:1472-1517): patch pixels first (row index fastest), then the random pixels; rays / colours / per-pixel
    maps gathered at those coordinates — against the reference's own statements (fixture `patch`)."""
    from consistentnerf_amd import raybank as RB
    g = golden("patch")
    Hh, Ww = (int(v) for v in g["hw"])
    K = I.intrinsics(Hh, Ww, 30.0)
    pose = I.camera_pose(20.0, -15.0, 3.0)
    ro, rd = O.get_rays_np(Hh, Ww, K, pose[:3, :4])
    for tag in ("full", "crop"):
        pre = tuple(int(v) for v in g["crop_dhw"]) if tag == "crop" else None
        np.random.seed(21)
        starts = RB.draw_patch_starts(Hh, Ww, 4, 16, pre)
        assert np.array_equal(RB.patch_coords(starts, 16).numpy(), g[f"{tag}_patch_idxs"])
        depth_map = np.arange(Hh * Ww, dtype=np.float32).reshape(Hh, Ww)
        rays, tgt, sel, (dsel,) = RB.sample_patch_rays(T(g["image"], dev), pose, Hh, Ww, K, 37, starts,
                                                       select_inds=g[f"{tag}_select_inds"],
                                                       precrop_frac=0.9 if pre else None, extras=(depth_map,))
        sc = g[f"{tag}_select_coords"]
        assert np.array_equal(sel.cpu().numpy(), sc) and rays.shape == (2, 1024 + 37, 3)
        assert np.array_equal(tgt.cpu().numpy(), g["image"][sc[:, 0], sc[:, 1]])
        assert np.array_equal(dsel.cpu().numpy(), depth_map[sc[:, 0], sc[:, 1]])
        check(rays[0], ro[sc[:, 0], sc[:, 1]], 0.0, "rays_o"); check(rays[1], rd[sc[:, 0], sc[:, 1]], 1e-6, "rays_d")


def test_patch_depth_term_golden(dev):
    """f-5 (V:1681-1719): value and gradient of the monocular-depth patch term vs the reference's own statements run
    under autograd — ties in the patch extrema, non-positive / NaN depths, a patch without any valid prior pixel.
    Tolerance: 1e-5 relative on the loss, 1e-4 of the patch's largest |gradient| per element (fp32 normalisation
    by a range the 1/1e-4 outliers of case b inflate to 1e4)."""
    from consistentnerf_amd import ops, run_nerf_view as V
    g = golden("patch")
    for tag in ("a", "b"):
        dp, mono = T(g[f"term_{tag}_depth"], dev), T(g[f"term_{tag}_mono"], dev)
        loss, d = ops.patch_depth_loss(dp, mono, 4, 256)
        assert abs(loss.item() - float(g[f"term_{tag}_loss"])) <= 1e-5 * float(g[f"term_{tag}_loss"])
        ref = g[f"term_{tag}_grad"]
        got = d.cpu().numpy()
        assert np.array_equal(np.isnan(got), np.isnan(ref[:1024]))
        for p in range(4):
            sl = slice(256 * p, 256 * (p + 1))
            scale = np.nanmax(np.abs(ref[sl]))
            err = np.nanmax(np.abs(got[sl] - ref[sl]))
            assert err <= 1e-4 * scale + 1e-30, (tag, p, err, scale)
        # the autograd surface: gradient reaches all 1088 rays' depth (zeros past the patches), scaled by the upstream
        dpr = dp.clone().requires_grad_(True)
        (3.0 * V.midas_patch_loss(dpr, mono, 4, 16)).backward()
        gg = dpr.grad.cpu().numpy()
        assert gg.shape == (1088,) and not gg[1024:].any()
        m = ~np.isnan(ref[:1024])
        assert np.allclose(gg[:1024][m], 3.0 * got[m], rtol=1e-6, atol=0)
    # a second launch gives the same bits (fixed reduction order)
    l2, d2 = ops.patch_depth_loss(dp, mono, 4, 256)
    assert torch.equal(l2, loss) and np.array_equal(d2.cpu().numpy(), got, equal_nan=True)


@pytest.mark.parametrize("N,thr0,ndc", [(1280, 0.1, False), (4096, 1e-6, False), (5000, 0.05, True), (1, 0.1, False)])
def test_ss_ref_rays_one_launch_equals_the_lines(dev, N, thr0, ndc):
    """cnerf_ss_ref_rays (VT:905-925 as ONE launch) against the same block composed launch by launch the way round 4 ran it —
    `get_ref_rays(variant="VT")` (warp kernel + boolean indexing + gathers), |z - D_ref|, the doubling rule on a ladder, the double
    boolean selection via cumsum — on a random batch: masks, compaction order, gathered colours / depths, threshold bit for bit;
    ray directions 1e-6 (the lines' `directions @ c2w.T` is a rocBLAS GEMM); the packed rows == cnerf_pack_rays of the rays."""
    from consistentnerf_amd import ops, run_nerf_view as V
    from consistentnerf_amd.run_nerf_helpers import ndc_coefficients
    g = golden("warp")
    Hh, Ww = g["images"][1].shape[:2]
    K = g["K"]
    rs = np.random.RandomState(N)
    ro = T(np.tile(g["poses"][0][:3, 3], (N, 1)) + rs.normal(0, 0.05, (N, 3)).astype(np.float32), dev)
    rd = T(rs.normal(0, 1, (N, 3)).astype(np.float32) * 0.3 + g["poses"][0][:3, :3] @ np.array([0, 0, -1], np.float32), dev)
    dpt = T(rs.uniform(1.0, 6.0, N).astype(np.float32), dev)
    c2w = np.eye(4, dtype=np.float32); c2w[:3, :4] = g["poses"][1]
    w2c = torch.inverse(torch.from_numpy(c2w)).numpy()
    img, dep = T(g["images"][1], dev), T(g["depths"][1], dev)
    near, far = 0.5, 7.0
    coef = ndc_coefficients(Hh, Ww, K[0][0]) if ndc else (0., 0.)
    o = ops.ss_ref_rays(ro, rd, dpt, w2c, c2w, K, Hh, Ww, img, dep, thr0, near, far, True, ndc, coef)
    # the lines
    Pw = ro + dpt[:, None] * rd
    rgb_l, d_l, Xc_l, ro_l, rd_l, mb_l = V.get_ref_rays(T(w2c, dev)[None], T(c2w, dev)[None], T(K, dev)[None], Pw[None, :, None, :],
                                                        img.permute(2, 0, 1)[None], dep[None], variant="VT")
    M = int(mb_l.sum().item())
    assert o["M"] == M and np.array_equal(o["inb"].cpu().numpy().astype(bool), mb_l.reshape(-1).cpu().numpy())
    if M == 0:
        return
    assert torch.equal(o["target"], rgb_l[0].t().contiguous()) and torch.equal(o["depth_tgt"], d_l.reshape(-1))
    adiff = (Xc_l[..., -1].reshape(-1) - d_l.reshape(-1)).abs()
    assert torch.equal(o["depth_diff"], adiff)
    cand = torch.full((64,), thr0, dtype=torch.float32) * torch.pow(2.0, torch.arange(64, dtype=torch.float32))
    k = int((adiff.min().cpu() >= cand).sum().clamp(max=63))
    assert o["k"] == k and np.float32(o["thr"]) == cand[k].numpy()
    mk = adiff < cand[k].to(dev)
    assert mk.any() and torch.equal(o["mask"].bool(), mk)
    pos = (torch.cumsum(mb_l.reshape(-1).long(), 0) - 1).clamp_(min=0)
    assert torch.equal(o["sel"], (mb_l.reshape(-1) & mk[pos]).float())
    assert torch.equal(o["rank"].long()[mb_l.reshape(-1)], torch.arange(M, device=dev))
    assert torch.equal(o["rays_od"][0], ro_l)
    err = (o["rays_od"][1] - rd_l).abs().max().item()
    assert err <= 1e-6 * rd_l.abs().max().item(), err
    rows = ops.pack_rays(o["rays_od"][0].contiguous(), o["rays_od"][1].contiguous(), near, far, True, ndc, coef)
    assert torch.equal(o["rows"], rows)


def _ss_random_batch(g, N, seed, thr0, dev):
    """A random batch of N rays around view 0 of the `warp` fixture's scene with depth priors in [1, 6], WITHOUT the rows whose warp is
    numerically ambiguous between two fp32 evaluations of the same formulas (the kernel's scalar fmas-off arithmetic vs ATen's
    matmul on the CPU): projected pixel within 1e-3 px of a rounding tie or of the image border, |z - D_ref| within 1e-4 relative of
    the threshold the oracle ends up with (or of half of it: the previous rung of the doubling ladder).  Removing a row can change
    the threshold, so the filter is iterated to a fixed point.  -> (ro, rd, depth) numpy, the oracle's block on them."""
    Hh, Ww = g["images"][1].shape[:2]
    K = torch.from_numpy(g["K"])
    rs = np.random.RandomState(seed)
    ro = (np.tile(g["poses"][0][:3, 3], (N, 1)) + rs.normal(0, 0.05, (N, 3))).astype(np.float32)
    rd = (rs.normal(0, 1, (N, 3)) * 0.3 + g["poses"][0][:3, :3] @ np.array([0, 0, -1.0])).astype(np.float32)
    dpt = rs.uniform(1.0, 6.0, N).astype(np.float32)
    c2w = torch.eye(4); c2w[:3, :4] = torch.from_numpy(g["poses"][1])
    w2c = torch.inverse(c2w)
    img, dep = torch.from_numpy(g["images"][1]).permute(2, 0, 1), torch.from_numpy(g["depths"][1])
    for _ in range(6):
        P = T(ro) + T(dpt)[:, None] * T(rd)
        Xc = P.double() @ w2c[:3, :3].double().t() + w2c[:3, 3].double()
        pix = Xc @ K.double().t()
        px, py = pix[:, 0] / pix[:, 2], pix[:, 1] / pix[:, 2]
        frac = lambda v: (v - torch.floor(v) - 0.5).abs()    # noqa: E731  (distance to a .5 tie)
        edge = torch.minimum(torch.minimum((px - 0.5).abs(), (px - (Ww - 1.5)).abs()),
                             torch.minimum((py - 0.5).abs(), (py - (Hh - 1.5)).abs()))
        bad = (frac(px) < 1e-3) | (frac(py) < 1e-3) | (edge < 1e-3)
        keep = ~bad.numpy()
        ro, rd, dpt = ro[keep], rd[keep], dpt[keep]
        blk = O.ss_block(T(ro), T(rd), T(dpt), torch.from_numpy(g["poses"][1]), K, img, dep, thr0)
        inb = blk["mask_bound"].reshape(-1)
        diff = torch.zeros(inb.shape[0])
        Pk = T(ro) + T(dpt)[:, None] * T(rd)
        zc = (Pk @ w2c[:3, :3].t() + w2c[:3, 3])[:, 2]
        diff[inb] = (zc[inb] - blk["rays_depth_ref"].reshape(-1)).abs()
        thr = blk["thr"]
        amb = inb & (((diff - thr).abs() < 1e-4 * thr) | ((diff - thr / 2).abs() < 1e-4 * thr))
        if not bool(amb.any()) and bool(keep.all()):
            return ro, rd, dpt, blk
        k2 = ~amb.numpy()
        ro, rd, dpt = ro[k2], rd[k2], dpt[k2]
    raise AssertionError("the tie filter did not reach a fixed point")


@pytest.mark.parametrize("N,thr0,ndc", [(4096, 0.1, False), (5000, 1e-6, False), (4096, 0.02, True), (5000, 3e-4, False)])
def test_ss_ref_rays_vs_oracle(dev, N, thr0, ndc):
    """ops.ss_ref_rays AND ops.ss_batch (VT:905-925 as one launch) against **O.ss_block** — the oracle's block,
    pinned on the reference's own fixture in the CPU suite (tests/test_oracle_golden.py::test_ss_block_golden) — on random 4096- and
    5000-ray batches, incl. thresholds that need many doublings and the NDC row packing: mask_bound, the compaction order, the
    occlusion mask, `sel`, the gathered colours / depth priors and the threshold bit for bit; the reference rays 1e-6; the packed
    rows against the oracle's build_ray_batch (R:100-125) 1e-6; ss_batch's combined batch (primary rows, live count, masks,
    padding) against the same."""
    from consistentnerf_amd import ops
    from consistentnerf_amd.run_nerf_helpers import ndc_coefficients
    g = golden("warp")
    Hh, Ww = g["images"][1].shape[:2]
    K = g["K"]
    ro, rd, dpt, blk = _ss_random_batch(g, N, N + int(1e6 * thr0), thr0, dev)
    n = ro.shape[0]
    assert n >= N - 64, "the tie filter removed an implausible number of rows"
    if thr0 < 1e-3:
        assert blk["thr"] > thr0, "this case is meant to need threshold doublings"
    c2w = np.eye(4, dtype=np.float32); c2w[:3, :4] = g["poses"][1]
    w2c = torch.inverse(torch.from_numpy(c2w)).numpy()
    img, dep = T(g["images"][1], dev), T(g["depths"][1], dev)
    near, far = 0.5, 7.0
    coef = ndc_coefficients(Hh, Ww, K[0][0]) if ndc else (0., 0.)
    M = int(blk["mask_bound"].sum())
    rows_ref = O.build_ray_batch(blk["rays_ref"][0], blk["rays_ref"][1], near, far, True, ndc, Hh, Ww, float(K[0][0]))
    want_mask = blk["mask"].reshape(-1).numpy()

    def check_block(o, mask_u8, Mk, thr_f, k):
        assert Mk == M and np.array_equal(o["inb"].cpu().numpy().astype(bool), blk["mask_bound"].reshape(-1).numpy())
        assert np.float32(thr_f) == np.float32(blk["thr"]), (thr_f, blk["thr"])
        assert np.float32(thr0) * np.float32(2.0) ** k == np.float32(blk["thr"])
        assert np.array_equal(mask_u8[:M].cpu().numpy().astype(bool), want_mask), "occlusion mask"
        assert np.array_equal(o["sel"].cpu().numpy(), blk["sel"].numpy()), "sel"
        rank = o["rank"].cpu().numpy()
        inb = blk["mask_bound"].reshape(-1).numpy()
        assert np.array_equal(rank[inb], np.arange(M)) and (rank[~inb] == -1).all(), "compaction order"
    o = ops.ss_ref_rays(T(ro, dev), T(rd, dev), T(dpt, dev), w2c, c2w, K, Hh, Ww, img, dep, thr0, near, far, True, ndc, coef)
    check_block(o, o["mask"], o["M"], o["thr"], o["k"])
    assert np.array_equal(o["target"].cpu().numpy(), blk["rgb_target_ref"][0].t().numpy())
    assert np.array_equal(o["depth_tgt"].cpu().numpy(), blk["rays_depth_ref"].reshape(-1).numpy())
    check(o["rays_od"], blk["rays_ref"], 1e-6, "rays_ref")
    check(o["rows"], rows_ref, 2e-6 if ndc else 1e-6, "rows")
    # the combined batch of the one-render step
    tgt_s = torch.rand(n, 3, generator=torch.Generator().manual_seed(3))
    b = ops.ss_batch(T(ro, dev), T(rd, dev), T(dpt, dev), tgt_s.to(dev), w2c, c2w, K, Hh, Ww, img, dep, thr0, near, far, True, ndc, coef)
    meta = b["meta"].cpu().numpy()
    thr_b = float(np.int32(meta[2]).view(np.float32))
    check_block(b, b["occ"], int(meta[0]), thr_b, int(meta[1]))
    assert int(b["live"].item()) == n + M and int(meta[5]) == int(blk["sel"].sum())
    amin = float(np.int32(meta[4]).view(np.float32))
    assert amin < blk["thr"] and (blk["thr"] == thr0 or amin >= blk["thr"] / 2), "the local minimum of |z - D_ref| (meta[4])"
    rows, tg, pr, mk = (b[k].cpu() for k in ("rows", "target", "prior", "mask"))
    rows_p = O.build_ray_batch(T(ro), T(rd), near, far, True, ndc, Hh, Ww, float(K[0][0]))
    check(rows[:n], rows_p, 2e-6 if ndc else 1e-6, "primary rows")
    assert torch.equal(rows[n:n + M], o["rows"].cpu()), "reference rows of the combined batch"
    assert torch.equal(tg[:n], tgt_s) and torch.equal(tg[n:n + M], o["target"].cpu()) and not tg[n + M:].any()
    assert torch.equal(pr[:n], T(dpt)) and torch.equal(pr[n:n + M], o["depth_tgt"].cpu()) and not pr[n + M:].any()
    assert torch.equal(mk[:n], blk["sel"]) and bool((mk[n:n + M] == 1).all()) and not mk[n + M:].any()
    pad = rows[n + M:]
    assert pad.shape[0] == n - M and bool(torch.isfinite(pad).all()) and bool((pad[:, 6] == near).all()) and bool((pad[:, 7] == far).all())
    assert bool((pad[:, 3:6].abs().sum(1) > 0).all()), "padding rows must be valid rays"


@pytest.mark.parametrize("coins,with_depth", [((1, 1, 1, 1), True), ((0, 0, 0, 0), True), ((1, 0, 0, 1), True), ((0, 1, 1, 0), True),
                                              ((1, 1, 0, 0), True), ((1, 0, 1, 0), False), ((0, 0, 0, 0), False)])
def test_ss_step_loss_one_call_equals_the_lines(dev, coins, with_depth):
    """run_nerf_view.ss_step_loss (VT:899-969 as one call: the primary render's four coin-gated terms folded into its compositing
    launches through cnerf_closs_finish_ss, the second render's folded as before) against the same step written as the
    reference's lines — render, ss_consistency, ss_primary_losses (masked-loss / img2mse launches + autograd) — on the `ssloss`
    fixture's scene and networks: loss 2e-6 relative, every parameter gradient 2e-6 of the tensor's largest (the folded form sums
    per-workgroup fp64 partials and merges the coarse colour term's fallback into one seed weight: summation order only), incl.
    the reference's fallback of the COARSE colour term to the FINE rgb (coin 0, VT:959)."""
    from consistentnerf_amd import run_nerf_view as V
    g = golden("ssloss")
    Hh, Ww, far = 32, 40, 7.0
    K, poses = g["K"], g["poses"]
    ro, rd = O.get_rays_np(Hh, Ww, K, poses[0][:3, :4])
    coarse, _ = make_model(4, 128, True, 5, 31, dev)
    fine, _ = make_model(4, 128, True, 5, 32, dev)
    kw = _kwargs(coarse, fine, 16, 16, 0.0, False, 0.0, False)
    kw.update(near=2.0, far=far, ndc=False, use_viewdirs=True)
    sel, r = g["a.sel"], int(g["a.ref_index"])
    rays = torch.stack([T(ro.reshape(-1, 3)[sel], dev), T(rd.reshape(-1, 3)[sel], dev)], 0)
    tgt = T(g["images"][0].reshape(-1, 3)[sel].astype(np.float32), dev)
    prior = T(g["depths"][0].reshape(-1)[sel], dev)
    params = [p for m in (coarse, fine) for p in m.parameters()]

    def grads():
        out = [None if p.grad is None else p.grad.detach().clone() for p in params]
        for p in params:
            p.grad = None
        return out

    rgb, disp, acc, depth, extras = V.render(Hh, Ww, K, chunk=4096, rays=rays, retraw=True, **kw)
    ss = V.ss_consistency(rays[0], rays[1], prior, poses[r], K, g["images"][r], g["depths"][r], Hh, Ww, kw, chunk=4096,
                          occlusion_threshold=0.1, with_depth_loss=with_depth)
    lines_coins = list(coins) if with_depth else [coins[0], coins[2]]
    lp, il, il0 = V.ss_primary_losses(rgb, depth, extras, tgt, prior, ss["mask_bound"], ss["mask"], with_depth_loss=with_depth,
                                      coins=lines_coins)
    loss_l = ss["loss"] + lp
    loss_l.backward()
    g_l = grads()
    loss_o, info = V.ss_step_loss(Hh, Ww, K, rays, tgt, prior, poses[r], g["images"][r], g["depths"][r], kw, chunk=4096,
                                  occlusion_threshold=0.1, with_depth_loss=with_depth, coins=coins, route="two_renders")
    loss_o.backward()
    g_o = grads()
    assert torch.equal(info["mask"], ss["mask"]) and torch.equal(info["sel"], ss["sel"])
    assert torch.equal(info["rgb"], rgb) and torch.equal(info["depth_pred"], depth) and torch.equal(info["rgb_ref"], ss["rgb_ref"])
    ll, lo = loss_l.item(), loss_o.item()
    print(f"  coins {coins}: loss lines {ll:.8f} one call {lo:.8f}; img_loss {il.item():.8f} / {info['img_loss'].item():.8f}")
    assert abs(lo - ll) <= 2e-6 * abs(ll)
    assert abs(info["img_loss"].item() - il.item()) <= 2e-6 * abs(il.item())
    assert abs(info["img_loss0"].item() - il0.item()) <= 2e-6 * abs(il0.item())
    assert abs(info["loss_ref"].item() - ss["loss"].item()) <= 1e-7 * abs(ss["loss"].item())
    worst = 0.0
    for a, b in zip(g_l, g_o):
        assert (a is None) == (b is None)
        if a is not None:
            worst = max(worst, (a - b).abs().max().item() / max(a.abs().max().item(), 1e-30))
    print(f"  worst relative gradient difference {worst:.2e}")
    assert worst <= 2e-6


def _ss_scene(dev, N, seed=0, D=4, W=128, Nc=64, Nf=128, perturb=0.0, owned=False, coarse_arch=None):
    """The `ssloss` fixture's 3-view scene with N rays drawn (with replacement) from view 0, two seeded networks and render kwargs at
    sample counts the live-row gate applies to (multiples of 32)."""
    g = golden("ssloss")
    Hh, Ww, far = 32, 40, 7.0
    K, poses = g["K"], g["poses"]
    ro, rd = O.get_rays_np(Hh, Ww, K, poses[0][:3, :4])
    rs = np.random.RandomState(seed)
    pix = rs.randint(0, Hh * Ww, N)
    coarse, _ = make_model(*(coarse_arch or (D, W)), True, 5, 31 + seed, dev)
    fine, _ = make_model(D, W, True, 5, 32 + seed, dev)
    kw = _kwargs(coarse, fine, Nc, Nf, perturb, False, 0.0, False)
    kw.update(near=2.0, far=far, ndc=False, use_viewdirs=True)
    rays = torch.stack([T(ro.reshape(-1, 3)[pix], dev), T(rd.reshape(-1, 3)[pix], dev)], 0)
    tgt = T(g["images"][0].reshape(-1, 3)[pix].astype(np.float32), dev)
    prior = T(g["depths"][0].reshape(-1)[pix], dev)
    opt = None
    if owned:
        from consistentnerf_amd.optim import FusedAdam
        opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
    return dict(H=Hh, W=Ww, K=K, poses=poses, g=g, coarse=coarse, fine=fine, kw=kw, rays=rays, tgt=tgt, prior=prior, opt=opt)


def _grads_of(params):
    out = [None if p.grad is None else p.grad.detach().clone() for p in params]
    for p in params:
        p.grad = None
    return out


@pytest.mark.parametrize("coins,with_depth,thr", [((1, 1, 1, 1), True, 0.1), ((0, 0, 0, 0), True, 0.1), ((1, 0, 0, 1), True, 1e-4),
                                                  ((0, 1, 1, 0), True, 0.1), ((1, 0, 1, 0), False, 0.1), ((0, 0, 0, 0), False, 1e-4)])
def test_ss_step_one_render_equals_two_renders(dev, coins, with_depth, thr):
    """run_nerf_view.ss_step_loss(route="one_render") — primary and warped rays as ONE batch of 2N rows built by
    the warp launch (ops.ss_batch), the MLP kernels stopping at the DEVICE-side live-row count, the two-segment loss tail
    (cnerf_closs_finish_ss2), no host synchronisation — against round 5's two-render form (itself equal to the reference's lines:
    test_ss_step_loss_one_call_equals_the_lines): rays are independent, so every map of every live row is bit-identical; the loss
    equals to summation order (2e-6); every parameter gradient 3e-6 of the tensor's largest; the padding rows contribute nothing
    (zero raw outputs, zero weight); the device-side count equals the two-render route's read-back."""
    from consistentnerf_amd import run_nerf_view as V
    sc = _ss_scene(dev, 1024)
    H, W, K, kw, rays, tgt, prior, g = sc["H"], sc["W"], sc["K"], sc["kw"], sc["rays"], sc["tgt"], sc["prior"], sc["g"]
    r = 1
    params = [p for m in (sc["coarse"], sc["fine"]) for p in m.parameters()]
    args = (H, W, K, rays, tgt, prior, sc["poses"][r], g["images"][r], g["depths"][r], kw)
    l2, i2 = V.ss_step_loss(*args, chunk=4096, occlusion_threshold=thr, with_depth_loss=with_depth, coins=coins, route="two_renders")
    l2.backward()
    g2 = _grads_of(params)
    l1, i1 = V.ss_step_loss(*args, chunk=4096, occlusion_threshold=thr, with_depth_loss=with_depth, coins=coins, route="one_render")
    assert i1["route"] == "one_render" and i2["route"] == "two_renders"
    l1.backward()
    g1 = _grads_of(params)
    h = V.ss_host_view(i1)
    M = h["M"]
    assert M == i2["batch_rays_ref"].shape[1] and 0 < M < 1024 and int(i1["live"].item()) == 1024 + M
    assert float(h["threshold"]) == float(i2["threshold"]) and torch.equal(h["mask"], i2["mask"]) and torch.equal(i1["sel"], i2["sel"])
    assert torch.equal(h["mask_bound"], i2["mask_bound"]) and torch.equal(h["batch_rays_ref"], i2["batch_rays_ref"])
    assert torch.equal(h["rgb_target_ref"], i2["rgb_target_ref"]) and torch.equal(h["rays_depth_ref"], i2["rays_depth_ref"])
    for k in ("rgb", "depth_pred", "acc"):
        assert torch.equal(i1[k], i2[k]), k
    assert torch.equal(h["rgb_ref"], i2["rgb_ref"]) and torch.equal(h["depth_pred_ref"], i2["depth_pred_ref"])
    for k in ("rgb0", "depth0", "raw"):
        assert torch.equal(i1["extras"][k], i2["extras"][k]) and torch.equal(h["extras_ref"][k], i2["extras_ref"][k]), k
    # padding rows: zero raw outputs (the gated tiles), nothing else depends on them.  (The opt-in bf16x3 training forward has no
    # gate: it evaluates the padding rays like any other — valid rays of weight 0 — so its padding outputs are not zero.)
    assert BF3 or not i1["extras_ref"]["raw"][M:].any()
    a, b = l1.item(), l2.item()
    print(f"  coins {coins} thr {thr}: M={M} loss one render {a:.8f} two renders {b:.8f}")
    assert abs(a - b) <= 2e-6 * abs(b)
    t = {k: float(v) for k, v in i1["terms"].items()}
    assert t["M"] == M and abs(t["img_loss"] - float(i2["img_loss"])) <= 2e-6 * abs(float(i2["img_loss"]))
    assert abs(t["img_loss0"] - float(i2["img_loss0"])) <= 2e-6 * abs(float(i2["img_loss0"]))
    ref_sum = t["img_loss_ref"] + t["depth_loss_ref"] + t["img_loss0_ref"] + t["depth_loss0_ref"]
    assert abs(ref_sum - i2["loss_ref"].item()) <= 2e-6 * abs(i2["loss_ref"].item())
    worst = 0.0
    for x, y in zip(g1, g2):
        assert (x is None) == (y is None)
        if x is not None:
            worst = max(worst, (x - y).abs().max().item() / max(y.abs().max().item(), 1e-30))
    print(f"  worst relative gradient difference {worst:.2e}")
    assert worst <= 3e-6


@pytest.mark.parametrize("case", ["single_level", "single_level_no_depth", "ndc", "too_big_for_one_chunk", "n_not_a_multiple_of_8"])
def test_ss_step_one_render_other_shapes(dev, case):
    """The one-render route off its main shape, each against the two-render route (loss 2e-6, gradients 3e-6, maps of the live rows
    bit-identical): a render WITHOUT a fine network (one level: cnerf_closs_finish_ss2 with one workspace, cnerf_mlp_bwd_live on the
    one network), the same without the depth terms, NDC rays (the rows of both segments go through the NDC warp in the batch
    assembly); and batches that do not qualify — 2N rows that do not fit one chunk, N not a multiple of 8: route=None falls back to
    two renders, route="one_render" raises."""
    from consistentnerf_amd import ops, run_nerf_view as V
    sc = _ss_scene(dev, 512, seed=3)
    H, W, K, kw, rays, tgt, prior, g = sc["H"], sc["W"], sc["K"], dict(sc["kw"]), sc["rays"], sc["tgt"], sc["prior"], sc["g"]
    nets = [sc["coarse"], sc["fine"]]
    with_depth, coins, chunk = True, (1, 1, 1, 0), 4096
    if case.startswith("single_level"):
        kw.update(network_fine=None, N_importance=0)
        nets = [sc["coarse"]]
        with_depth, coins = case == "single_level", (1, 1, 0, 0)
    elif case == "ndc":
        kw.update(ndc=True, near=0.0, far=1.0)
        # forward-facing rays for the NDC warp (d_z < 0, origins behind the near plane at z = -1)
        rays = torch.stack([rays[0] * 0.1 + torch.tensor([0.0, 0.0, 0.5], device=dev), torch.cat([rays[1][:, :2] * 0.3, -torch.ones(512, 1, device=dev)], 1)], 0)
        prior = prior * 0.2
    elif case == "too_big_for_one_chunk":
        chunk = 768
    params = [p for m in nets for p in m.parameters()]
    args = (H, W, K, rays, tgt, prior, sc["poses"][1], g["images"][1], g["depths"][1], kw)
    if case == "n_not_a_multiple_of_8":      # (a compositing workgroup's loss partial must belong to one segment)
        rays, tgt, prior = rays[:, :509].contiguous(), tgt[:509], prior[:509]
        args = (H, W, K, rays, tgt, prior, sc["poses"][1], g["images"][1], g["depths"][1], kw)
    if case in ("too_big_for_one_chunk", "n_not_a_multiple_of_8"):
        with pytest.raises(ops.CnerfError):
            V.ss_step_loss(*args, chunk=chunk, with_depth_loss=True, coins=coins, route="one_render")
        _, info = V.ss_step_loss(*args, chunk=chunk, with_depth_loss=True, coins=coins)
        assert info["route"] == "two_renders"
        return
    l2, i2 = V.ss_step_loss(*args, chunk=chunk, occlusion_threshold=0.1, with_depth_loss=with_depth, coins=coins, route="two_renders")
    l2.backward()
    g2 = _grads_of(params)
    l1, i1 = V.ss_step_loss(*args, chunk=chunk, occlusion_threshold=0.1, with_depth_loss=with_depth, coins=coins)
    assert i1["route"] == "one_render"
    l1.backward()
    g1 = _grads_of(params)
    h = V.ss_host_view(i1)
    M = h["M"]
    assert M == i2["batch_rays_ref"].shape[1] and M > 0
    assert torch.equal(i1["rgb"], i2["rgb"]) and torch.equal(h["rgb_ref"], i2["rgb_ref"]) and torch.equal(h["depth_pred_ref"], i2["depth_pred_ref"])
    assert torch.equal(h["mask"], i2["mask"]) and torch.equal(i1["sel"], i2["sel"])
    if case == "ndc":
        assert torch.equal(i1["rows"][512:512 + M], i2["batch_rays_ref"]._cnerf_packed.rows), "NDC rows of the second segment"
    print(f"  {case}: M={M} loss one render {l1.item():.8f} two renders {l2.item():.8f}")
    assert abs(l1.item() - l2.item()) <= 2e-6 * abs(l2.item())
    worst = 0.0
    for x, y in zip(g1, g2):
        assert (x is None) == (y is None)
        if x is not None and float(y.abs().max()) > 0:
            worst = max(worst, (x - y).abs().max().item() / y.abs().max().item())
    print(f"  worst relative gradient difference {worst:.2e}")
    assert worst <= 3e-6


def test_ss_step_one_render_with_no_ray_in_view(dev):
    """M == 0: no point of the batch projects into the reference view (here: every depth-prior point sits in the reference camera's
    z = 0 plane).  The reference's `while mask.sum() == 0` (VT:921) never ends there and the two-render route raises; the one-render
    route cannot know on the host — the tail kernel drops the second render's terms (0 instead of 0 / 0), nothing is selected, so the
    coin-gated primary terms take their un-masked branch: loss == 2 x img2mse(rgb, target) (VT:942 + the fallback of VT:959), finite,
    terms['M'] == 0 tells the caller, and the gradient is that loss's."""
    from consistentnerf_amd import ops, run_nerf as R, run_nerf_view as V
    sc = _ss_scene(dev, 256, seed=6)
    H, W, K, kw, tgt, g = sc["H"], sc["W"], sc["K"], sc["kw"], sc["tgt"], sc["g"]
    pose = sc["poses"][1]
    ro = T(np.tile(pose[:3, 3], (256, 1)).astype(np.float32), dev)
    rd = T(np.tile(pose[:3, 0], (256, 1)).astype(np.float32), dev) * torch.linspace(0.5, 2.0, 256, device=dev)[:, None]
    rays, prior = torch.stack([ro, rd], 0), torch.ones(256, device=dev)
    params = [p for m in (sc["coarse"], sc["fine"]) for p in m.parameters()]
    args = (H, W, K, rays, tgt, prior, pose, g["images"][1], g["depths"][1], kw)
    with pytest.raises(ops.CnerfError):
        V.ss_step_loss(*args, chunk=4096, with_depth_loss=True, coins=(1, 1, 1, 1), route="two_renders")
    loss, info = V.ss_step_loss(*args, chunk=4096, with_depth_loss=True, coins=(1, 1, 1, 1), route="one_render")
    loss.backward()
    g1 = _grads_of(params)
    t = {k: float(v) for k, v in info["terms"].items()}
    assert t["M"] == 0.0 and int(info["live"].item()) == 256 and not info["sel"].any()
    assert t["img_loss_ref"] == t["depth_loss_ref"] == t["img_loss0_ref"] == t["depth_loss0_ref"] == 0.0 and t["depth_loss"] == 0.0
    rgb, _, _, _, _ = V.render(H, W, K, chunk=4096, rays=rays, retraw=True, **kw)
    want = 2.0 * R.img2mse(rgb, tgt)
    want.backward()
    g0 = _grads_of(params)
    assert np.isfinite(loss.item()) and abs(loss.item() - want.item()) <= 2e-6 * abs(want.item()), (loss.item(), want.item())
    for x, y in zip(g1, g0):
        if y is None or float(y.abs().max()) == 0:
            assert x is None or float(x.abs().max()) == 0
        else:
            assert float((x - y).abs().max()) <= 3e-6 * float(y.abs().max())


@fp32_only
@pytest.mark.parametrize("coins,coarse_arch", [((0, 1, 0, 0), None), ((1, 1, 1, 1), None), ((1, 0, 0, 0), None), ((0, 1, 0, 0), (2, 64)),
                                               ((1, 1, 0, 1), (2, 64))])
def test_ss_step_one_render_merged_backward_with_skip(dev, coins, coarse_arch):
    """The production route of the one-render step: both networks owned by FusedAdam -> ONE dgrad grid + ONE wgrad grid for both levels
    (cnerf_mlp_dgrad_pair_live / cnerf_mlp_wgrad_pair_live: tiles / re-cut point ranges stop at the device-side live-row count),
    accumulating straight into the flat gradient.  With both coarse coins 0 (VT:959, VT:966) the primary rays' coarse level is left
    out of the backward (first_ray = N) exactly as the two-render form leaves the whole coarse backward of the primary render out.
    Flat gradient vs the two-render route 6e-6 of each tensor's largest; the launch's arguments checked."""
    from consistentnerf_amd import ops, run_nerf as R, run_nerf_view as V
    seen = []
    orig = ops.mlp_backward_pair

    def spy(*a, **k):
        seen.append((a[3], a[10], k.get("live") is not None, k.get("first0", 0), k.get("first1", 0)))
        return orig(*a, **k)
    grads = {}
    for route in ("two_renders", "one_render"):
        sc = _ss_scene(dev, 1024, seed=4, owned=True, coarse_arch=coarse_arch)     # (coarse_arch: two architectures -> the merged
        args = (sc["H"], sc["W"], sc["K"], sc["rays"], sc["tgt"], sc["prior"], sc["poses"][1], sc["g"]["images"][1], sc["g"]["depths"][1],
                sc["kw"])                                                              #  backward is two dgrad launches + one wgrad grid)
        ops.mlp_backward_pair = spy
        try:
            loss, info = V.ss_step_loss(*args, chunk=4096, occlusion_threshold=0.1, with_depth_loss=True, coins=coins, route=route)
            sc["opt"].zero_grad()
            R.backward(loss)
        finally:
            ops.mlp_backward_pair = orig
        sc["opt"].materialize_grad()
        grads[route] = (sc["opt"].flat_grad.clone(), loss.item(), sc["opt"])
    skip = not (coins[2] or coins[3])
    one = [c for c in seen if c[2]]
    assert len(one) == 1 and one[0][0] == one[0][1] == 2048 and one[0][3] == 0 and one[0][4] == (1024 if skip else 0), seen
    (g2, l2, opt), (g1, l1, _) = grads["two_renders"], grads["one_render"]
    assert abs(l1 - l2) <= 2e-6 * abs(l2)
    worst = 0.0
    for p, o in zip(opt.params, opt._offsets):
        a, b = g1[o:o + p.numel()], g2[o:o + p.numel()]
        if float(b.abs().max()) == 0:
            assert float(a.abs().max()) == 0
            continue
        worst = max(worst, float((a - b).abs().max()) / float(b.abs().max()))
    print(f"  coins {coins} (coarse primary backward {'skipped' if skip else 'kept'}): worst relative gradient difference {worst:.2e}")
    assert worst <= 6e-6       # (measured 1e-6 ... 3.2e-6: the re-cut point ranges associate the ~10^5-term sums differently)


@pytest.mark.parametrize("nshards", [2, 3])
def test_ss_step_loss_sharded(dev, nshards):
    """VERDICT r05 missing 3 / SURVEY 8e: the `--ss_loss` step of a batch SHARDED over ranks equals the 1-rank step.  The two
    batch-global quantities of VT:917-966 — the minimum |z - D_ref| that drives the threshold doubling, and the ray counts the masked
    means divide by — are exchanged (run_nerf_view.ss_global_stats: all-reduce MIN, then SUM of three counts); here the shards run
    one after the other on the one GPU and the exchanges are done by hand: the per-shard losses ADD UP to the unsharded loss (1e-6),
    the summed gradients equal the unsharded gradient (3e-6), every shard applies the GLOBAL threshold (a threshold of 3e-4 that
    needs doublings, with shards whose own minimum differs)."""
    from consistentnerf_amd import ops, run_nerf_view as V
    N = 1536
    sc = _ss_scene(dev, N, seed=2)
    H, W, K, kw, rays, tgt, prior, g = sc["H"], sc["W"], sc["K"], sc["kw"], sc["rays"], sc["tgt"], sc["prior"], sc["g"]
    r, thr, coins = 2, 3e-4, (1, 1, 0, 1)
    # (rays are drawn with replacement from 1280 pixels: a per-ray offset on the priors makes every point — and the shards' minima — distinct)
    prior = prior + torch.linspace(0.0, 0.05, N, device=dev)
    params = [p for m in (sc["coarse"], sc["fine"]) for p in m.parameters()]
    full, info = V.ss_step_loss(H, W, K, rays, tgt, prior, sc["poses"][r], g["images"][r], g["depths"][r], kw, chunk=8192,
                                occlusion_threshold=thr, with_depth_loss=True, coins=coins, route="one_render")
    full.backward()
    g_full = _grads_of(params)
    hv = V.ss_host_view(info)
    assert float(hv["threshold"]) > thr, "the case is meant to need doublings"
    bounds = [(N * k // nshards) // 8 * 8 for k in range(nshards)] + [N]
    shards = [slice(bounds[k], bounds[k + 1]) for k in range(nshards)]
    # exchange 1: MIN of the shards' minimum |z - D_ref|  (what dist.all_reduce(op=MIN) does in ss_global_stats)
    c2w, w2c = V._ss_pose(sc["poses"][r])
    img, dep = T(g["images"][r], dev), T(g["depths"][r], dev)
    metas = []
    for sl in shards:
        b = ops.ss_batch(rays[0][sl], rays[1][sl], prior[sl], tgt[sl], w2c.numpy(), c2w.numpy(), K, H, W, img, dep, thr, 2.0, 7.0, True,
                         False)
        metas.append(b["meta"])
    amins = torch.stack([m[4:5].view(torch.float32) for m in metas])
    assert float(amins.max()) > float(amins.min()), "the shards' own minima should differ"
    amin_g = amins.min(0).values
    # exchange 2: SUM of (selected, primary, warped) counts under the global threshold
    counts = torch.zeros(3, device=dev)
    for sl in shards:
        b = ops.ss_batch(rays[0][sl], rays[1][sl], prior[sl], tgt[sl], w2c.numpy(), c2w.numpy(), K, H, W, img, dep, thr, 2.0, 7.0, True,
                         False, amin_global=amin_g)
        m = b["meta"]
        assert float(np.int32(int(m[2])).view(np.float32)) == float(hv["threshold"]), "every shard applies the GLOBAL threshold"
        counts += torch.stack([m[5].float(), torch.tensor(float(sl.stop - sl.start), device=dev), m[0].float()])
    assert int(counts[2]) == hv["M"] and int(counts[0]) == int(info["sel"].sum()) and int(counts[1]) == N
    total, g_sum = 0.0, None
    for sl in shards:
        rs_ = torch.stack([rays[0][sl], rays[1][sl]], 0)
        ls, _ = V.ss_step_loss(H, W, K, rs_, tgt[sl], prior[sl], sc["poses"][r], g["images"][r], g["depths"][r], kw, chunk=8192,
                               occlusion_threshold=thr, with_depth_loss=True, coins=coins, global_stats=(amin_g, counts))
        ls.backward()
        gs = _grads_of(params)
        total += ls.item()
        g_sum = gs if g_sum is None else [None if a is None else a + b_ for a, b_ in zip(g_sum, gs)]
    print(f"  {nshards} shards: sum of shard losses {total:.8f} vs unsharded {full.item():.8f}")
    assert abs(total - full.item()) <= 2e-6 * abs(full.item())
    worst = 0.0
    for x, y in zip(g_sum, g_full):
        assert (x is None) == (y is None)
        if x is not None:
            worst = max(worst, (x - y).abs().max().item() / max(y.abs().max().item(), 1e-30))
    print(f"  worst relative gradient difference {worst:.2e}")
    assert worst <= 3e-6


def test_graphed_ss_step_equals_eager(dev):
    """The one-render `--ss_loss` step has no host synchronisation, so graph.GraphedStep can record it — warp / batch assembly, both
    levels of the 2N-row render with the device-side row count, the two-segment loss, the merged backward, FusedAdam — and replay it
    on CHANGING batches (different rays, therefore a different M each step: the recording does not depend on it): weights after
    three replayed steps == the same three steps run eagerly, bit for bit."""
    from consistentnerf_amd import run_nerf as R, run_nerf_view as V
    from consistentnerf_amd.graph import GraphedStep
    coins = (1, 1, 0, 1)

    def build():
        sc = _ss_scene(dev, 512, seed=5, owned=True)
        r = 1
        img, dep = T(sc["g"]["images"][r], dev), T(sc["g"]["depths"][r], dev)

        def step(ro, rd, tgt, prior):
            loss, _ = V.ss_step_loss(sc["H"], sc["W"], sc["K"], torch.stack([ro, rd], 0), tgt, prior, sc["poses"][r], img, dep, sc["kw"],
                                     chunk=4096, occlusion_threshold=0.1, with_depth_loss=True, coins=coins, route="one_render")
            sc["opt"].zero_grad()
            R.backward(loss)
            sc["opt"].step()
            return loss
        return sc, step
    batches = []
    for k in range(4):
        b = _ss_scene(dev, 512, seed=10 + k)
        batches.append((b["rays"][0].contiguous(), b["rays"][1].contiguous(), b["tgt"], b["prior"]))
    sc_e, step_e = build()
    losses_e = [step_e(*batches[0]).item()]                      # (the graph's warm-up step)
    for b in batches[1:]:
        losses_e.append(step_e(*b).item())
    sc_g, step_g = build()
    gs = GraphedStep(step_g, sc_g["opt"], batches[0], warmup=1)
    losses_g = [gs(*b).item() for b in batches[1:]]
    torch.cuda.synchronize()
    print(f"  eager {losses_e[1:]} graphed {losses_g}")
    assert losses_g == losses_e[1:]
    assert torch.equal(sc_g["opt"].flat_param, sc_e["opt"].flat_param)
    assert len(set(losses_g)) == len(losses_g), "the replays must see their own batches"


@pytest.mark.parametrize("case", ["single_level", "chunked_fallback"])
def test_ss_step_loss_other_routes(dev, case):
    """ss_step_loss off its main route: a render without a fine network (one level: cnerf_closs_finish_ss with one workspace), and a
    batch larger than `chunk` (the folded form does not apply: render + ss_primary_losses on the launch's `sel`) — both against the
    reference's lines (render, ss_consistency, ss_primary_losses)."""
    from consistentnerf_amd import run_nerf_view as V
    g = golden("ssloss")
    Hh, Ww, far = 32, 40, 7.0
    K, poses = g["K"], g["poses"]
    ro, rd = O.get_rays_np(Hh, Ww, K, poses[0][:3, :4])
    coarse, _ = make_model(4, 128, True, 5, 31, dev)
    fine, _ = make_model(4, 128, True, 5, 32, dev)
    one = case == "single_level"
    kw = _kwargs(coarse, None if one else fine, 16, 0 if one else 16, 0.0, False, 0.0, False)
    kw.update(near=2.0, far=far, ndc=False, use_viewdirs=True)
    chunk = 4096 if one else 64
    sel, r = g["a.sel"], int(g["a.ref_index"])
    rays = torch.stack([T(ro.reshape(-1, 3)[sel], dev), T(rd.reshape(-1, 3)[sel], dev)], 0)
    tgt = T(g["images"][0].reshape(-1, 3)[sel].astype(np.float32), dev)
    prior = T(g["depths"][0].reshape(-1)[sel], dev)
    params = [p for m in ((coarse,) if one else (coarse, fine)) for p in m.parameters()]

    def grads():
        out = [None if p.grad is None else p.grad.detach().clone() for p in params]
        for p in params:
            p.grad = None
        return out

    coins = (1, 1, 0, 0) if one else (1, 0, 1, 1)
    rgb, disp, acc, depth, extras = V.render(Hh, Ww, K, chunk=chunk, rays=rays, retraw=True, **kw)
    ss = V.ss_consistency(rays[0], rays[1], prior, poses[r], K, g["images"][r], g["depths"][r], Hh, Ww, kw, chunk=chunk,
                          occlusion_threshold=0.1, with_depth_loss=True)
    lp, il, il0 = V.ss_primary_losses(rgb, depth, extras, tgt, prior, ss["mask_bound"], ss["mask"], with_depth_loss=True,
                                      coins=list(coins[:2]) if one else list(coins))
    (ss["loss"] + lp).backward()
    g_l = grads()
    loss_o, info = V.ss_step_loss(Hh, Ww, K, rays, tgt, prior, poses[r], g["images"][r], g["depths"][r], kw, chunk=chunk,
                                  occlusion_threshold=0.1, with_depth_loss=True, coins=coins, route="two_renders")
    loss_o.backward()
    g_o = grads()
    ll, lo = (ss["loss"] + lp).item(), loss_o.item()
    print(f"  {case}: loss lines {ll:.8f} one call {lo:.8f}")
    assert abs(lo - ll) <= 2e-6 * abs(ll)
    assert (il0 is None) == one
    for a, b in zip(g_l, g_o):
        assert (a is None) == (b is None)
        if a is not None:
            assert (a - b).abs().max().item() <= 2e-6 * max(a.abs().max().item(), 1e-30)


def test_in_loop_consistency_golden(dev):
    """a15 (VT:905-938): warp of the batch's depth-prior points into a reference view, occlusion threshold doubling,
    second render on the warped rays, the four loss terms and their weight gradients — against the reference's own
    statements executed on the same seeded scene and networks (fixture `ssloss`).  Masks exact; rays 1e-5; rendered
    colours / depths and the loss at the end-to-end render tolerance (5e-3, 5e-3 * far); gradients by summary."""
    from consistentnerf_amd import run_nerf_view as V
    g = golden("ssloss")
    Hh, Ww, far = 32, 40, 7.0
    K, poses = g["K"], g["poses"]
    ro, rd = O.get_rays_np(Hh, Ww, K, poses[0][:3, :4])
    for tag, thr in (("a", 0.1), ("b", 1e-4)):
        coarse, sd_c = make_model(4, 128, True, 5, 31, dev)
        fine, sd_f = make_model(4, 128, True, 5, 32, dev)
        kw = _kwargs(coarse, fine, 16, 16, 0.0, False, 0.0, False)
        kw.update(near=2.0, far=far, ndc=False, use_viewdirs=True, _debug=True)
        sel, r = g[tag + ".sel"], int(g[tag + ".ref_index"])
        out = V.ss_consistency(T(ro.reshape(-1, 3)[sel], dev), T(rd.reshape(-1, 3)[sel], dev),
                               T(g["depths"][0].reshape(-1)[sel], dev), poses[r], K, g["images"][r], g["depths"][r],
                               Hh, Ww, kw, chunk=4096, occlusion_threshold=thr, with_depth_loss=True)
        assert np.array_equal(out["mask_bound"].cpu().numpy(), g[tag + ".mask_bound"])
        assert np.array_equal(out["mask"].cpu().numpy(), g[tag + ".mask"])
        assert abs(2.0 * out["threshold"].item() - float(g[tag + ".thr_next"])) < 1e-6 * float(g[tag + ".thr_next"])
        check(out["batch_rays_ref"], g[tag + ".rays_ref"], 1e-5, "rays_ref")
        check(out["rgb_target_ref"], g[tag + ".rgb_target_ref"], 0.0, "rgb_target_ref")
        check(out["rays_depth_ref"], g[tag + ".rays_depth_ref"], 0.0, "rays_depth_ref")
        check(out["rgb_ref"], g[tag + ".rgb_ref"], 5e-3, "rgb_ref")
        check(out["depth_pred_ref"], g[tag + ".depth_pred_ref"], 5e-3 * far, "depth_pred_ref")
        ref_loss = float(g[tag + ".loss"])
        assert abs(out["loss"].item() - ref_loss) < 5e-3 * ref_loss
        out["loss"].backward()
        check_param_grads(fine, g, tag + ".gf.__full__", tag + ".gf.", rtol=2e-1, l2tol=1e-1)
        check_param_grads(coarse, g, tag + ".gc.__full__", tag + ".gc.", rtol=2e-1, l2tol=1e-1)
        # the loose bounds above compare two FREE-RUNNING renders (the 2^9-frequency encoding amplifies a 1e-7
        # difference of a resampled depth).  The same block with the ORACLE evaluated at the kernel's own sample depths — O.query /
        # O.composite of both levels on the warped rays, then the four img2mse terms of VT:930-938 and autograd — at the
        # equal-sample-set tolerances: maps 2e-5 (depth 2e-5 * far), loss 2e-5 relative, every weight gradient 2e-4 of its tensor's
        # largest (no teacher forcing of the ReLU patterns here: a unit within round-off of zero may flip)
        ncfg = O.NetCfg(4, 128, output_ch=5)
        osd = [O.as_tensors(sd_c, True), O.as_tensors(sd_f, True)]
        rr = out["batch_rays_ref"].detach().cpu()
        o_, d_ = rr[0], rr[1]
        vd = d_ / torch.norm(d_, dim=-1, keepdim=True)
        tgt_o, dtg_o = out["rgb_target_ref"][0].t().cpu(), out["rays_depth_ref"].reshape(-1).cpu()
        ex = out["extras_ref"]
        comps = []
        for sdk, z in ((osd[0], ex["_z_coarse"].cpu()), (osd[1], ex["_z_vals"].cpu())):
            raw_o = O.query(sdk, o_[:, None, :] + d_[:, None, :] * z[:, :, None], vd, ncfg)
            comps.append(O.composite(raw_o, z, d_))
        loss_o = O.mse(comps[1][0], tgt_o) + O.mse(comps[1][4], dtg_o) + O.mse(comps[0][0], tgt_o) + O.mse(comps[0][4], dtg_o)
        check(out["rgb_ref"], comps[1][0].detach(), 2e-5, "rgb_ref at the kernel's depths")
        check(out["depth_pred_ref"], comps[1][4].detach(), 2e-5 * far, "depth_pred_ref at the kernel's depths")
        check(ex["rgb0"], comps[0][0].detach(), 2e-5, "rgb0_ref at the kernel's depths")
        assert abs(out["loss"].item() - loss_o.item()) <= 2e-5 * abs(loss_o.item()), (out["loss"].item(), loss_o.item())
        loss_o.backward()
        for model, sdk, lv in ((coarse, osd[0], "coarse"), (fine, osd[1], "fine")):
            worst = 0.0
            for name, p_ in model.named_parameters():
                if name not in sdk or sdk[name].grad is None:
                    assert p_.grad is None or float(p_.grad.abs().max()) == 0.0, (lv, name)
                    continue
                go = sdk[name].grad
                worst = max(worst, float((p_.grad.cpu() - go).abs().max()) / max(float(go.abs().max()), 1e-30))
            print(f"  {tag} {lv}: worst weight-gradient difference vs the oracle at the kernel's depths {worst:.2e} of the tensor's largest")
            assert worst <= 2e-4, (tag, lv, worst)


@fp32_only
@pytest.mark.parametrize("Nf,vd,perturb", [(128, True, 1.0), (0, True, 1.0), (32, False, 0.0)])
def test_render_fwd_bwd_single_call_equals_python_surface(dev, Nf, vd, perturb):
    """cnerf_render_fwd / cnerf_render_bwd (render_rays and its autograd as one C call each) against the
    Python surface that drives the same kernels launch by launch: every map and every parameter gradient bit for bit,
    with and without a fine level, with and without view directions / jitter."""
    from consistentnerf_amd import ops, run_nerf as R
    from consistentnerf_amd.run_nerf_helpers import pytest_uniform, sample_u
    D, W, och, B, Nc = (8, 256, 5, 96, 64) if vd else (4, 128, 5, 70, 32)
    coarse, _ = make_model(D, W, vd, och, 51, dev)
    fine, _ = make_model(D, W, vd, och, 52, dev)
    rays = T(I.ray_batch(B, seed=9), dev)
    if not vd:
        rays = rays[:, :8].contiguous()
    kw = _kwargs(coarse, fine if Nf else None, Nc, Nf, perturb, True, 0.0, False)
    ret = R.render_rays(rays, retraw=True, pytest=True, _with_depth=True, **kw)
    keys = ["rgb_map", "disp_map", "acc_map", "depth_map"] + (["rgb0", "disp0", "acc0", "depth0"] if Nf else [])
    rs = np.random.RandomState(4)
    gin = {k: T(rs.normal(size=tuple(ret[k].shape)).astype(np.float32), dev) for k in keys}
    sum((ret[k] * gin[k]).sum() for k in keys).backward()
    # the same randoms the surface drew (pytest hook: np.random.seed(0) stream per call)
    t_rand = pytest_uniform((B, Nc), dev) if perturb > 0 else None
    u = sample_u(B, Nf, perturb == 0., True, dev) if Nf else None
    out, st = ops.render_forward(coarse.spec(), R._packed(coarse), fine.spec() if Nf else None,
                                 R._packed(fine) if Nf else None, rays, Nc, Nf, t_rand=t_rand, u=u, white_bkgd=True,
                                 train=True, retraw=True)
    for k in keys + ["raw"] + (["z_std"] if Nf else []):
        assert torch.equal(out[k], ret[k]) or (k.startswith("disp") and np.array_equal(
            out[k].cpu().numpy(), ret[k].detach().cpu().numpy(), equal_nan=True)), k
    gc = [torch.empty_like(p) for p in coarse.kernel_tensors()]
    gf = [torch.empty_like(p) for p in fine.kernel_tensors()] if Nf else None
    ops.render_backward(st, gin, gc, gf)
    def same(gs, model):
        for got, p in zip(gs, model.kernel_tensors()):   # tensors the network does not use get zeros / no .grad
            assert torch.equal(got, p.grad) if p.grad is not None else not got.any()
    same(gc, coarse)
    if Nf:
        same(gf, fine)


def test_plain_c_program_uses_the_abi(dev, tmp_path):
    """The boundary without PyTorch: tests/c_abi/render_smoke.c (C99, gcc, links libcnerf_hip.so + the HIP runtime) packs
    two networks, renders a ray batch and runs the backward through include/cnerf.h; its maps and all parameter
    gradients equal the same calls made from Python through ctypes, bit for bit."""
    import subprocess
    from consistentnerf_amd import _lib, ops, run_nerf as R
    from consistentnerf_amd.run_nerf_helpers import pytest_uniform
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    exe = str(tmp_path / "render_smoke")
    libdir = os.path.dirname(_lib.LIB_PATH)
    cc = ["gcc", "-std=c99", "-O1", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(root, "include"),
          os.path.join(here, "c_abi", "render_smoke.c"), "-o", exe, "-L/opt/rocm/lib", "-lamdhip64", "-L" + libdir,
          "-l:" + os.path.basename(_lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath," + libdir]
    r = subprocess.run(cc, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    D, W, B, Nc, Nf = 8, 256, 80, 64, 128
    coarse, _ = make_model(D, W, True, 5, 61, dev)
    fine, _ = make_model(D, W, True, 5, 62, dev)
    rays = T(I.ray_batch(B, seed=12), dev)
    t_rand, u = pytest_uniform((B, Nc), dev), pytest_uniform((B, Nf), dev)
    keys = ["rgb_map", "disp_map", "acc_map", "depth_map", "rgb0", "disp0", "acc0", "depth0"]
    rs = np.random.RandomState(6)
    gin = {k: T(rs.normal(size=(B, 3) if k.startswith("rgb") else (B,)).astype(np.float32), dev) for k in keys}
    # reference: the same entry points through ctypes
    out, st = ops.render_forward(coarse.spec(), R._packed(coarse), fine.spec(), R._packed(fine), rays, Nc, Nf,
                                 t_rand=t_rand, u=u, white_bkgd=True, train=True, retraw=True)
    gc = [torch.empty_like(p) for p in coarse.kernel_tensors()]
    gf = [torch.empty_like(p) for p in fine.kernel_tensors()]
    ops.render_backward(st, gin, gc, gf)
    torch.cuda.synchronize()
    # the blob the C program reads
    blob = [np.array([D, W, 10, 4, 1, 5, 4, B, Nc, Nf, 1, 11], np.int32).tobytes()]
    for m in (coarse, fine):
        blob += [p.detach().cpu().numpy().astype(np.float32).tobytes() for p in m.kernel_tensors()]
    blob += [rays.cpu().numpy().tobytes(), ops._t_vals(Nc, dev).cpu().numpy().tobytes(), t_rand.cpu().numpy().tobytes(),
             u.cpu().numpy().tobytes()] + [gin[k].cpu().numpy().tobytes() for k in keys]
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    fin.write_bytes(b"".join(blob))
    r = subprocess.run([exe, str(fin), str(fout)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "render_smoke ok" in r.stdout, r.stdout + r.stderr
    got = np.frombuffer(fout.read_bytes(), np.float32)
    want = [out[k] for k in keys] + [out["z_std"], out["raw"]]
    off = 0
    for k, t in zip(keys + ["z_std", "raw"], want):
        n = t.numel()
        assert np.array_equal(got[off:off + n], t.cpu().numpy().reshape(-1), equal_nan=True), k
        off += n
    off += 2 * B * (Nc + Nf)            # z_vals and weights of the last level (not returned by ops.render_forward)
    for gs in (gc, gf):
        for i, t in enumerate(gs):
            n = t.numel()
            assert np.array_equal(got[off:off + n], t.cpu().numpy().reshape(-1)), f"grad tensor {i}"
            off += n
    assert off == got.size


def test_rays_are_independent_and_tiny_batches_work(dev):
    """Size-independent property: a ray's outputs do not depend on what else is in the batch, bit for bit (each output is
    one fmaf chain in a fixed order wherever the point lands in a tile) — B = 1, 2, 33 against rows of B = 70; an empty
    batch returns empty maps."""
    from consistentnerf_amd import run_nerf as R
    coarse, _ = make_model(8, 256, True, 5, 71, dev)
    fine, _ = make_model(8, 256, True, 5, 72, dev)
    kw = _kwargs(coarse, fine, 64, 128, 0.0, False, 0.0, False)
    rays = T(I.ray_batch(70, seed=31), dev)
    with torch.no_grad():
        full = R.render_rays(rays, retraw=True, _with_depth=True, **kw)
        for lo, n in ((0, 1), (5, 2), (17, 33), (69, 1)):
            part = R.render_rays(rays[lo:lo + n].contiguous(), retraw=True, _with_depth=True, **kw)
            for k in ("rgb_map", "depth_map", "acc_map", "rgb0", "depth0", "z_std", "raw"):
                assert torch.equal(part[k], full[k][lo:lo + n]), (k, lo, n)
        empty = R.render_rays(rays[:0].contiguous(), retraw=True, _with_depth=True, **kw)
        assert empty["rgb_map"].shape == (0, 3) and empty["raw"].shape == (0, 192, 4) and empty["z_std"].shape == (0,)


def test_wgrad_beyond_128_point_ranges(dev):
    """9.2 M points in ONE backward call (more than 128 x 65 536: the point ranges per GEMM grow past 128 so that a range's
    rows stay within 32-bit byte offsets): its weight gradients equal the sum of the gradients of the two half batches
    (each on the usual 128-range path) to fp32 summation accuracy."""
    from consistentnerf_amd import ops
    from consistentnerf_amd.run_nerf import _packed
    model, _ = make_model(4, 128, True, 4, 81, dev)
    spec, packed = model.spec(), _packed(model)
    B, S = 36000, 256
    g = torch.Generator(device=dev).manual_seed(5)
    pts = (torch.rand(B * S, 3, device=dev, generator=g) * 4 - 2)
    dirs = torch.nn.functional.normalize(torch.randn(B, 3, device=dev, generator=g), dim=-1)
    G = torch.randn(B * S, 4, device=dev, generator=g) * 1e-3

    def grads(lo, hi):
        raw, stash = ops.mlp_forward(spec, packed, hi - lo, S, pts=pts[lo * S:hi * S], dirs=dirs[lo:hi], want_stash=True)
        out = ops.mlp_backward(spec, packed, G[lo * S:hi * S].contiguous(), hi - lo, S, stash)
        del stash
        return out
    whole = grads(0, B)
    a, b = grads(0, B // 2), grads(B // 2, B)
    for i, (w, x, y) in enumerate(zip(whole, a, b)):
        ref = x.double() + y.double()
        scale = float(ref.abs().max())
        if scale == 0.0:
            assert not w.any()
            continue
        assert float((w.double() - ref).abs().max()) <= 2e-5 * scale, f"tensor {i}"


@fp32_only
def test_inference_runs_the_inference_kernel(dev):
    """Under torch.no_grad() (render_path, evaluation) the forward must be the stash-free kernel: a Function sees
    needs_input_grad = True for parameters even when grad mode is off, which once made every inference pass write a
    10 KB-per-point training stash.  With grad mode on (training) the stash variant runs."""
    from consistentnerf_amd import ops, run_nerf as R
    coarse, _ = make_model(4, 128, True, 5, 91, dev)
    fine, _ = make_model(4, 128, True, 5, 92, dev)
    kw = _kwargs(coarse, fine, 16, 16, 0.0, False, 0.0, False)
    rays = T(I.ray_batch(40, seed=2), dev)
    try:
        ops.PROFILE = []
        with torch.no_grad():
            a = R.render_rays(rays, **kw)
            xe = torch.cat([O.embed(T(I.ray_batch(8, seed=1)[:, :3]), 10), O.embed(T(I.ray_batch(8, seed=1)[:, 3:6]), 4)], -1)
            coarse(xe.to(dev))
        assert {n for n, *_ in ops.PROFILE} == {"mlp_fwd"}
        ops.PROFILE = []
        b = R.render_rays(rays, **kw)
        assert {n for n, *_ in ops.PROFILE} == {"mlp_fwd_train"}
    finally:
        ops.PROFILE = None
    assert torch.equal(a["rgb_map"], b["rgb_map"].detach())


@fp32_only
def test_autograd_usage_patterns(dev):
    """Host-side plumbing under less common autograd uses: a frozen coarse network (inference kernel for it, gradients only
    for the fine one), two forward passes before one backward (gradients add up; each pass keeps its own packed weights and
    stash), a side stream, torch.autograd.grad instead of .backward()."""
    from consistentnerf_amd import ops, run_nerf as R
    coarse, _ = make_model(4, 128, True, 5, 93, dev)
    fine, _ = make_model(4, 128, True, 5, 94, dev)
    kw = _kwargs(coarse, fine, 16, 16, 0.0, False, 0.0, False)
    r1, r2 = T(I.ray_batch(33, seed=4), dev), T(I.ray_batch(21, seed=5), dev)
    tgt1, tgt2 = torch.rand(33, 3, device=dev), torch.rand(21, 3, device=dev)

    def loss_of(rays, tgt):
        o = R.render_rays(rays, **kw)
        return R.img2mse(o["rgb_map"], tgt) + R.img2mse(o["rgb0"], tgt)

    params = [p for m in (coarse, fine) for p in m.kernel_tensors()]
    g1 = torch.autograd.grad(loss_of(r1, tgt1), params, allow_unused=True)
    g2 = torch.autograd.grad(loss_of(r2, tgt2), params, allow_unused=True)
    (loss_of(r1, tgt1) + loss_of(r2, tgt2)).backward()               # two graphs alive, one backward
    for p, a, b in zip(params, g1, g2):
        if a is None:
            assert p.grad is None or not p.grad.any()
            continue
        ref = a.double() + b.double()
        assert float((p.grad.double() - ref).abs().max()) <= 1e-6 * max(float(ref.abs().max()), 1e-30)
    # frozen coarse network
    for p in coarse.parameters():
        p.requires_grad_(False)
    for p in fine.parameters():
        p.grad = None
    try:
        ops.PROFILE = []
        loss_of(r1, tgt1).backward()
        kinds = [n for n, *_ in ops.PROFILE]
    finally:
        ops.PROFILE = None
    assert kinds.count("mlp_fwd") == 1 and kinds.count("mlp_fwd_train") == 1 and kinds.count("mlp_wgrad") == 1
    for p, a in zip(fine.kernel_tensors(), g1[len(coarse.kernel_tensors()):]):
        if a is not None:
            assert float((p.grad - a).abs().max()) <= 1e-6 * max(float(a.abs().max()), 1e-30)
    # a side stream
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s), torch.no_grad():
        side = R.render_rays(r2, **kw)["rgb_map"]
    torch.cuda.current_stream().wait_stream(s)
    with torch.no_grad():
        assert torch.equal(side, R.render_rays(r2, **kw)["rgb_map"])


def test_static_camera_and_render_factor(dev):
    """render(c2w=..., c2w_staticcam=...) (R:104-108): geometry from the static camera, view directions from the moving one;
    render_path(render_factor=f) (R:145-149) renders H//f x W//f with focal/f and the caller's K (as the reference does)."""
    from consistentnerf_amd import run_nerf as R
    from consistentnerf_amd.run_nerf_helpers import get_rays
    g = golden("render_full_tiny")
    K = g["K"]
    c2w, cam = T(g["c2w"], dev), T(I.camera_pose(30.0, -10.0, 4.0)[:3, :4], dev)
    batch, sh = R._ray_batch(16, 16, K, None, c2w, False, 2.0, 6.0, True, cam, dev)
    ro, rd = get_rays(16, 16, K, cam)
    _, rd_mov = get_rays(16, 16, K, c2w)
    assert sh == (16, 16)
    check(batch[:, 0:3], ro.reshape(-1, 3).cpu().numpy(), 0.0, "static origin")
    check(batch[:, 3:6], rd.reshape(-1, 3).cpu().numpy(), 1e-6, "static direction")
    vd = rd_mov.reshape(-1, 3) / rd_mov.reshape(-1, 3).norm(dim=-1, keepdim=True)
    check(batch[:, 8:11], vd.cpu().numpy(), 1e-6, "moving view direction")
    coarse, _ = make_model(4, 128, True, 5, 31, dev)
    kw = _kwargs(coarse, None, 16, 0, 0.0, False, 0.0, False)
    kw.update(near=2.0, far=6.0, ndc=False, use_viewdirs=True)
    pose4 = torch.cat([c2w[:3, :4], torch.tensor([[0., 0., 0., 1.]], device=dev)], 0)
    rgbs, disps = R.render_path([pose4], (32, 32, float(K[0][0]) * 2), K, 4096, kw, render_factor=2)
    assert rgbs.shape == (1, 16, 16, 3) and disps.shape == (1, 16, 16)
    with torch.no_grad():
        direct = R.render(16, 16, K, chunk=4096, c2w=c2w[:3, :4], **kw)[0]
    assert np.array_equal(rgbs[0], direct.cpu().numpy())


def test_input_gradients_fail_loudly(dev):
    """Only parameter gradients exist (as in the reference's training loops): asking for a gradient w.r.t. the inputs must
    raise, not return None silently."""
    from consistentnerf_amd import ops
    model, _ = make_model(4, 128, True, 5, 95, dev)
    x = torch.rand(16, 90, device=dev, requires_grad=True)
    with pytest.raises(ops.CnerfError):
        model(x)
    assert model(x.detach()).shape == (16, 4)


# ------------------------------------------------------------------------------------------------
# round 2: the branches C3 / C4 rely on — clip + grad_scale in the optimiser tail (V:1983), masked means under sharding,
# the coin-flip consumers of the in-loop masks (VT:941-969), FusedAdam under torch.autograd.grad
def _view_args(tmp):
    import argparse
    return argparse.Namespace(
        multires=10, i_embed=0, use_viewdirs=True, multires_views=4, N_importance=32, netdepth=4, netwidth=128,
        netdepth_fine=4, netwidth_fine=128, netchunk=1024 * 64, lrate=5e-4, lrate_decay=250, basedir=tmp, expname="exp",
        ft_path=None, no_reload=True, perturb=1.0, N_samples=32, stable_init=True, white_bkgd=False, raw_noise_std=0.0,
        dataset_type="dtu", no_ndc=True, lindisp=False)


@pytest.mark.parametrize("route", ["fused_clip", "torch_clip_on_grad_views"])
def test_train_10_steps_view_variant_golden(dev, route):
    """The ConsistentNeRF loop against the reference's own 10 steps (`train_10steps_V`: V.create_nerf, V.render, the masked
    rgb + depth losses on both levels V:1645-1648 / 1737 / 1786-1788 / 1865, then the reference's own tail V:1982-1994 =
    backward, clip_grad_value_(grad_vars, 0.1), Adam, lr decay).  Two routes: the clip folded into the Adam kernel
    (`clip_value=0.1`), and torch's own clip_grad_value_ on grad_vars (whose .grad are views of FusedAdam's flat buffer)
    followed by the unclipped kernel.  Pinned where the clip matters: the Adam moments and final values of the parameters
    whose gradient exceeded 0.1 at step 0 — the fixture's no-clip control differs there by 24 % / >50 % / >1e-4."""
    import tempfile
    from consistentnerf_amd import run_nerf_view as V
    g = golden("train_10steps_V")
    with tempfile.TemporaryDirectory() as tmp:
        args = _view_args(tmp)
        kw_train, kw_test, start, grad_vars, optimizer = V.create_nerf(args)
    c0, f0 = kw_train["network_fn"].state_dict(), kw_train["network_fine"].state_dict()
    assert all(torch.equal(c0[k], f0[k]) for k in c0), "V:321: the coarse net starts as a copy of the fine one"
    for net, seed in ((kw_train["network_fn"], 51), (kw_train["network_fine"], 52)):
        net.load_state_dict({k: T(v) for k, v in I.nerf_state_dict(4, 128, 10, 4, 5, True, seed=seed, gain=0.6).items()})
    near, far = 2.0, 6.0
    kw_train.update(near=near, far=far)
    K = I.intrinsics(100, 100, 138.0)
    if route == "fused_clip":
        optimizer.param_groups[0]["clip_value"] = 0.1
    global_step = start
    for i in range(10):
        rays = T(I.ray_batch(256, seed=300 + i, near=near, far=far), dev)
        rs = np.random.RandomState(400 + i)
        target = T(rs.uniform(size=(256, 3)).astype(np.float32), dev)
        prior = T(rs.uniform(near, far, size=(256,)).astype(np.float32), dev)
        mask = T((rs.uniform(size=(256,)) < 0.6).astype(np.float32), dev)
        rgb, disp, acc, depth, extras = V.render(100, 100, K, chunk=32768, rays=torch.stack([rays[:, 0:3], rays[:, 3:6]], 0),
                                                 retraw=True, pytest=True, **kw_train)
        optimizer.zero_grad()
        il, dl = V.hardmask_losses(rgb, target, mask, 0.2, depth, prior, far)
        il0, dl0 = V.hardmask_losses(extras["rgb0"], target, mask, 0.2, extras["depth0"], prior, far)
        for t, k in zip((il, dl, il0, dl0), ("img_loss", "depth_loss", "img_loss0", "depth_loss0")):
            rel = abs(t.item() - g[k][i]) / abs(g[k][i])
            assert rel < 1e-3, (i, k, t.item(), g[k][i])
        loss = il + dl + il0 + dl0
        loss.backward()
        n_over = int((optimizer.flat_grad.abs() > 0.1).sum())
        print(f"  step {i}: loss {loss.item():.6f} ref {g['loss'][i]:.6f}; |g| > 0.1: {n_over} (ref {int(g['n_clipped'][i])})")
        if i == 0:
            assert np.array_equal(torch.nonzero(optimizer.flat_grad.abs() > 0.1).reshape(-1).cpu().numpy(), g["clipped_idx"])
        if route == "torch_clip_on_grad_views":
            torch.nn.utils.clip_grad_value_(grad_vars, 0.1)
            assert float(optimizer.flat_grad.abs().max()) <= 0.1000001
        optimizer.step()
        new_lrate = args.lrate * (0.1 ** (global_step / (args.lrate_decay * 1000)))
        for pg in optimizer.param_groups:
            pg["lr"] = new_lrate
        global_step += 1
    assert abs(optimizer.param_groups[0]["lr"] - float(g["lr_final"])) < 1e-12
    idx = T(g["clipped_idx"], dev)
    check(optimizer.flat_param[idx], g["final_at_clipped"], 2e-6, "final values of the clipped parameters")   # measured 7e-9
    effect = np.abs(g["final_at_clipped"] - g["final_at_clipped_noclip"])
    assert np.median(effect) > 1e-4
    m, m_ref, m_noclip = optimizer.exp_avg[idx].cpu().numpy(), g["exp_avg_at_clipped"], g["exp_avg_at_clipped_noclip"]
    v, v_ref = optimizer.exp_avg_sq[idx].cpu().numpy(), g["exp_avg_sq_at_clipped"]
    rel_m, rel_v = np.abs(m - m_ref) / np.abs(m_ref).max(), np.abs(v - v_ref) / np.abs(v_ref).max()
    print(f"  Adam moments at the clipped parameters: rel |d m| {rel_m.max():.2e}  rel |d v| {rel_v.max():.2e}; the no-clip "
          f"control is {(np.abs(m_noclip - m_ref) / np.abs(m_ref).max()).max():.2f} away")
    assert rel_m.max() < 2e-4 and rel_v.max() < 2e-4       # measured 1.4e-6 / 1.3e-5; the no-clip control is O(1) away
    assert (np.abs(m_noclip - m_ref) / np.abs(m_ref).max()).max() > 0.2
    worst = 0.0
    for tag, net in (("c", kw_train["network_fn"]), ("f", kw_train["network_fine"])):
        for k, p in net.state_dict().items():
            worst = max(worst, float(np.abs(p.reshape(-1)[::7].cpu().numpy() - g[f"final.{tag}.{k}.sub"]).max()))
    print(f"  final weights: max|d| = {worst:.3e}")
    assert worst < 5e-5                                    # measured 3.3e-6


@pytest.mark.parametrize("clip,grad_scale", [(0.0, 1.0), (0.1, 1.0), (0.0, 0.125), (0.05, 3.0)])
def test_adam_kernel_clip_and_grad_scale(dev, clip, grad_scale):
    """cnerf_adam_step: gradient scaling, then clip_grad_value_ (V:1983), then torch.optim.Adam's update — 3 steps against
    the oracle's ATen Adam on a buffer whose gradients straddle the clip."""
    from consistentnerf_amd import ops
    rs = np.random.RandomState(5)
    n = 100003
    p0 = rs.normal(size=n).astype(np.float32)
    p, m, v = T(p0, dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    pr, mr, vr = T(p0), torch.zeros(n), torch.zeros(n)
    for step in (1, 2, 3):
        g = (rs.normal(size=n) * 0.2 / max(grad_scale, 1e-3) * (1.0 if step < 3 else 1e-3)).astype(np.float32)
        ops.adam_step(p, T(g, dev), m, v, step, 5e-4, clip=clip, grad_scale=grad_scale)
        O.adam_step(pr, T(g) * grad_scale, mr, vr, step, 5e-4, clip=clip)
        if clip > 0 and step == 1:
            assert (np.abs(g * grad_scale) > clip).mean() > 0.2
    # one fp32 rounding per operation on each side, in slightly different association (lerp / addcmul / addcdiv): a few ulp
    # of each element (the hyper-parameters cross the C ABI as doubles: 1 - beta2 from a float-rounded 0.999 would be off by
    # 1.3e-5 relative, which this bound catches)
    for name, a_, b_ in (("exp_avg", m, mr), ("exp_avg_sq", v, vr), ("param", p, pr)):
        a_, b_ = a_.cpu().double(), b_.double()
        # relative to the element, with a floor for elements that are differences of nearly equal terms (exp_avg mixes signs)
        floor = {"exp_avg": 0.05, "exp_avg_sq": 1e-6, "param": 1e-3}[name] * float(b_.abs().max())
        rel = float(((a_ - b_).abs() / b_.abs().clamp_min(floor)).max())
        print(f"  {name}: max relative |d| = {rel:.2e}")
        assert rel <= 2e-6, (name, rel)


@pytest.mark.parametrize("nshards", [2, 3])
def test_masked_losses_global_counts_sharded(dev, nshards):
    """SURVEY hard part 7: masked means under ray sharding.  Each shard calls the loss kernel with the GLOBAL set sizes
    (`counts`, distributed.global_mask_counts): the shard losses add up to — and the shard gradients concatenate to — the
    single-call result and the reference capture (`losses_mask`)."""
    from consistentnerf_amd import distributed as D, run_nerf_view as V
    g = golden("losses_mask")
    far, c = float(g["far"]), float(g["coef"])
    B = g["mask"].shape[0]
    mask = T(g["mask"], dev)
    counts = D.global_mask_counts(mask)
    assert counts.tolist() == [float((g["mask"] == 1).sum()), float((g["mask"] == 0).sum())]
    bounds = [D.shard_bounds(B, r, nshards) for r in range(nshards)]
    assert bounds[0][0] == 0 and bounds[-1][1] == B and all(a[1] == b[0] for a, b in zip(bounds, bounds[1:]))
    tot_rgb, tot_dep, d_rgb, d_dep = 0.0, 0.0, [], []
    for lo, hi in bounds:
        r = T(g["rgb"][lo:hi], dev).requires_grad_(True)
        d = T(g["depth"][lo:hi], dev).requires_grad_(True)
        lr, ld = V.hardmask_losses(r, T(g["target"][lo:hi], dev), mask[lo:hi], c, d, T(g["prior"][lo:hi], dev), far,
                                   counts=counts)
        (lr + ld).backward()
        tot_rgb, tot_dep = tot_rgb + lr.item(), tot_dep + ld.item()
        d_rgb.append(r.grad); d_dep.append(d.grad)
    assert abs(tot_rgb - float(g["mixed.l_rgb"])) < 1e-7 and abs(tot_dep - float(g["mixed.l_depth"])) < 1e-7
    check(torch.cat(d_rgb), g["mixed.d_rgb"], 1e-9, "d_rgb (sharded, global counts)")
    check(torch.cat(d_dep), g["mixed.d_depth"], 1e-9, "d_depth (sharded, global counts)")
    # without the global counts the shard means are local means: NOT the same thing (the branch is doing something)
    lr_local, _ = V.hardmask_losses(T(g["rgb"][:bounds[0][1]], dev), T(g["target"][:bounds[0][1]], dev),
                                    mask[:bounds[0][1]], c)
    assert abs(lr_local.item() * 1.0 - tot_rgb) > 1e-4


def test_ss_primary_losses_golden(dev):
    """VT:941-969: the primary render's rgb / depth terms restricted to `[mask_bound][mask]` per coin, incl. the coarse rgb
    term's fallback to the fine rgb — value and gradients against the reference's own statements (`ssloss_primary`)."""
    from consistentnerf_amd import run_nerf_view as V
    g = golden("ssloss_primary")
    for wd in (True, False):
        for coins in ((1, 1, 1, 1), (0, 0, 0, 0), (1, 0, 0, 1), (0, 1, 1, 0)):
            tag = f"{'d' if wd else 'n'}{''.join(map(str, coins))}."
            leaf = {k: T(g[k], dev).requires_grad_(True) for k in ("rgb", "rgb0", "depth_pred", "depth0")}
            seq = coins if wd else (coins[0], coins[2])
            loss, il, il0 = V.ss_primary_losses(leaf["rgb"], leaf["depth_pred"], dict(rgb0=leaf["rgb0"], depth0=leaf["depth0"]),
                                                T(g["target_s"], dev), T(g["depth_cas_s"], dev), T(g["mask_bound"], dev),
                                                T(g["mask"], dev), wd, seq)
            check(loss, g[tag + "loss"], 2e-6, tag + "loss"); check(il, g[tag + "img_loss"], 2e-7, tag + "img_loss")
            check(il0, g[tag + "img_loss0"], 2e-7, tag + "img_loss0")
            check(V.mse2psnr(il), g[tag + "psnr"], 2e-5, tag + "psnr")
            loss.backward()
            for k, t in leaf.items():
                got = t.grad if t.grad is not None else torch.zeros_like(t)
                check(got, g[tag + "d_" + k], 2e-8 if "rgb" in k else 2e-6, tag + "d_" + k)
    # coins=None draws like the reference: random.randint(0, 1) in call order
    import random
    random.seed(3)
    expect = [random.randint(0, 1) for _ in range(4)]
    random.seed(3)
    leaf = {k: T(g[k], dev) for k in ("rgb", "rgb0", "depth_pred", "depth0")}
    a = V.ss_primary_losses(leaf["rgb"], leaf["depth_pred"], dict(rgb0=leaf["rgb0"], depth0=leaf["depth0"]),
                            T(g["target_s"], dev), T(g["depth_cas_s"], dev), T(g["mask_bound"], dev), T(g["mask"], dev), True)
    b = V.ss_primary_losses(leaf["rgb"], leaf["depth_pred"], dict(rgb0=leaf["rgb0"], depth0=leaf["depth0"]),
                            T(g["target_s"], dev), T(g["depth_cas_s"], dev), T(g["mask_bound"], dev), T(g["mask"], dev), True,
                            expect)
    assert a[0].item() == b[0].item()


def test_fused_adam_params_under_autograd_grad(dev):
    """torch.autograd.grad on FusedAdam-owned parameters: the engine captures instead of accumulating, so _MlpFn.backward
    must hand back real tensors and must NOT touch the optimizer's flat gradient (it used to write into it and return
    None).  .backward() afterwards still takes the direct route (no per-tensor gradients, flat buffer filled)."""
    from consistentnerf_amd import ops, run_nerf as R
    from consistentnerf_amd.optim import FusedAdam
    coarse, _ = make_model(4, 128, True, 5, 93, dev)
    fine, _ = make_model(4, 128, True, 5, 94, dev)
    kw = _kwargs(coarse, fine, 16, 16, 0.0, False, 0.0, False)
    rays, tgt = T(I.ray_batch(40, seed=4), dev), torch.rand(40, 3, device=dev)

    def loss_of():
        o = R.render_rays(rays, **kw)
        return R.img2mse(o["rgb_map"], tgt) + R.img2mse(o["rgb0"], tgt)
    params = [p for m in (coarse, fine) for p in m.kernel_tensors()]
    ref = torch.autograd.grad(loss_of(), params)                      # plain nn.Parameters: tensor route
    opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
    opt.zero_grad()
    got = torch.autograd.grad(loss_of(), params)                      # FusedAdam-owned: must still be the tensor route
    for a, b in zip(got, ref):
        assert a is not None and torch.equal(a, b)
    assert float(opt.flat_grad.abs().max()) == 0.0, "autograd.grad must not write into the optimizer's gradient buffer"
    loss_of().backward()
    flat_ref = torch.cat([(dict(zip(map(id, params), ref)).get(id(p), torch.zeros_like(p))).reshape(-1) for p in opt.params])
    assert float((opt.flat_grad - flat_ref).abs().max()) <= 1e-6 * float(flat_ref.abs().max())
    loss_of().backward(inputs=params)                                 # accumulating backward with explicit inputs: direct too
    assert float((opt.flat_grad - 2 * flat_ref).abs().max()) <= 2e-6 * float(flat_ref.abs().max())


DIST_GPU_WORKER = r"""
import os, sys, tempfile, numpy as np, torch
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _inputs as I
from test_gpu_parity import _view_args
from consistentnerf_amd import distributed as D, run_nerf_view as V
import torch.distributed as dist
dev = torch.device("cuda:0")

def run(use_dist):
    with tempfile.TemporaryDirectory() as tmp:
        args = _view_args(tmp)
        kw, _, start, grad_vars, opt = V.create_nerf(args)
    for net, seed in ((kw["network_fn"], 51), (kw["network_fine"], 52)):
        net.load_state_dict({k: torch.from_numpy(v) for k, v in I.nerf_state_dict(4, 128, 10, 4, 5, True, seed=seed, gain=0.6).items()})
    kw.update(near=2.0, far=6.0)
    opt.param_groups[0]["clip_value"] = 0.1
    red = D.GradReducer(opt, [kw["network_fn"], kw["network_fine"]], mean=False, timing=True) if use_dist else None
    K = I.intrinsics(100, 100, 138.0)
    losses = []
    for i in range(3):
        rays = torch.from_numpy(I.ray_batch(256, seed=300 + i)).to(dev)
        rs = np.random.RandomState(400 + i)
        target = torch.from_numpy(rs.uniform(size=(256, 3)).astype(np.float32)).to(dev)
        prior = torch.from_numpy(rs.uniform(2, 6, size=(256,)).astype(np.float32)).to(dev)
        mask = torch.from_numpy((rs.uniform(size=(256,)) < 0.6).astype(np.float32)).to(dev)
        rays, target, prior, mask = D.shard_batch(rays, target, prior, mask)      # this rank's rays (all of them at world 1)
        counts = D.global_mask_counts(mask) if use_dist else None
        rgb, disp, acc, depth, ex = V.render(100, 100, K, chunk=32768, rays=torch.stack([rays[:, 0:3], rays[:, 3:6]], 0),
                                             retraw=True, pytest=True, **kw)
        opt.zero_grad()
        il, dl = V.hardmask_losses(rgb, target, mask, 0.2, depth, prior, 6.0, counts=counts)
        il0, dl0 = V.hardmask_losses(ex["rgb0"], target, mask, 0.2, ex["depth0"], prior, 6.0, counts=counts)
        loss = il + dl + il0 + dl0
        loss.backward()
        if use_dist:
            red.finish()
        opt.step()
        losses.append(loss.item())
    torch.cuda.synchronize()
    return opt.flat_param.clone(), losses, (red.exposed_ms() if use_dist else None)

p0, l0, _ = run(False)
assert not dist.is_initialized()
rank, world, local = D.init_from_env("nccl")            # CNERF_FORCE_DIST=1: a 1-rank RCCL group
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
p1, l1, ms = run(True)
assert l0 == l1, (l0, l1)
assert torch.equal(p0, p1), float((p0 - p1).abs().max())
assert len(ms) == 3 and all(m >= 0 for m in ms)
D.barrier()
dist.destroy_process_group()
print("DIST_GPU_OK", l1, ms)
"""


def test_forced_dist_world1_step_is_bit_identical(dev, tmp_path):
    """The full product step of the data-parallel path — shard_batch -> global_mask_counts (RCCL) -> hardmask_losses(counts=)
    -> backward (HIP kernels accumulate into FusedAdam.flat_grad) -> GradReducer (RCCL all-reduce of the per-network slices,
    issued from _MlpFn.backward) -> FusedAdam(clip) — on a 1-rank `nccl` group (CNERF_FORCE_DIST=1), 3 steps: losses and
    final parameters bit-identical to the same steps without a process group.  (World > 1 needs more than the one GPU of the
    test box; the 2-rank logic runs under gloo in tests/test_host.py.)"""
    import subprocess
    import sys
    script = tmp_path / "dist_gpu_worker.py"
    script.write_text(DIST_GPU_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CNERF_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
               MASTER_PORT="29577", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script), root], capture_output=True, text=True, env=env, timeout=900)
    print(r.stdout[-1500:])
    assert r.returncode == 0 and "DIST_GPU_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.parametrize("cfg0,cfg1,M0,M1", [((8, 256, True, 5), (8, 256, True, 5), 3 * 192, 3 * 64),
                                             ((8, 256, True, 5), (4, 128, True, 5), 2 * 192 + 7, 1000),
                                             ((4, 128, False, 5), (4, 128, False, 5), 33, 4097),
                                             # point counts whose ceil-divided ranges leave the last of the nominal 19 / 63 empty
                                             # (round 3: the plan drops it; the kernel treats a range past the end as empty)
                                             ((8, 256, True, 5), (8, 256, True, 5), 51 * 192, 51 * 64),
                                             ((8, 256, True, 5), (8, 256, True, 5), 513 * 192, 513 * 64)])
def test_bwd_pair_equals_two_separate_backwards(dev, cfg0, cfg1, M0, M1):
    """cnerf_mlp_bwd_pair (one dgrad grid when the architectures match, one wgrad grid + one reduction always) against two
    cnerf_mlp_bwd calls: bit-identical gradients, overwrite and accumulate, ragged point counts, mixed architectures."""
    from consistentnerf_amd import ops
    nets = []
    for (D, W, vd, och), M, seed in ((cfg0, M0, 61), (cfg1, M1, 62)):
        model, _ = make_model(D, W, vd, och, seed, dev)
        spec = model.spec()
        packed = ops.pack_weights(spec, model.kernel_tensors())
        rs = np.random.RandomState(seed)
        pts = T(rs.uniform(-2, 2, size=(M, 3)).astype(np.float32), dev)
        dirs = T(rs.normal(size=(M, 3)).astype(np.float32), dev) if vd else None
        raw, stash = ops.mlp_forward(spec, packed, M, 1, pts=pts, dirs=dirs, want_stash=True)
        d_raw = T(rs.normal(size=tuple(raw.shape)).astype(np.float32), dev)
        nets.append((spec, packed, d_raw, M, stash))
    sep = [ops.mlp_backward(sp, pk, dr, M, 1, st) for sp, pk, dr, M, st in nets]
    got = [[torch.full_like(g, 7.0) for g in gs] for gs in sep]
    (s0, p0, d0, m0, st0), (s1, p1, d1, m1, st1) = nets
    ops.mlp_backward_pair(s0, p0, d0, m0, 1, st0, got[0], s1, p1, d1, m1, 1, st1, got[1], accumulate=False)
    for a, b in zip(got[0] + got[1], sep[0] + sep[1]):
        assert torch.equal(a, b)
    ops.mlp_backward_pair(s0, p0, d0, m0, 1, st0, got[0], s1, p1, d1, m1, 1, st1, got[1], accumulate=True)
    for a, b in zip(got[0] + got[1], sep[0] + sep[1]):
        assert torch.equal(a, b + b)
    with pytest.raises(ops.CnerfError):      # one gradient set for both networks would race in the reduction
        ops.mlp_backward_pair(s0, p0, d0, m0, 1, st0, got[0], s0, p0, d0, m0, 1, st0, got[0])


def test_training_step_merges_the_two_levels_backward(dev):
    """With both networks FusedAdam-owned, loss.backward() runs ONE dgrad and ONE wgrad launch for the coarse and the fine
    level together (the fine node parks its inputs, the coarse node launches the pair); the flat gradient is bit-identical
    to the unmerged route (CNERF_MERGE_BWD=0), also when only the fine level is in the loss (nothing is parked then), and
    torch.autograd.grad still returns per-tensor gradients."""
    from consistentnerf_amd import ops, run_nerf as R
    from consistentnerf_amd.optim import FusedAdam
    coarse, fine, rays = _c2(dev, 192)
    kw = _kwargs(coarse, fine, 64, 128, 0.0, False, 0.0, False)
    tgt = torch.rand(192, 3, device=dev)
    opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)

    def run(merge, both=True):
        old, R.MERGE_BWD = R.MERGE_BWD, merge
        try:
            ops.PROFILE = []
            opt.zero_grad()
            out = R.render_rays(rays, **kw)
            loss = R.img2mse(out["rgb_map"], tgt) + (R.img2mse(out["rgb0"], tgt) if both else 0.0)
            loss.backward()
            kinds = [n for n, *_ in ops.PROFILE]
        finally:
            ops.PROFILE, R.MERGE_BWD = None, old
        opt.materialize_grad()       # (a network whose backward did not run: its dropped gradient reads as zeros)
        return opt.flat_grad.clone(), kinds
    g_sep, k_sep = run(False)
    g_mrg, k_mrg = run(True)
    DG, WG = ("mlp_dgrad_bf3", "mlp_wgrad_bf3") if BF3 else ("mlp_dgrad", "mlp_wgrad")   # (W = 128, viewdirs: covered by bf16x3)
    assert k_sep.count(DG) == 2 and k_sep.count(WG) == 2
    assert k_mrg.count(DG) == 1 and k_mrg.count(WG) == 1
    assert torch.equal(g_sep, g_mrg) and float(g_mrg.abs().max()) > 0
    g1, k1 = run(True, both=False)            # the coarse node does not run: the fine node must not park
    g2, k2 = run(False, both=False)
    assert k1.count(DG) == 1 and torch.equal(g1, g2)
    n_c = sum(p.numel() for p in coarse.parameters())
    assert float(g1[:n_c].abs().max()) == 0.0 and float(g1[n_c:].abs().max()) > 0
    params = [p for m in (coarse, fine) for p in m.kernel_tensors()]
    opt.zero_grad(set_to_none=False)
    out = R.render_rays(rays, **kw)
    gs = torch.autograd.grad(R.img2mse(out["rgb_map"], tgt) + R.img2mse(out["rgb0"], tgt), params)
    assert all(g is not None for g in gs) and float(opt.flat_grad.abs().max()) == 0.0
    flat = torch.cat([(dict(zip(map(id, params), gs)).get(id(p), torch.zeros_like(p))).reshape(-1) for p in opt.params])
    assert float((flat - g_mrg).abs().max()) <= 1e-6 * float(g_mrg.abs().max())


def test_img2mse_fused_kernel(dev):
    """img2mse (H:9) as one kernel: value and both gradients against ATen's expression, any shape; broadcasting inputs
    take the reference's expression."""
    from consistentnerf_amd import run_nerf as R
    rs = np.random.RandomState(3)
    for shape in ((4096, 3), (4096,), (7, 5, 3), (1,), (378, 504, 3), (65537,)):     # (the last two: the two-stage whole-image form)
        x = T(rs.uniform(size=shape).astype(np.float32), dev).requires_grad_(True)
        y = T(rs.uniform(size=shape).astype(np.float32), dev).requires_grad_(True)
        l = R.img2mse(x, y)
        (3.0 * l).backward()
        xr, yr = x.detach().clone().requires_grad_(True), y.detach().clone().requires_grad_(True)
        lr = torch.mean((xr - yr) ** 2)
        (3.0 * lr).backward()
        assert l.shape == lr.shape and abs(l.item() - lr.item()) <= 2e-7 * max(lr.item(), 1e-30) + 1e-12
        tol = 4e-7 * float(xr.grad.abs().max())     # (2/n) (x - y) g: the same three factors, associated differently
        check(x.grad, xr.grad, tol, f"d img2mse / dx {shape}"); check(y.grad, yr.grad, tol, f"d img2mse / dy {shape}")
    a, b = torch.rand(5, 3, device=dev), torch.rand(3, device=dev)
    assert torch.equal(R.img2mse(a, b), torch.mean((a - b) ** 2))


def test_packed_weights_double_buffer(dev):
    """The kernel-layout weights a forward pass used survive ONE parameter update before its backward (two buffers
    alternate, no per-step clone); after two updates the backward refuses instead of differentiating through overwritten
    weights (the reference fails there too: autograd's version check)."""
    from consistentnerf_amd import ops, run_nerf as R
    from consistentnerf_amd.optim import FusedAdam
    coarse, _ = make_model(4, 128, True, 5, 93, dev)
    fine, _ = make_model(4, 128, True, 5, 94, dev)
    kw = _kwargs(coarse, fine, 16, 16, 0.0, False, 0.0, False)
    rays, tgt = T(I.ray_batch(40, seed=4), dev), torch.rand(40, 3, device=dev)
    opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=1e-2)

    def fwd():
        o = R.render_rays(rays, **kw)
        return R.img2mse(o["rgb_map"], tgt) + R.img2mse(o["rgb0"], tgt)
    opt.zero_grad()
    fwd().backward()
    ref = opt.flat_grad.clone()
    l = fwd()                     # forward with the current weights ...
    opt.step()                    # ... one update (re-packed into the OTHER buffer by the next forward) ...
    with torch.no_grad():
        R.render_rays(rays, **kw)
    opt.zero_grad()
    l.backward()                  # ... and the backward still sees the weights its forward used
    assert torch.equal(opt.flat_grad, ref)
    l = fwd()
    for _ in range(2):
        opt.step()
        with torch.no_grad():
            R.render_rays(rays, **kw)
    with pytest.raises(ops.CnerfError):
        l.backward()


# ------------------------------------------------------------------------------------------------
# round 2: BASELINE configs[4] (C5) and configs[2] (C3) at FULL size, through size-independent properties
def test_c5_full_frame_properties(dev):
    """One 756x1008 NDC frame (C5: 762 048 rays x (64 + 192) samples, D=8/W=256, perturb 0) through render(c2w=...):
    finite, chunk-invariant (R:79-80) bit for bit, and equal to the frame assembled from a 3-way row split through
    render(rays=...) (what three ranks of distributed.render_path_sharded compute)."""
    from consistentnerf_amd import distributed as D, run_nerf as R
    from consistentnerf_amd.run_nerf_helpers import get_rays
    H, W, focal = 756, 1008, 815.0
    coarse, fine, _ = _c2(dev, 1)
    kw = _kwargs(coarse, fine, 64, 128, 0.0, False, 0.0, False)
    kw.pop("lindisp")
    kw.update(ndc=True, near=0.0, far=1.0, use_viewdirs=True)
    K = I.intrinsics(H, W, focal)
    c2w = T(I.camera_pose(5.0, 0.0, 4.0), dev)
    with torch.no_grad():
        rgb, disp, acc, ex = R.render(H, W, K, chunk=32768, c2w=c2w, **kw)
        rgb2, disp2, acc2, _ = R.render(H, W, K, chunk=131072, c2w=c2w, **kw)
    assert rgb.shape == (H, W, 3) and disp.shape == (H, W)
    assert torch.isfinite(rgb).all() and torch.isfinite(acc).all() and float(rgb.min()) >= 0.0 and float(rgb.max()) <= 1.0
    assert float(acc.min()) >= 0.0 and float(acc.max()) <= 1.0 + 1e-5
    assert torch.equal(rgb, rgb2) and torch.equal(acc, acc2) and torch.equal(torch.nan_to_num(disp), torch.nan_to_num(disp2))
    assert float(rgb.std()) > 1e-3, "a random-init net still has to produce a non-constant image"
    ro, rd = get_rays(H, W, K, c2w[:3, :4])
    parts = []
    for r in range(3):
        lo, hi, idx = D.row_block(H, r, 3)
        with torch.no_grad():
            out = R.render(H, W, K, chunk=32768, rays=torch.stack([ro[idx.to(dev)], rd[idx.to(dev)]], 0), **kw)
        parts.append(out[0][:hi - lo])
    assert torch.equal(torch.cat(parts, 0), rgb)


def test_c3_full_step_is_additive_over_half_batches(dev):
    """The C3 training step at full size (5120 rays, 64 + 128 samples, D=8/W=256: hard-mask rgb + depth losses on both
    levels + the monocular patch term on the first 4 x 256 rays, through FusedAdam's flat gradient): the gradient of the
    whole batch equals the sum of the gradients of its two halves when each half normalises by the GLOBAL mask counts
    (what two ranks compute before the all-reduce; SURVEY hard part 7) — up to fp32 summation order."""
    from consistentnerf_amd import distributed as D, run_nerf_view as V
    from consistentnerf_amd.optim import FusedAdam
    B, far = 5120, 12.0
    coarse, fine, _ = _c2(dev, 1)
    rays = T(I.ray_batch(B, seed=17, near=1.2, far=far), dev)
    kw = _kwargs(coarse, fine, 64, 128, 0.0, False, 0.0, False)
    opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
    rs = np.random.RandomState(9)
    target = T(rs.uniform(size=(B, 3)).astype(np.float32), dev)
    prior = T(rs.uniform(1.2, far, size=(B,)).astype(np.float32), dev)
    mask = T((rs.uniform(size=(B,)) < 0.55).astype(np.float32), dev)
    mono = T(rs.uniform(0.05, 1.0, size=(1024,)).astype(np.float32), dev)
    counts = D.global_mask_counts(mask)

    def accumulate(sl, counts_, patch):
        out = V.render_rays(rays[sl], **kw)
        il, dl = V.hardmask_losses(out["rgb_map"], target[sl], mask[sl], 0.2, out["depth_map"], prior[sl], far, counts=counts_)
        il0, dl0 = V.hardmask_losses(out["rgb0"], target[sl], mask[sl], 0.2, out["depth0"], prior[sl], far, counts=counts_)
        loss = il + il0 + 0.1 * (dl + dl0)
        if patch:
            loss = loss + 0.001 * (V.midas_patch_loss(out["depth_map"], mono, 4, 16) + V.midas_patch_loss(out["depth0"], mono, 4, 16))
        loss.backward()
        return loss.item()
    opt.zero_grad()
    l_full = accumulate(slice(0, B), None, True)
    g_full = opt.flat_grad.clone()
    opt.zero_grad()
    l_a = accumulate(slice(0, B // 2), counts, True)       # the patch rays are the first 1024: they sit in the first half
    l_b = accumulate(slice(B // 2, B), counts, False)
    g_sum = opt.flat_grad.clone()
    assert np.isfinite(l_full) and abs((l_a + l_b) - l_full) <= 1e-5 * abs(l_full)
    scale = float(g_full.abs().max())
    assert scale > 0 and float((g_sum - g_full).abs().max()) <= 2e-5 * scale


@pytest.mark.parametrize("ndc,perturb,noise,white", [(False, 0.0, 0.0, False), (True, 0.0, 0.0, False), (False, 1.0, 1.0, True),
                                                      (True, 1.0, 0.0, False)])
def test_camera_path_generates_rays_in_kernel(dev, ndc, perturb, noise, white):
    """render(c2w=...) for inference (SURVEY 8 f-2 last clause): the rays of the camera are generated inside coarse_z / the
    MLP forward / compositing (cnerf_render_fwd_cam, one C call per chunk) instead of being written to HBM as an
    [H*W, 11] tensor first — bit-identical to the ray-tensor path on every output, NDC on and off, with the reference's
    deterministic jitter / noise hooks, ragged last chunk; the R and V surfaces; with autograd on, the ray-tensor path."""
    from consistentnerf_amd import ops, run_nerf as R, run_nerf_view as V
    g = golden("render_full_tiny")
    K, c2w = g["K"], T(g["c2w"], dev)
    coarse, _ = make_model(4, 128, True, 5, 31, dev)
    fine, _ = make_model(4, 128, True, 5, 32, dev)
    H, W = 13, 16
    kw = _kwargs(coarse, fine, 16, 16, perturb, white, noise, False)
    if ndc:
        kw.pop("lindisp")
    near, far = (0.0, 1.0) if ndc else (2.0, 6.0)
    kw.update(ndc=ndc, near=near, far=far, use_viewdirs=True, pytest=True, retraw=True)

    def run(stock, mod):
        kw["network_query_fn"]._cnerf_stock = stock
        ops.PROFILE = []
        try:
            with torch.no_grad():
                out = mod.render(H, W, K, chunk=100, c2w=c2w, **kw)
            kinds = [n for n, *_ in ops.PROFILE]
        finally:
            ops.PROFILE = None
        return out, kinds
    for mod in (R, V):
        ref, k_ref = run(False, mod)
        got, k_got = run(True, mod)
        assert "render_fwd_cam" in k_got and "mlp_fwd" not in k_got and "render_fwd_cam" not in k_ref
        assert k_got.count("render_fwd_cam") == 3      # 208 rays in chunks of 100
        for a, b in zip(got[:-1], ref[:-1]):
            assert a.shape == b.shape and torch.equal(torch.nan_to_num(a), torch.nan_to_num(b))
        assert set(got[-1]) == set(ref[-1])
        for k in ref[-1]:
            assert torch.equal(torch.nan_to_num(got[-1][k]), torch.nan_to_num(ref[-1][k])), k
    kw["network_query_fn"]._cnerf_stock = True
    ops.PROFILE = []
    try:
        out = R.render(H, W, K, chunk=100, c2w=c2w, **kw)       # autograd on: a graph is needed -> the ray-tensor path
        kinds = [n for n, *_ in ops.PROFILE]
    finally:
        ops.PROFILE = None
    assert "render_fwd_cam" not in kinds and out[0].requires_grad


# ------------------------------------------------------------------------------------------------
# round 2: OPT-IN reduced-precision inference (bf16 planes) — its own tolerance tier, never the default path
@pytest.mark.parametrize("D,W,tag", [(8, 256, "mlp_D8W256_vd"), (4, 128, "mlp_D4W128_vd")])
def test_bf16_plane_inference_forward(dev, D, W, tag):
    """cnerf_mlp_fwd_bf against the reference capture (fixture mlp_*: the reference's own network on the same points) and
    the exact-fp32 kernel, per plane count.  Tiers (relative to max|raw|): bf16x3 2e-5 (fp32-like: 6 of the 9 cross terms of
    a 3-way split), bf16x2 2e-3, bf16 1e-1.  The errors must also ORDER — each extra plane buys more than 10x — which a
    mis-paired plane or a slip in the k permutation of the panels would break.  Ragged point count (padding lanes)."""
    from consistentnerf_amd import ops
    g = golden(tag)
    model, _ = make_model(D, W, True, 4, 11, dev)
    spec = model.spec()
    pts, dirs = T(g["pts"], dev).reshape(-1, 3).contiguous(), T(g["dirs"], dev)
    M = pts.shape[0]
    dirs = dirs if dirs.shape[0] == M else dirs[:, None, :].expand(-1, M // dirs.shape[0], -1).reshape(-1, 3).contiguous()
    ref, _ = ops.mlp_forward(spec, ops.pack_weights(spec, model.kernel_tensors()), M, 1, pts=pts, dirs=dirs)
    cap = T(g["raw"], dev).reshape(M, 1, 4)
    scale = max(1.0, float(cap.abs().max()))
    check(ref, cap, 3e-5 * scale, "fp32 kernel vs capture")
    errs = {}
    for name, planes in (("bf16", 1), ("bf16x2", 2), ("bf16x3", 3)):
        pk = ops.pack_weights_bf(spec, model.kernel_tensors(), planes)
        # full, ragged (padding lanes of the last 32-point tile) and a size that leaves whole waves of the last workgroup
        # without points (they still move their share of the panel stream and meet the barriers)
        for Mr in (M, M - 13, 300):
            raw = ops.mlp_forward_bf(spec, pk, planes, Mr, 1, pts=pts[:Mr].contiguous(), dirs=dirs[:Mr].contiguous())
            assert torch.isfinite(raw).all()
            errs[name] = max(errs.get(name, 0.0), float((raw - cap[:Mr]).abs().max()) / scale)
            # the default kernel (four waves of a workgroup share the panel stream through an LDS ring) against the per-wave one
            # (CNERF_BF_PERWAVE=1: every wave streams the panels itself): the same arithmetic in the same order
            for force in ("1",):
                os.environ["CNERF_BF_PERWAVE"] = force
                try:
                    raw_v = ops.mlp_forward_bf(spec, pk, planes, Mr, 1, pts=pts[:Mr].contiguous(), dirs=dirs[:Mr].contiguous())
                finally:
                    del os.environ["CNERF_BF_PERWAVE"]
                assert torch.equal(raw, raw_v), (name, Mr, force, float((raw - raw_v).abs().max()))
        print(f"  {name}: max|d raw| / max|raw| vs the capture = {errs[name]:.3e}")
    # repeated launches of the shared-panel kernel (LDS-DMA ring, one barrier per K-step) on a batch that fills the chip stay
    # bit-identical to the per-wave kernel: a publish / reuse race in the ring would show here
    rs = np.random.RandomState(5)
    big = T(rs.uniform(-2, 2, size=(40000, 3)).astype(np.float32), dev)
    bdirs = T(rs.normal(size=(40000, 3)).astype(np.float32), dev)
    for planes in (1, 2, 3):
        pk = ops.pack_weights_bf(spec, model.kernel_tensors(), planes)
        os.environ["CNERF_BF_PERWAVE"] = "1"
        try:
            want = ops.mlp_forward_bf(spec, pk, planes, 40000, 1, pts=big, dirs=bdirs)
        finally:
            del os.environ["CNERF_BF_PERWAVE"]
        for rep in range(8):
            got = ops.mlp_forward_bf(spec, pk, planes, 40000, 1, pts=big, dirs=bdirs)
            assert torch.equal(got, want), (planes, rep, float((got - want).abs().max()))
    assert errs["bf16x3"] <= 2e-5 and errs["bf16x2"] <= 2e-3 and errs["bf16"] <= 1e-1
    assert errs["bf16"] > 10 * errs["bf16x2"] and (errs["bf16x2"] > 10 * errs["bf16x3"] or errs["bf16x3"] < 5e-6)


@fp32_only
def test_bf16_plane_inference_is_opt_in_and_never_trains(dev):
    """NeRF.inference_precision routes ONLY graph-free forwards to the bf16 kernel: the default is the exact path, a
    forward that may need gradients stays fp32, render() under no_grad uses it for both levels (ragged chunk, rays from
    a ray tensor), and the rendered image stays within the tier's PSNR bound of the fp32 render."""
    from consistentnerf_amd import ops, run_nerf as R
    coarse, fine, rays = _c2(dev, 777)
    kw = _kwargs(coarse, fine, 64, 128, 0.0, False, 0.0, False)

    def kinds_of(fn):
        ops.PROFILE = []
        try:
            out = fn()
            return out, [n for n, *_ in ops.PROFILE]
        finally:
            ops.PROFILE = None
    with torch.no_grad():
        ref, k0 = kinds_of(lambda: R.render_rays(rays, _with_depth=True, **kw))
    assert k0.count("mlp_fwd") == 2
    # PSNR (dB) of the rendered colours vs the fp32 render; measured on these random-init gain-1 nets (whose 2^9-frequency
    # encodings amplify input-side differences ~1e3x): 48.7 / 68.5 / 86.4
    floor = {"bf16": 40.0, "bf16x2": 60.0, "bf16x3": 80.0}
    try:
        for prec in ("bf16", "bf16x2", "bf16x3"):
            coarse.inference_precision = fine.inference_precision = prec
            with torch.no_grad():
                out, k = kinds_of(lambda: R.render_rays(rays, _with_depth=True, **kw))
            assert k.count("mlp_fwd_bf%d" % ops.PRECISION_PLANES[prec]) == 2 and "mlp_fwd" not in k
            mse = float(((out["rgb_map"] - ref["rgb_map"]) ** 2).mean())
            psnr = 99.0 if mse == 0 else -10.0 * np.log10(mse)
            print(f"  {prec}: rendered rgb vs fp32 render {psnr:.1f} dB; max|d depth| {float((out['depth_map'] - ref['depth_map']).abs().max()):.2e}")
            assert psnr >= floor[prec]
            out, k = kinds_of(lambda: R.render_rays(rays, **kw))        # autograd on: the exact kernels, with a stash
            assert k.count("mlp_fwd_train") == 2 and not any(n.startswith("mlp_fwd_bf") for n in k)
        with pytest.raises(ValueError):
            coarse.inference_precision = "fp8"
            with torch.no_grad():
                R.render_rays(rays, **kw)
    finally:
        coarse.inference_precision = fine.inference_precision = "fp32"


DIST2_GPU_WORKER = r"""
import os, sys, tempfile, numpy as np, torch
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _inputs as I
from test_gpu_parity import _view_args
from consistentnerf_amd import distributed as D, run_nerf_view as V
import torch.distributed as dist
dev = torch.device("cuda:0")                       # both ranks share the box's one GPU; the exchange goes through gloo
rank, world, _ = D.init_from_env("gloo")
assert world == 2 and dist.get_backend() == "gloo"

def run(sharded):
    with tempfile.TemporaryDirectory() as tmp:
        kw, _, start, grad_vars, opt = V.create_nerf(_view_args(tmp))
    for net, seed in ((kw["network_fn"], 51), (kw["network_fine"], 52)):
        net.load_state_dict({k: torch.from_numpy(v) for k, v in I.nerf_state_dict(4, 128, 10, 4, 5, True, seed=seed, gain=0.6).items()})
    kw.update(near=2.0, far=6.0, perturb=0.0)       # deterministic sampling: the sharded and the whole batch see the same rays
    opt.param_groups[0]["clip_value"] = 0.1
    red = D.GradReducer(opt, [kw["network_fn"], kw["network_fine"]], mean=False) if sharded else None
    K = I.intrinsics(100, 100, 138.0)
    losses = []
    for i in range(3):
        rays = torch.from_numpy(I.ray_batch(512, seed=300 + i)).to(dev)
        rs = np.random.RandomState(400 + i)
        target = torch.from_numpy(rs.uniform(size=(512, 3)).astype(np.float32)).to(dev)
        prior = torch.from_numpy(rs.uniform(2, 6, size=(512,)).astype(np.float32)).to(dev)
        mask = torch.from_numpy((rs.uniform(size=(512,)) < 0.6).astype(np.float32)).to(dev)
        counts = None
        if sharded:
            rays, target, prior, mask = D.shard_batch(rays, target, prior, mask)
            counts = D.global_mask_counts(mask)          # all-reduced: the GLOBAL set sizes
        rgb, disp, acc, depth, ex = V.render(100, 100, K, chunk=32768, rays=torch.stack([rays[:, 0:3], rays[:, 3:6]], 0),
                                             retraw=True, **kw)
        opt.zero_grad()
        il, dl = V.hardmask_losses(rgb, target, mask, 0.2, depth, prior, 6.0, counts=counts)
        il0, dl0 = V.hardmask_losses(ex["rgb0"], target, mask, 0.2, ex["depth0"], prior, 6.0, counts=counts)
        loss = il + dl + il0 + dl0
        loss.backward()
        if sharded:
            red.finish()
            loss = D.allreduce_scalar_sum(loss.detach())
        opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    return opt.flat_param.clone(), losses

p_sh, l_sh = run(True)           # 2 ranks x 256 rays, flat gradient all-reduced per network slice
p_1, l_1 = run(False)            # the same 512-ray steps on one rank (every rank computes it: the reference trajectory)
# replicas stay identical
chk = p_sh.clone(); dist.broadcast(chk, 0)
assert torch.equal(chk, p_sh), "ranks diverged"
rel_l = max(abs(a - b) / abs(b) for a, b in zip(l_sh, l_1))
# Adam normalises the step: a gradient element near zero whose sign differs by summation order moves a weight by ~lr, so
# the bound on the weights is a few lr (3 steps of 5e-4), the bound on the losses is tight
dmax = float((p_sh - p_1).abs().max())
frac = float(((p_sh - p_1).abs() > 1e-6).float().mean())
assert rel_l < 1e-5, (l_sh, l_1)
assert dmax < 2e-3 and frac < 0.02, (dmax, frac)

# ---- the in-loop consistency step (a15, VT:899-969) sharded over the two ranks: run_nerf_view.ss_step_loss(group=...) runs the two
# batch-global exchanges itself (all-reduce MIN of the minimum |z - D_ref| for the threshold-doubling rule, all-reduce SUM of the
# three ray counts the masked means divide by); the per-rank losses add up, the flat gradient is SUMMED (GradReducer(mean=False))
from test_gpu_parity import _ss_scene

def run_ss(sharded):
    sc = _ss_scene(dev, 1024, seed=7, owned=True)
    opt, kw = sc["opt"], sc["kw"]
    red = D.GradReducer(opt, [sc["coarse"], sc["fine"]], mean=False) if sharded else None
    img, dep = torch.from_numpy(sc["g"]["images"][1]).to(dev), torch.from_numpy(sc["g"]["depths"][1]).to(dev)
    losses, thr = [], []
    for i in range(2):
        rays, tgt = sc["rays"], sc["tgt"]
        prior = sc["prior"] + 0.01 * i + torch.linspace(0.0, 0.05, 1024, device=dev)
        if sharded:
            lo, hi = D.shard_bounds(1024)
            rays, tgt, prior = rays[:, lo:hi].contiguous(), tgt[lo:hi], prior[lo:hi]
        loss, info = V.ss_step_loss(sc["H"], sc["W"], sc["K"], rays, tgt, prior, sc["poses"][1], img, dep, kw, chunk=4096,
                                    occlusion_threshold=3e-4, with_depth_loss=True, coins=(1, 1, 0, 1), route="one_render",
                                    group=dist.group.WORLD if sharded else None)
        opt.zero_grad()
        loss.backward()
        if sharded:
            red.finish()
            loss = D.allreduce_scalar_sum(loss.detach())
        opt.step()
        losses.append(float(loss))
        thr.append(float(V.ss_host_view(info)["threshold"]))
    torch.cuda.synchronize()
    return opt.flat_param.clone(), losses, thr

ps_sh, ls_sh, t_sh = run_ss(True)
ps_1, ls_1, t_1 = run_ss(False)
chk = ps_sh.clone(); dist.broadcast(chk, 0)
assert torch.equal(chk, ps_sh), "ranks diverged (ss step)"
assert t_sh == t_1 and t_1[0] > 3e-4, (t_sh, t_1)          # every rank applied the WHOLE batch's (doubled) threshold
rel_ss = max(abs(a - b) / abs(b) for a, b in zip(ls_sh, ls_1))
dmax_ss = float((ps_sh - ps_1).abs().max())
frac_ss = float(((ps_sh - ps_1).abs() > 1e-6).float().mean())
assert rel_ss < 1e-5, (ls_sh, ls_1)
assert dmax_ss < 2e-3 and frac_ss < 0.02, (dmax_ss, frac_ss)
D.barrier()
if rank == 0:
    print("DIST2_GPU_OK", l_sh, l_1, dmax, frac, "ss:", ls_sh, ls_1, dmax_ss, frac_ss)
dist.destroy_process_group()
"""


def test_two_ranks_product_step_on_one_gpu(dev, tmp_path):
    """The product path of the data-parallel step with WORLD SIZE 2: two processes (both on the box's one MI355X, the
    collectives through gloo, which stages device tensors through the host) shard each 512-ray batch, all-reduce the mask
    counts, render + backward through the HIP kernels into their FusedAdam.flat_grad, exchange it with GradReducer (one
    all-reduce per network slice, issued from inside loss.backward()) and step.  After 3 steps the replicas are identical,
    the summed losses equal the single-rank losses of the whole batch (1e-5), and the weights agree up to Adam's sensitivity
    to the summation order of near-zero gradient elements.  Round 6: the same for the in-loop consistency step (a15) —
    run_nerf_view.ss_step_loss(group=WORLD): global threshold (all-reduce MIN) + global ray counts (all-reduce SUM), summed flat
    gradient — two sharded steps == the two single-rank steps."""
    import subprocess
    import sys
    script = tmp_path / "dist2_gpu_worker.py"
    script.write_text(DIST2_GPU_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29591", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    env.pop("CNERF_FORCE_DIST", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
                        "127.0.0.1", "--master-port", "29591", str(script), root], capture_output=True, text=True, env=env,
                       timeout=900)
    print(r.stdout[-1500:])
    assert r.returncode == 0 and "DIST2_GPU_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_bench_gpus_2_launched_plainly_is_its_own_launcher(dev):
    """`python bench.py --gpus 2 ...` with NO torchrun around it (the shape of the driver's N = 1 command)
    re-executes itself under torch.distributed.run and still prints exactly one JSON line from rank 0.  Two ranks on the box's
    one MI355X over gloo (CNERF_DIST_BACKEND=gloo; the timing means nothing, the contract does): n_gpus 2, dist.ranks 2,
    exit status 0.  With RCCL and one GPU the same command answers with one JSON error line and a non-zero status."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "CNERF_FORCE_DIST")}
    env.update(CNERF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extra",
                        "--no-cpu-baseline", "--rays-per-gpu", "512"], capture_output=True, text=True, env=env, timeout=900)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, lines, r.stderr[-3000:])
    o = json.loads(lines[0])
    assert o["n_gpus"] == 2 and o["dist"]["ranks"] == 2 and o["dist"]["backend"] == "gloo" and o["value"] > 0
    assert o["config"]["rays_per_gpu"] == 512 and o["steps"] == 3
    if torch.cuda.device_count() < 2:
        env.pop("CNERF_DIST_BACKEND")
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                           capture_output=True, text=True, env=env, timeout=300)
        lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
        assert r.returncode != 0 and len(lines) == 1 and "error" in json.loads(lines[0]), (r.returncode, lines)
        assert "Traceback" not in r.stderr


def test_bench_default_line_is_small_and_complete(dev, tmp_path):
    """VERDICT r05 item 1 on the REAL line: `python bench.py --steps 3 --warmup 1` (every leg, the PMC passes, the CPU baseline —
    the shape of the driver's N = 1 command) prints exactly one stdout line of at most bench.LINE_LIMIT (< 6 KB) bytes that carries
    the contract keys, `roofline` (bound, achieved, peak, frac, traffic, the kernel table, whole_step_frac), `cpu_baseline` and the
    flat leg_* scalars; the legs' detail is in the side file the line names."""
    import json
    import subprocess
    import sys
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "CNERF_FORCE_DIST", "CNERF_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1"], capture_output=True, text=True,
                       env=env, timeout=1500, cwd=str(tmp_path))
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, [ln[:200] for ln in lines], r.stderr[-3000:])
    assert len(lines[0].encode()) <= bench.LINE_LIMIT < 6144, len(lines[0])
    o = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "detail"):
        assert k in o, k
    assert o["n_gpus"] == 1 and o["steps"] == 3 and o["value"] > 1e6 and o["dtype"] == "f32" and "extra" not in o
    rf = o["roofline"]
    assert rf["bound"] == "mfma" and rf["peak"] == bench.PEAK_FP32_MFMA_TFLOPS and 0.3 < rf["frac"] <= 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and len(rf["kernels"]) == 4 and 0.3 < rf["whole_step_frac"] <= 1.0
    assert rf["traffic"] is None or rf["traffic"] > 1e9
    cb = o["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["sample"]
    cfg = o["config"]
    for k in ("leg_c4_shard_ms_per_step_graph", "leg_c5_frame_s", "leg_c3_ms_per_step", "leg_c3_ss_ms_per_step", "leg_c3_frac_of_peak"):
        assert isinstance(cfg.get(k), (int, float)), (k, cfg.get(k))
    detail = json.load(open(os.path.join(str(tmp_path), o["detail"])))
    assert detail["value"] == o["value"] and "hbm_kernels" in detail["extra"] and "c3_ss" in detail["extra"]


def test_bench_gpus_2_default_legs_line_first_then_the_side_leg(dev):
    """The shape of the driver's N > 1 command — no --no-extra: at world > 1 the contract line is printed BEFORE the C4 strong-shard
    side leg runs (nothing that leg does can void the scaling measurement), stdout still carries exactly one line, the leg's result
    follows on stderr as its own JSON object, and the exit status is the main leg's.  Two ranks over gloo on the one GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT",
                                                            "CNERF_FORCE_DIST")}
    env.update(CNERF_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", CNERF_BENCH_LEG_TIMEOUT="600")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=1200)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert r.returncode == 0 and len(lines) == 1, (r.returncode, lines, r.stderr[-3000:])
    o = json.loads(lines[0])
    assert o["n_gpus"] == 2 and o["dist"]["ranks"] == 2 and o["value"] > 0 and o["config"]["rays_per_gpu"] == 4096
    assert "extra" not in o or "c4_strong" not in o.get("extra", {})          # the line left before the leg started
    after = [json.loads(ln) for ln in r.stderr.splitlines() if ln.startswith('{"after_the_line"')]
    assert len(after) == 1, r.stderr[-3000:]
    leg = after[0]["after_the_line"]["c4_strong"]
    assert "error" not in leg and leg["rays_per_gpu"] == 2048 and leg["ms_per_step_eager"] > 0, leg


def test_graphed_step_equals_eager_steps(dev):
    """graph.GraphedStep: the whole training step (render of both levels, fused losses, the merged backward, FusedAdam with its
    scalars in device memory, weight packing) recorded once as a hipGraph and replayed — bit-identical to the same steps run
    eagerly, with the batch and the decayed lr changing from step to step.  The replays run WITHOUT any host synchronisation
    between them (the host enqueues all six while the GPU is still on the first): every step must still see ITS lr / bias
    corrections (they travel through a ring of pinned blocks uploaded stream-ordered ahead of each replay — a single pinned
    buffer read by a recorded copy would hand step N the scalars of step N+k)."""
    from consistentnerf_amd import run_nerf as R
    from consistentnerf_amd.graph import GraphedStep
    from consistentnerf_amd.optim import FusedAdam

    def build():
        coarse, _ = make_model(4, 128, True, 5, 93, dev)
        fine, _ = make_model(4, 128, True, 5, 94, dev)
        kw = _kwargs(coarse, fine, 16, 16, 0.0, False, 0.0, False)
        opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4, clip_value=0.1)

        def step_fn(rays, tgt):
            out = R.render_rays(rays, **kw)
            opt.zero_grad()
            loss = R.img2mse(out["rgb_map"], tgt) + R.img2mse(out["rgb0"], tgt)
            loss.backward()
            opt.step()
            return loss
        return opt, step_fn, kw
    nstep = 10                                       # more than FusedAdam.RING: the ring wraps while the GPU is behind
    batches = [(T(I.ray_batch(96, seed=40 + i), dev), torch.rand(96, 3, device=dev)) for i in range(nstep)]
    lrs = [5e-4 * (0.1 ** (i / 3.0)) for i in range(nstep)]
    # eager: 3 "warm-up" steps on batch 0 (what GraphedStep does before recording), then the steps
    opt_e, step_e, _ = build()
    opt_e.make_capturable()
    for _ in range(3):
        step_e(*batches[0])
    le = []
    for (rays, tgt), lr in zip(batches, lrs):
        opt_e.param_groups[0]["lr"] = lr
        le.append(step_e(rays, tgt).item())
    opt_g, step_g, _ = build()
    gs = GraphedStep(step_g, opt_g, batches[0], warmup=3)
    big = torch.empty(1 << 28, device=dev)
    lg = []
    big.normal_()                                    # ~a millisecond of queued GPU work: the host runs ahead of the replays
    for (rays, tgt), lr in zip(batches, lrs):
        opt_g.param_groups[0]["lr"] = lr
        lg.append(gs(rays, tgt).clone())             # (device-side copy of the static loss: no host sync)
    lg = [float(x) for x in lg]
    assert le == lg, (le, lg)
    assert torch.equal(opt_e.flat_param, opt_g.flat_param)
    assert opt_e._step == opt_g._step == 3 + nstep


def test_eval_render_between_graphed_steps_sees_the_new_weights(dev):
    """render -> replay -> render: a graph replay moves the weights behind Python's back (the recorded optimizer.step() ran
    its Python once, at capture), so GraphedStep bumps the parameters' epochs after every replay and an evaluation render
    between graphed steps re-packs — the second render must equal a render of an eagerly stepped twin, not the first one.
    Also with warmup=0 right after a render (the pack cached by that render must not be what the recording's forward reads)."""
    from consistentnerf_amd import run_nerf as R
    from consistentnerf_amd.graph import GraphedStep
    from consistentnerf_amd.optim import FusedAdam

    def build():
        coarse, _ = make_model(4, 128, True, 5, 95, dev)
        fine, _ = make_model(4, 128, True, 5, 96, dev)
        kw = _kwargs(coarse, fine, 16, 16, 0.0, False, 0.0, False)
        opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-3)

        def step_fn(rays, tgt):
            out = R.render_rays(rays, **kw)
            opt.zero_grad()
            loss = R.img2mse(out["rgb_map"], tgt) + R.img2mse(out["rgb0"], tgt)
            loss.backward()
            opt.step()
            return loss
        return opt, step_fn, kw
    rays, tgt = T(I.ray_batch(96, seed=50), dev), torch.rand(96, 3, device=dev)
    probe = T(I.ray_batch(64, seed=51), dev)

    def ev(kw):
        with torch.no_grad():
            return R.render_rays(probe, **kw)["rgb_map"].clone()
    for warm in (2, 0):
        opt_e, step_e, kw_e = build()
        opt_e.make_capturable()
        opt_g, step_g, kw_g = build()
        r0 = ev(kw_g)                                   # caches a pack of the initial weights
        gs = GraphedStep(step_g, opt_g, (rays, tgt), warmup=warm)
        for _ in range(warm):
            step_e(rays, tgt)
        assert torch.equal(ev(kw_e), ev(kw_g))
        renders = []
        for k in range(3):
            gs(rays, tgt)
            step_e(rays, tgt)
            a, b = ev(kw_g), ev(kw_e)
            assert torch.equal(a, b), f"warmup={warm}: eval render after replay {k} used stale packed weights"
            renders.append(a)
        assert not torch.equal(renders[0], r0) and not torch.equal(renders[1], renders[0]) and not torch.equal(renders[2], renders[1])
        assert torch.equal(opt_e.flat_param, opt_g.flat_param)


GRAPH_DIST_WORKER = r"""
import os, sys, numpy as np, torch
ROOT = sys.argv[1]; MODE = sys.argv[2]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _inputs as I
from test_gpu_parity import make_model, _kwargs, T
from consistentnerf_amd import distributed as D, run_nerf as R
from consistentnerf_amd.graph import GraphedStep
from consistentnerf_amd.optim import FusedAdam
import torch.distributed as dist
dev = torch.device("cuda:0")
backend = "gloo" if MODE == "gloo2" else "nccl"
rank, world, _ = D.init_from_env(backend)
assert dist.is_initialized() and world == (2 if MODE == "gloo2" else 1)
NB, NSTEP = 256, 5

def build():
    coarse, _ = make_model(4, 128, True, 5, 93, dev)
    fine, _ = make_model(4, 128, True, 5, 94, dev)
    kw = _kwargs(coarse, fine, 16, 16, 0.0, False, 0.0, False)
    opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4, clip_value=0.1)
    def fwd_bwd(rays, tgt):
        out = R.render_rays(rays, **kw)
        opt.zero_grad()
        loss = R.img2mse(out["rgb_map"], tgt) + R.img2mse(out["rgb0"], tgt)
        loss.backward()
        return loss
    return opt, fwd_bwd, (coarse, fine)

batches = [(T(I.ray_batch(NB, seed=40 + i), dev), torch.rand(NB, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(7 + i)))
           for i in range(NSTEP)]
lrs = [5e-4 * (0.1 ** (i / 3.0)) for i in range(NSTEP)]

# reference trajectory: the WHOLE batch on one rank, eager, no exchange (every rank computes it)
opt_1, f_1, _ = build()
for _ in range(2):
    f_1(*batches[0]); opt_1.step()
l_1 = []
for (rays, tgt), lr in zip(batches, lrs):
    opt_1.param_groups[0]["lr"] = lr
    l_1.append(float(f_1(rays, tgt))); opt_1.step()

res = {}
for collective in (("split",) if MODE == "gloo2" else ("split", "capture")):
    # strong sharding: each rank renders its B/world slice; mean losses of equal shards -> sum of gradients x 1/world
    opt_s, f_s, nets = build()
    red = D.GradReducer(opt_s, nets, mean=True, fold_scale=True)
    sh = [D.shard_batch(r, t) for r, t in batches]
    gs = GraphedStep(f_s, opt_s, sh[0], warmup=2, reducer=red, collective=collective)
    l_s = []
    for (rays, tgt), lr in zip(sh, lrs):
        opt_s.param_groups[0]["lr"] = lr
        l_s.append(gs(rays, tgt).clone())
    l_s = [float(D.allreduce_scalar_sum(x.detach()) / world) for x in l_s]
    chk = opt_s.flat_param.clone(); dist.broadcast(chk, 0)
    assert torch.equal(chk, opt_s.flat_param), "ranks diverged"
    assert opt_s._step == opt_1._step == 2 + NSTEP
    rel_l = max(abs(a - b) / abs(b) for a, b in zip(l_s, l_1))
    dmax = float((opt_s.flat_param - opt_1.flat_param).abs().max())
    frac = float(((opt_s.flat_param - opt_1.flat_param).abs() > 1e-6).float().mean())
    if world == 1:      # one rank: the exchange is an identity, the graphed sharded step IS the eager step, bit for bit
        assert l_s == l_1 and dmax == 0.0, (collective, l_s, l_1, dmax)
    else:               # summation order differs: the tolerance of test_two_ranks_product_step_on_one_gpu
        # (7 Adam steps of two differently-sharded runs: a trajectory comparison, measured 1e-4 in fp32 and 1.2e-4 in the
        #  bf16x3 arithmetic, whose point-range sums round differently per shard size)
        assert rel_l < (2e-4 if os.environ.get("CNERF_TRAIN_PRECISION", "fp32") == "bf16x3" else 1e-4), (l_s, l_1)
        assert dmax < 3e-3 and frac < 0.15, (dmax, frac)     # (7 Adam steps; measured 4.7e-4 / 0.066)
    res[collective] = (rel_l, dmax, frac)
D.barrier()
if rank == 0:
    print("GRAPH_DIST_OK", MODE, res)
dist.destroy_process_group()
"""


@pytest.mark.parametrize("mode", ["rccl1", "gloo2"])
def test_graphed_sharded_step_with_the_exchange(dev, tmp_path, mode):
    """GraphedStep with the gradient exchange (C4: the strong-scaling shard is launch-bound when stepped eagerly).
    rccl1 — a 1-rank RCCL group (CNERF_FORCE_DIST=1): both forms, two graphs around the eager all-reduce ("split") and the RCCL
    all-reduce recorded INSIDE one graph ("capture"), are bit-identical to the eager whole-batch steps.
    gloo2 — WORLD SIZE 2 on the box's one GPU (gloo stages through the host, so only "split" applies): each rank steps its
    half of every 256-ray batch through the graphs, 1/world folded into the Adam kernel; after 5 steps the replicas are
    identical, the losses (all downstream of 2 warm-up Adam steps on differently summed gradients) equal the single-rank
    whole-batch losses to 1e-4, and the weights agree up to Adam's sensitivity to the summation order
    (the tolerance of test_two_ranks_product_step_on_one_gpu).  Precedent for the semantics:
    RegNeRF/internal/utils.py:63-66 (shard), RegNeRF/train.py:246-274 (pmean of the gradient, then the optimizer)."""
    import subprocess
    import sys
    script = tmp_path / "graph_dist_worker.py"
    script.write_text(GRAPH_DIST_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if mode == "rccl1":
        env = dict(os.environ, CNERF_FORCE_DIST="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT="29579", HSA_ENABLE_IPC_MODE_LEGACY="0")
        cmd = [sys.executable, str(script), root, mode]
    else:
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29593", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
        env.pop("CNERF_FORCE_DIST", None)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
               "--master-port", "29593", str(script), root, mode]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    print(r.stdout[-1500:])
    assert r.returncode == 0 and "GRAPH_DIST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


def test_engine_query_fallback_gives_identical_gradients(dev, monkeypatch):
    """The default training route asks the autograd engine (a private torch symbol, probed at import) whether it accumulates
    into .grad and whether the coarse node runs in the same pass.  With the query unusable (`_ENGINE_QUERY = None`, what the
    import-time probe selects when the symbol is missing or misbehaves) the step takes the plain route — per-tensor gradients
    returned to autograd, one backward per level — and the flat gradient and the stepped weights are bit-identical."""
    from consistentnerf_amd import ops, run_nerf as R
    from consistentnerf_amd.optim import FusedAdam

    def run(disable):
        if disable:
            monkeypatch.setattr(R, "_ENGINE_QUERY", None)
        coarse, _ = make_model(4, 128, True, 5, 97, dev)
        fine, _ = make_model(4, 128, True, 5, 98, dev)
        kw = _kwargs(coarse, fine, 16, 16, 0.0, False, 0.0, False)
        opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
        rays, tgt = T(I.ray_batch(200, seed=60), dev), torch.rand(200, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
        kinds = []
        orig_pair, orig_one = ops.mlp_backward_pair, ops.mlp_backward
        monkeypatch.setattr(ops, "mlp_backward_pair", lambda *a, **k: (kinds.append("pair"), orig_pair(*a, **k))[1])
        monkeypatch.setattr(ops, "mlp_backward", lambda *a, **k: (kinds.append("one"), orig_one(*a, **k))[1])
        out = R.render_rays(rays, **kw)
        opt.zero_grad()
        (R.img2mse(out["rgb_map"], tgt) + R.img2mse(out["rgb0"], tgt)).backward()
        g = opt.flat_grad.clone()
        opt.step()
        monkeypatch.setattr(ops, "mlp_backward_pair", orig_pair)
        monkeypatch.setattr(ops, "mlp_backward", orig_one)
        return g, opt.flat_param.clone(), kinds
    g0, p0, k0 = run(False)
    g1, p1, k1 = run(True)
    assert k0 == ["pair"] and k1 == ["one", "one"], (k0, k1)
    assert torch.equal(g0, g1) and torch.equal(p0, p1)


# ------------------------------------------------------------------------------------------------
# round 3: oracle-grade gradient parity AT THE BENCHMARKED LAUNCH SIZE (4096 rays, merged cnerf_mlp_bwd_pair)
def _stash_blocks_dev(stash, M, D, W, vd, in_chp=64):
    """stash_blocks() for launches too large to move to the host: the same column map, fp32 views ON THE DEVICE (the caller
    converts what it needs to fp64), ReLU sign bits decoded with torch integer ops and checked against the stored
    activations."""
    Mp = (M + 31) // 32 * 32
    rows = stash.numel() // Mp
    full = stash.reshape(Mp // 32, rows // 8, 32, 8).permute(0, 2, 1, 3).reshape(Mp, rows)   # (a copy: tile-major -> row-major)
    if Mp > M:
        assert float(full[M:].abs().max()) == 0.0
    s = full[:M]
    out, r = {}, 0
    for name, n in ([("enc", in_chp)] + [(f"h{l}", W) for l in range(D)] + ([("feat", W), ("denc", 32), ("hv", W // 2)] if vd else [])):
        out[name] = s[:, r:r + n]
        r += n
    nt = W // 32
    md, mdv = (nt + 1) // 2, (nt // 2 + 1) // 2
    nmask = (D * 2 * md + (2 * mdv if vd else 0) + 7) // 8 * 8
    assert r + nmask == s.shape[1]
    words = s[:, r:].contiguous().view(torch.int32).to(torch.int64) & 0xffffffff

    def decode(col0, tiles, m_d):
        bits = torch.zeros((M, 32 * tiles), dtype=torch.bool, device=stash.device)
        for hh in range(2):
            for d in range(m_d):
                wd = words[:, col0 + hh * m_d + d]
                pos = 31
                for par in range(2):
                    for t in range(2 * d, min(2 * d + 2, tiles)):
                        for rr in range(par, 16, 2):
                            bits[:, 32 * t + 8 * (rr >> 2) + 4 * hh + (rr & 3)] = ((wd >> pos) & 1).bool()
                            pos -= 1
        return bits
    for l in range(D):
        assert torch.equal(decode(l * 2 * md, nt, md), out[f"h{l}"] > 0), f"sign bits of layer {l}"
    if vd:
        assert torch.equal(decode(D * 2 * md, max(nt // 2, 1), mdv), out["hv"] > 0), "sign bits (view branch)"
    return out


def _fp64_replay_dev(model, blk, g_raw, D=8, W=256):
    """dgrad + wgrad of NeRF.forward (H:107-130) in fp64 with torch ON THE GPU from the kernel's own stashed activations (so
    the ReLU masks are the kernel's by construction), D=8/W=256/viewdirs -> {parameter name: gradient}."""
    Wd = {k: v.detach().double() for k, v in model.state_dict().items()}
    Gd = g_raw.reshape(-1, 4).double()
    d_rgb, d_sig = Gd[:, :3], Gd[:, 3:4]
    f = lambda k: blk[k].double()   # noqa: E731
    ref = {}
    hv = f("hv")
    dZv = (d_rgb @ Wd["rgb_linear.weight"]) * (hv > 0)
    ref["rgb_linear.weight"], ref["rgb_linear.bias"] = d_rgb.t() @ hv, d_rgb.sum(0)
    del hv
    vin = torch.cat([f("feat"), f("denc")[:, :27]], 1)
    ref["views_linears.0.weight"], ref["views_linears.0.bias"] = dZv.t() @ vin, dZv.sum(0)
    del vin
    dF = dZv @ Wd["views_linears.0.weight"][:, :W]
    hl = f(f"h{D-1}")
    ref["feature_linear.weight"], ref["feature_linear.bias"] = dF.t() @ hl, dF.sum(0)
    ref["alpha_linear.weight"], ref["alpha_linear.bias"] = d_sig.t() @ hl, d_sig.sum(0)
    dZ = (dF @ Wd["feature_linear.weight"] + d_sig @ Wd["alpha_linear.weight"]) * (hl > 0)
    del hl, dF
    enc = f("enc")[:, :63]
    for l in range(D - 1, 0, -1):
        skip_layer = l == 5
        hp = f(f"h{l-1}")
        inp = torch.cat([enc, hp], 1) if skip_layer else hp
        ref[f"pts_linears.{l}.weight"], ref[f"pts_linears.{l}.bias"] = dZ.t() @ inp, dZ.sum(0)
        Wl = Wd[f"pts_linears.{l}.weight"]
        dZ = (dZ @ (Wl[:, 63:] if skip_layer else Wl)) * (hp > 0)
        del hp, inp
    ref["pts_linears.0.weight"], ref["pts_linears.0.bias"] = dZ.t() @ enc, dZ.sum(0)
    return ref


def test_c2_full_size_backward_exact(dev, monkeypatch):
    """ONE C2 training step exactly as bench.py times it — 4096 rays, coarse 64 + fine 192 samples, D=8/W=256, both networks
    FusedAdam-owned, so loss.backward() runs the MERGED cnerf_mlp_bwd_pair (one dgrad grid over M = 786 432 + 262 144 points,
    one wgrad grid of 128 + 64 point ranges x 28 GEMM jobs, one fixed-order reduction of the partials) accumulating into the
    flat gradient.  Checked at THAT launch size:
      * every weight / bias gradient against an fp64 replay of dgrad + wgrad from the kernel's own stashes (identical ReLU
        masks by construction), torch fp64 on the GPU:  <= 1e-5 * max|g| per tensor;
      * the merged launch bit-identical to two separate cnerf_mlp_bwd launches on the same inputs;
      * the forward of the same launch on a 512-ray sub-slice against the CPU oracle (rgb0 / depth0 2e-5; raw of the coarse
        level 3e-5 * max|raw|; the fine level with the oracle evaluated at the kernel's own sample depths).
    Reference: run_nerf_helpers.py:107-130 (NeRF.forward), run_nerf.py:265-308 (raw2outputs)."""
    from consistentnerf_amd import ops, run_nerf as R, run_nerf_view as V
    from consistentnerf_amd.optim import FusedAdam
    coarse, fine, rays = _c2(dev)
    kw = _kwargs(coarse, fine, 64, 128, 1.0, False, 0.0, False)
    opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
    tgt = torch.rand(4096, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(3))
    seen = {}
    orig = ops.mlp_backward_pair

    def spy(*a, **k):
        seen["args"], seen["kw"] = a, k
        return orig(*a, **k)
    monkeypatch.setattr(ops, "mlp_backward_pair", spy)
    torch.manual_seed(11)
    out = V.render_rays(rays, retraw=True, _debug=True, **kw)
    opt.zero_grad()
    loss = R.img2mse(out["rgb_map"], tgt) + R.img2mse(out["rgb0"], tgt)
    loss.backward()
    torch.cuda.synchronize()
    assert "args" in seen, "the step did not take the merged coarse+fine backward"
    fs, fp, fg, fB, fS, fst, fgr, cs, cp, cg, cB, cS, cst, cgr = seen["args"]
    assert (fB, fS, cB, cS) == (4096, 192, 4096, 64)
    names = [n for n, _ in I.nerf_param_shapes(8, 256, 63, 27, 5, True)][3:]
    worst = {}
    for tag, model, st, g_raw, M, got in (("fine", fine, fst, fg, fB * fS, fgr), ("coarse", coarse, cst, cg, cB * cS, cgr)):
        blk = _stash_blocks_dev(st, M, 8, 256, True)
        ref = _fp64_replay_dev(model, blk, g_raw)
        del blk
        for n, g in zip(names, got):
            if n not in ref:
                continue
            r = ref[n].reshape(g.shape)
            d = float((g.double() - r).abs().max() / r.abs().max().clamp(min=1e-30))
            worst[f"{tag}.{n}"] = d
        del ref
        torch.cuda.empty_cache()
    w = max(worst, key=worst.get)
    print(f"  fp64 replay at M = 786432 + 262144 (merged grid): worst rel-max diff {worst[w]:.3e} ({w})")
    for n, d in worst.items():
        assert d <= 1e-5, f"{n}: rel max diff {d:.3e} vs the fp64 replay"
    # merged == two separate launches, bit for bit, at this size
    sep_f = ops.mlp_backward(fs, fp, fg, fB, fS, fst, packed_bf=seen["kw"].get("packed_bf0"))     # (None: exact fp32; else the
    sep_c = ops.mlp_backward(cs, cp, cg, cB, cS, cst, packed_bf=seen["kw"].get("packed_bf1"))     #  same bf16x3 kernels, unmerged)
    for a, b in zip(list(fgr) + list(cgr), sep_f + sep_c):
        assert torch.equal(a, b.reshape(a.shape)), "merged backward differs from the separate launches"
    # forward of the SAME launch, a 512-ray sub-slice, against the CPU oracle
    sl = slice(2048, 2560)
    sdc = O.as_tensors(I.nerf_state_dict(8, 256, 10, 4, 5, True, seed=21))
    sdf = O.as_tensors(I.nerf_state_dict(8, 256, 10, 4, 5, True, seed=22))
    net = O.NetCfg(8, 256, output_ch=5)
    r_c = rays[sl].cpu()
    z_c, z_f = out["_z_coarse"][sl].cpu(), out["_z_vals"][sl].cpu()
    with torch.no_grad():
        pts = r_c[:, None, 0:3] + r_c[:, None, 3:6] * z_c[..., None]
        raw_c = O.query(sdc, pts, r_c[:, 8:11], net)
        rgb0, disp0, acc0, w0, depth0 = O.composite(raw_c, z_c, r_c[:, 3:6])
        pts = r_c[:, None, 0:3] + r_c[:, None, 3:6] * z_f[..., None]
        raw_f = O.query(sdf, pts, r_c[:, 8:11], net)
        rgb1, disp1, acc1, w1, depth1 = O.composite(raw_f, z_f, r_c[:, 3:6])
    check(out["rgb0"][sl], rgb0, 2e-5, "4096-ray launch, slice: rgb0 vs oracle")
    check(out["depth0"][sl], depth0, 2e-5 * 4.67, "4096-ray launch, slice: depth0 vs oracle")
    check(out["raw"][sl], raw_f[..., :4], 3e-5 * max(1.0, float(raw_f.abs().max())), "4096-ray launch, slice: fine raw vs oracle at the kernel's depths")
    check(out["rgb_map"][sl], rgb1, 2e-5, "4096-ray launch, slice: rgb_map vs oracle at the kernel's depths")
    check(out["depth_map"][sl], depth1, 2e-5 * 4.67, "4096-ray launch, slice: depth_map vs oracle at the kernel's depths")


def test_c4_eight_way_shard_equals_the_whole_batch_step(dev):
    """C4 at its true sizes, the arithmetic of the 8-GPU step done serially on one GPU: the 4096-ray C2 batch cut into 8 contiguous
    shards of 512 rays (distributed.shard_bounds), each shard's step — render, mean losses over ITS rays, backward through the
    merged cnerf_mlp_bwd_pair at M = 98 304 + 32 768 — accumulated into the flat gradient (what the all-reduce sums), then ONE
    FusedAdam step with grad_scale = 1/8 (GradReducer(fold_scale=True).grad_scale at world 8).  Against the single 4096-ray step:
    the averaged gradient to 5e-6 of its max (measured 2e-7; only the summation order of fp32 partials differs: 64 + 32 point
    ranges per shard vs 64 + 64 for the whole batch), the loss to 1e-6, the stepped weights to 1e-4 with < 1 % of them more than
    1e-7 apart (measured: max 1.3e-6, none) — Adam's sign normalisation only amplifies gradient elements within round-off of zero.  Reference semantics: RegNeRF/internal/utils.py:63-66 (shard), RegNeRF/train.py:246-274
    (pmean of the gradient, then the optimizer)."""
    from consistentnerf_amd import distributed as D, run_nerf as R, run_nerf_view as V
    from consistentnerf_amd.optim import FusedAdam
    _, _, rays = _c2(dev)
    tgt = torch.rand(4096, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(9))

    def build():
        coarse, fine, _ = _c2(dev, 1)
        kw = _kwargs(coarse, fine, 64, 128, 0.0, False, 0.0, False)       # deterministic sampling: shards see the batch's samples
        opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)
        return kw, opt

    def loss_of(kw, r, t):
        out = V.render_rays(r, **kw)
        return R.img2mse(out["rgb_map"], t) + R.img2mse(out["rgb0"], t)
    kw1, opt1 = build()
    opt1.zero_grad()
    l1 = loss_of(kw1, rays, tgt)
    l1.backward()
    g1 = opt1.flat_grad.clone()
    opt1.step()
    kw8, opt8 = build()
    opt8.zero_grad()
    l8 = 0.0
    for r in range(8):
        lo, hi = D.shard_bounds(4096, r, 8)
        assert hi - lo == 512
        l = loss_of(kw8, rays[lo:hi], tgt[lo:hi])
        l.backward()                                   # accumulates into the flat gradient, like the all-reduce's sum
        l8 = l8 + l.detach() / 8
    g8 = opt8.flat_grad.clone() / 8
    opt8.step(grad_scale=1.0 / 8)
    s = float(g1.abs().max())
    d = float((g8 - g1).abs().max())
    print(f"  8 x 512-ray shards vs one 4096-ray step: loss {float(l8):.7f} vs {float(l1):.7f}; max|d grad| {d:.3e} of max {s:.3e}")
    assert abs(float(l8) - float(l1)) <= 1e-6 * abs(float(l1)) + 1e-7
    assert d <= 5e-6 * s
    dp = (opt8.flat_param - opt1.flat_param).abs()
    frac = float((dp > 1e-7).float().mean())
    print(f"  stepped weights: max|d| {float(dp.max()):.3e}, fraction differing by > 1e-7: {frac:.4f}")
    assert float(dp.max()) <= 1e-4 and frac < 0.01


def test_sharded_ranks_consume_the_global_random_streams(dev):
    """render_rays(..., _global_rows=(offset, total)): a rank holding rows [offset, offset + n) of a global batch draws the
    jitter (R:368-382), the resampling u (H:214-229) and the density noise (R:287-288) for the WHOLE batch and takes its rows, so
    with the same generator state on every rank an N-rank step sees exactly the random numbers of the 1-rank step on the whole
    batch — the shard's outputs equal the whole batch's rows bit for bit."""
    from consistentnerf_amd import run_nerf_view as V
    coarse, _ = make_model(4, 128, True, 5, 71, dev)
    fine, _ = make_model(4, 128, True, 5, 72, dev)
    rays = T(I.ray_batch(384, seed=9), dev)
    kw = _kwargs(coarse, fine, 32, 48, 1.0, False, 1.0, False)       # perturb = 1, raw_noise_std = 1: all three streams live
    with torch.no_grad():
        torch.manual_seed(123)
        whole = V.render_rays(rays, _debug=True, **kw)
        for w in (2, 3):
            for r in range(w):
                lo, hi = r * 384 // w, (r + 1) * 384 // w
                torch.manual_seed(123)                              # (every rank holds the same generator state)
                part = V.render_rays(rays[lo:hi], _debug=True, _global_rows=(lo, 384), **kw)
                for k in ("rgb_map", "depth_map", "rgb0", "_z_coarse", "_z_vals", "z_std"):
                    assert torch.equal(part[k], whole[k][lo:hi]), (w, r, k)
        # through render() (one chunk per step, the training case: N_rand <= chunk)
        K = I.intrinsics(100, 100, 138.0)
        kwr = dict({k: v for k, v in kw.items() if k != "lindisp"}, lindisp=False, near=2.0, far=6.0, use_viewdirs=True, ndc=False)
        torch.manual_seed(123)
        ref = V.render(100, 100, K, chunk=32768, rays=torch.stack([rays[:, 0:3], rays[:, 3:6]], 0), **kwr)
        torch.manual_seed(123)
        out1 = V.render(100, 100, K, chunk=32768, rays=torch.stack([rays[128:384, 0:3], rays[128:384, 3:6]], 0), _global_rows=(128, 384), **kwr)
        assert torch.equal(out1[0], ref[0][128:384]) and torch.equal(out1[3], ref[3][128:384])
