"""GPU tests of the round-4 launch-count work: the jitter / resampling streams generated inside the kernels
that consume them (csrc/rng.hpp; oracle/philox.py is the numpy restatement pinned on Random123's known-answer vectors) and the
photometric loss folded into the compositing launches (run_nerf.render_loss = R:764-775 as one call).  Everything here is bit-exact:
the fused forms perform the operations of the separate ones in the same order."""
import numpy as np
import pytest
import torch

import _inputs as I
from oracle import philox as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch.device("cuda:0")


def same(a, b):
    """bit-equal, NaNs (disp of a ray that hits nothing: 0 / 0, R:302) in the same places"""
    return torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0))


def T(a, dev=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    return t.to(dev) if dev is not None else t


def make_model(D, W, vd, och, seed, dev):
    from consistentnerf_amd.run_nerf_helpers import NeRF
    sd = I.nerf_state_dict(D, W, 10, 4, och, vd, seed)
    m = NeRF(D=D, W=W, input_ch=63, output_ch=och, skips=[4], input_ch_views=27 if vd else 0, use_viewdirs=vd)
    m.load_state_dict({k: T(v) for k, v in sd.items()}, strict=True)
    return m.to(dev)


def _kwargs(coarse, fine, Nc, Nf, perturb, white=False):
    from consistentnerf_amd.run_nerf import run_network
    from consistentnerf_amd.run_nerf_helpers import get_embedder
    e, _ = get_embedder(10, 0)
    ed, _ = get_embedder(4, 0)
    q = lambda inputs, viewdirs, fn: run_network(inputs, viewdirs, fn, embed_fn=e, embeddirs_fn=ed)  # noqa: E731
    return dict(network_query_fn=q, perturb=perturb, N_importance=Nf, network_fine=fine, N_samples=Nc, network_fn=coarse,
                white_bkgd=white, raw_noise_std=0.0, lindisp=False)


# ------------------------------------------------------------------------------------------------ streams
@pytest.mark.parametrize("seed,offset,row0,rows,cols", [(0, 0, 0, 7, 5), (1234, 4, 0, 4096, 64), (2 ** 64 - 3, 2 ** 40 + 8, 0, 33, 128),
                                                        (99, 12, 3584, 512, 128), (5, 2 ** 63 + 4, 2 ** 33, 64, 64)])
def test_uniform_streams_equal_the_numpy_philox(dev, seed, offset, row0, rows, cols):
    """The kernel's stream vs oracle/philox.py (pinned on the Random123 vectors): bit for bit, including 64-bit seeds / offsets,
    element indices beyond 2^32, and a shard's rows being the rows of the global stream."""
    from consistentnerf_amd import ops
    got = ops.uniform_rng(ops.RngStream(seed, offset, None, row0), rows, cols, dev).cpu().numpy()
    ref = P.uniform(seed, offset, rows, cols, row0)
    assert got.dtype == ref.dtype == np.float32 and np.array_equal(got, ref)
    assert got.min() >= 0.0 and got.max() < 1.0
    if rows * cols >= 1 << 15:      # first two moments of U[0, 1) at 6 sigma
        n = rows * cols
        assert abs(got.mean() - 0.5) < 6 * np.sqrt(1 / 12 / n) and abs(got.var() - 1 / 12) < 6 * np.sqrt(1 / 180 / n)


def test_coarse_z_and_resample_consume_the_streams(dev):
    """cnerf_coarse_z_rng / cnerf_resample_rng == the tensor-fed entry points on the materialised streams (offset + 0 jitter, offset
    + 1 resampling), and a shard [row0, row0 + n) of a batch reproduces those rows of the unsharded call."""
    from consistentnerf_amd import ops
    B, Nc, Nf = 1000, 64, 128
    rays = T(I.ray_batch(B, seed=11), dev)
    rng = ops.RngStream(77, 40)
    z_a = ops.coarse_z(rays, Nc, None, False, rng=rng)
    z_b = ops.coarse_z(rays, Nc, ops.uniform_rng(rng, B, Nc, dev, 0), False)
    assert torch.equal(z_a, z_b)
    assert not torch.equal(z_a, ops.coarse_z(rays, Nc, None, False))       # (jittered at all)
    w = torch.rand(B, Nc, device=dev, generator=torch.Generator(device=dev).manual_seed(3)) ** 4
    f_a, s_a, smp_a, ind_a = ops.resample(z_a, w, None, want_samples=True, rng=rng, Nf=Nf)
    f_b, s_b, smp_b, ind_b = ops.resample(z_a, w, ops.uniform_rng(rng, B, Nf, dev, 1), want_samples=True)
    assert torch.equal(f_a, f_b) and torch.equal(s_a, s_b) and torch.equal(smp_a, smp_b) and torch.equal(ind_a, ind_b)
    lo, n = 376, 250
    sh = ops.RngStream(77, 40, None, lo)
    assert torch.equal(ops.coarse_z(rays[lo:lo + n].contiguous(), Nc, None, False, rng=sh), z_a[lo:lo + n])
    f_s, s_s = ops.resample(z_a[lo:lo + n].contiguous(), w[lo:lo + n].contiguous(), None, rng=sh, Nf=Nf)
    assert torch.equal(f_s, f_a[lo:lo + n]) and torch.equal(s_s, s_a[lo:lo + n])


def test_render_rays_streams_follow_the_torch_generator(dev):
    """The streams are named by the device generator's (seed, philox offset): torch.manual_seed() reproduces them, every
    render_rays call advances the offset by ops.RNG_STRIDE (host-side integers: no launch), and the depths render_rays reports are the
    ones the numpy Philox predicts from that pair."""
    from consistentnerf_amd import ops, run_nerf as R
    coarse, fine = make_model(4, 64, True, 5, 21, dev), make_model(4, 64, True, 5, 22, dev)
    kw = _kwargs(coarse, fine, 16, 24, 1.0)
    rays = T(I.ray_batch(200, seed=5), dev)
    gen = torch.cuda.default_generators[0]
    torch.manual_seed(31)
    off0 = gen.get_offset()
    with torch.no_grad():
        a = R.render_rays(rays, _debug=True, **kw)
        assert gen.get_offset() == off0 + ops.RNG_STRIDE
        b = R.render_rays(rays, _debug=True, **kw)
        torch.manual_seed(31)
        c = R.render_rays(rays, _debug=True, **kw)
    assert torch.equal(a["_z_vals"], c["_z_vals"]) and torch.equal(a["rgb_map"], c["rgb_map"])
    assert not torch.equal(a["_z_coarse"], b["_z_coarse"])
    t_rand = T(P.uniform(31, off0, 200, 16), dev)
    assert torch.equal(a["_z_coarse"], ops.coarse_z(rays, 16, t_rand, False))
    u = T(P.uniform(31, off0 + 1, 200, 24), dev)
    z_fine, _ = ops.resample(a["_z_coarse"], a["_weights_coarse"] if "_weights_coarse" in a else _coarse_weights(a, rays, kw, dev), u)
    assert torch.equal(a["_z_vals"], z_fine)
    # the tensor-fed round-3 form stays available
    ops.IN_KERNEL_RNG = False
    try:
        torch.manual_seed(31)
        with torch.no_grad():
            d = R.render_rays(rays, _debug=True, **kw)
        assert d["_z_coarse"].shape == a["_z_coarse"].shape and not torch.equal(d["_z_coarse"], a["_z_coarse"])
    finally:
        ops.IN_KERNEL_RNG = True


def _coarse_weights(out, rays, kw, dev):
    from consistentnerf_amd import ops
    return ops.composite_forward(out["_raw_coarse"], out["_z_coarse"], rays, None, False)[3]


# ------------------------------------------------------------------------------------------------ loss folded into compositing
@pytest.mark.parametrize("B", [1, 3, 4, 5, 255, 256, 257, 511, 4096, 10007, 65280])
@pytest.mark.parametrize("S,white", [(64, False), (192, True), (40, False)])
def test_composite_with_the_loss_folded_in(dev, B, S, white):
    """cnerf_composite_fwd_mse / _bwd_mse vs cnerf_composite_fwd + cnerf_mse + `d_x * g` + cnerf_composite_bwd: the maps and d_raw
    bit for bit; the loss to fp64-association round-off (a different, fixed, order of the same fp64 sum) — and identical across
    repeated launches (fixed-order second stage by whichever workgroup finishes last; the ticket counter re-arms itself)."""
    from consistentnerf_amd import ops
    g = torch.Generator(device=dev).manual_seed(B * 7 + S)
    raw = torch.randn(B, S, 4, device=dev, generator=g) * 2
    rays = T(I.ray_batch(B, seed=2), dev)
    z = ops.coarse_z(rays, S, torch.rand(B, S, device=dev, generator=g), False)
    tgt = torch.rand(B, 3, device=dev, generator=g)
    add = torch.rand(1, device=dev, generator=g)
    rgb, disp, acc, wts, depth = ops.composite_forward(raw, z, rays, None, white)
    loss_ref, d_x = ops.mse(rgb, tgt)
    outs = [ops.composite_forward_mse(raw, z, rays, None, white, tgt, loss_add=add) for _ in range(3 if B > 20000 else 25)]
    for o in outs:
        for a, b in zip(o[:5], (rgb, disp, acc, wts, depth)):
            assert same(a, b)
        assert torch.equal(o[5], outs[0][5])
    want = (loss_ref + add[0]).item()
    assert abs(outs[0][5].item() - want) <= 2e-7 * abs(want), (outs[0][5].item(), want)
    assert abs(ops.composite_forward_mse(raw, z, rays, None, white, tgt)[5].item() - loss_ref.item()) <= 2e-7 * loss_ref.item()
    for gl in (None, torch.tensor(0.37, device=dev)):
        seed = d_x if gl is None else d_x * gl
        want_d = ops.composite_backward(raw, z, rays, None, white, seed, None, None, None)
        got_d = ops.composite_backward_mse(raw, z, rays, None, white, rgb, tgt, gl)
        assert torch.equal(got_d, want_d)


def test_ticket_publish_under_concurrent_streams_and_hbm_pressure(dev):
    """VERDICT r04 item 6 / ADVICE r04: the loss form publishes a per-workgroup partial (write-through store, completed by an
    explicit vmcnt(0)) and then takes a two-level ticket; the last workgroup sums the partials.  Stress: two streams launch it 1000
    times each, concurrently, on their own counter blocks and workspaces (per-stream, ops._mse_counter), alternating between inputs
    of different size so that a workspace always holds the PREVIOUS launch's partials of OTHER data (a partial read before it was
    published would be a wrong, not a stale-but-equal, number), while a third stream keeps HBM saturated with 1 GiB copies.  Every
    one of the 2000 losses must equal, bit for bit, the value the same launch gives alone on an idle GPU, which in turn equals the
    two-kernel path (cnerf_composite_fwd + cnerf_mse) to the fp64-association round-off."""
    from consistentnerf_amd import ops
    g = torch.Generator(device=dev).manual_seed(5)
    cases = []
    for B, S in ((4096, 192), (1000, 64), (8200, 64), (513, 192)):
        raw = torch.randn(B, S, 4, device=dev, generator=g) * 2
        rays = T(I.ray_batch(B, seed=B), dev)
        z = ops.coarse_z(rays, S, torch.rand(B, S, device=dev, generator=g), False)
        tgt = torch.rand(B, 3, device=dev, generator=g)
        quiet = ops.composite_forward_mse(raw, z, rays, None, False, tgt)
        two = ops.mse(ops.composite_forward(raw, z, rays, None, False)[0], tgt, want_grad=False)[0]
        assert abs(quiet[5].item() - two.item()) <= 2e-7 * two.item()
        cases.append((raw, z, rays, tgt, quiet[5].clone(), quiet[0].clone()))
    torch.cuda.synchronize()
    N = 1000
    streams = [torch.cuda.Stream(device=dev) for _ in range(3)]
    got = [torch.empty(N, device=dev) for _ in range(2)]
    rgb_bad = [torch.zeros((), device=dev, dtype=torch.int32) for _ in range(2)]
    big = torch.empty(2, 1 << 28, device=dev)            # 2 x 1 GiB
    ctr_ptrs = []
    for rnd in range(N // 50):
        with torch.cuda.stream(streams[2]):
            for _ in range(12):                           # ~2 GiB of traffic per copy: the box's HBM stays busy under the launches
                big[1].copy_(big[0], non_blocking=True)
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                if rnd == 0:
                    ctr_ptrs.append(ops._mse_counter(dev).data_ptr())
                for i in range(rnd * 50, rnd * 50 + 50):
                    raw, z, rays, tgt, _, rgb_q = cases[(i + k) % len(cases)]
                    o = ops.composite_forward_mse(raw, z, rays, None, False, tgt)
                    got[k][i] = o[5][0]
                    if i % 100 == 0:
                        rgb_bad[k] += (o[0] != rgb_q).any().to(torch.int32)
    torch.cuda.synchronize()
    assert ctr_ptrs[0] != ctr_ptrs[1]                     # distinct counter blocks
    for k in range(2):
        want = torch.stack([cases[(i + k) % len(cases)][4][0] for i in range(N)])
        bad = (got[k] != want).nonzero().flatten()
        assert bad.numel() == 0, (k, bad[:10].tolist(), got[k][bad[:10]].tolist(), want[bad[:10]].tolist())
        assert int(rgb_bad[k]) == 0
        with torch.cuda.stream(streams[k]):
            assert int(ops._mse_counter(dev).abs().sum()) == 0     # re-armed
    del big


def test_composite_loss_limits(dev):
    """Beyond cnerf_composite_mse_max_rays() rays per call the entry point refuses (render_loss then takes the reference's lines)."""
    from consistentnerf_amd import ops
    from consistentnerf_amd._lib import CnerfError
    n = ops.composite_mse_max_rays()
    assert n == 130560
    B = n + 4
    raw, z = torch.zeros(B, 8, 4, device=dev), torch.rand(B, 8, device=dev).sort(-1)[0]
    rays = torch.ones(B, 11, device=dev)
    with pytest.raises(CnerfError):
        ops.composite_forward_mse(raw, z, rays, None, False, torch.zeros(B, 3, device=dev))
    # the counters are left zero by every call, also by the refused one
    assert int(ops._mse_counter(dev).abs().sum()) == 0


def _step_pair(dev, Nf, owned):
    """Two identical model pairs + their optimisers (FusedAdam-owned or plain parameters)."""
    from consistentnerf_amd.optim import FusedAdam
    out = []
    for _ in range(2):
        coarse = make_model(4, 128, True, 5 if Nf else 4, 93, dev)
        fine = make_model(4, 128, True, 5, 94, dev) if Nf else None
        params = list(coarse.parameters()) + (list(fine.parameters()) if fine is not None else [])
        opt = FusedAdam(params, lr=5e-4) if owned else None
        out.append((coarse, fine, params, opt, _kwargs(coarse, fine, 32, Nf, 1.0)))
    return out


@pytest.mark.parametrize("Nf", [48, 0])
@pytest.mark.parametrize("owned", [True, False])
def test_render_loss_equals_the_reference_lines(dev, Nf, owned):
    """run_nerf.render_loss + run_nerf.backward vs the training loop's own lines (R:764-775: render, img2mse, img_loss0, `+`,
    loss.backward()) from the same generator state: loss, maps and every parameter gradient bit for bit — on FusedAdam-owned
    parameters (direct-accumulate route, merged coarse + fine backward) and on plain ones (tensor route)."""
    from consistentnerf_amd import run_nerf as R
    (c0, f0, p0, o0, kw0), (c1, f1, p1, o1, kw1) = _step_pair(dev, Nf, owned)
    B = 300
    rays = T(I.ray_batch(B, seed=8), dev)
    rays_od = (rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous())
    tgt = torch.rand(B, 3, device=dev, generator=torch.Generator(device=dev).manual_seed(4))
    K = np.array([[100., 0, 20], [0, 100., 20], [0, 0, 1]])
    common = dict(chunk=4096, rays=rays_od, retraw=True, near=2.0, far=6.0, ndc=False, use_viewdirs=True)
    for rep in range(2):             # twice: the second pass runs on dropped / accumulated gradients
        torch.manual_seed(17 + rep)
        rgb, disp, acc, ex = R.render(40, 40, K, **common, **kw0)
        if o0 is not None:
            o0.zero_grad()
        loss0 = R.img2mse(rgb, tgt)
        if Nf:
            loss0 = loss0 + R.img2mse(ex["rgb0"], tgt)
        loss0.backward()
        torch.manual_seed(17 + rep)
        loss1, rgb1, disp1, acc1, ex1 = R.render_loss(40, 40, K, tgt, **common, **kw1)
        if o1 is not None:
            o1.zero_grad()
        R.backward(loss1)
        assert loss1.shape == loss0.shape == () and abs(loss1.item() - loss0.item()) <= 2e-7 * abs(loss0.item())
        assert same(rgb1, rgb) and same(disp1, disp) and same(acc1, acc) and same(ex1["raw"], ex["raw"])
        assert not rgb1.requires_grad and "loss" not in ex1
        if Nf:
            assert torch.equal(ex1["rgb0"], ex["rgb0"]) and torch.equal(ex1["z_std"], ex["z_std"])
        if owned:
            assert torch.equal(o1.flat_grad, o0.flat_grad) and o0.flat_grad.abs().max() > 0
        else:
            live = 0
            for a, b in zip(p0, p1):
                if a.grad is None:          # (the output_linear a view-dependent network never uses)
                    assert b.grad is None
                    continue
                assert torch.equal(a.grad, b.grad)
                live += int(a.grad.abs().max() > 0)
            assert live >= len(p0) // 2
    # loss.backward() (implicit ones seed) and a scaled loss give the same / the scaled gradient
    torch.manual_seed(5)
    la = R.render_loss(40, 40, K, tgt, **common, **kw1)[0]
    ga = torch.autograd.grad(la * 0.5, p1, allow_unused=True)
    torch.manual_seed(5)
    lb = R.render_loss(40, 40, K, tgt, **common, **kw1)[0]
    gb = torch.autograd.grad(lb, p1, allow_unused=True)
    for a, b in zip(ga, gb):
        if a is not None:
            assert torch.allclose(a, 0.5 * b, rtol=2e-6, atol=0)


def test_render_loss_falls_back_to_the_lines_beyond_one_chunk(dev):
    """Batches larger than `chunk` take R:764-775 literally (render + img2mse): same values as the fused call on one chunk up to the
    association of the loss sum — perturb = 0, so that chunking does not change the streams."""
    from consistentnerf_amd import run_nerf as R
    coarse, fine = make_model(4, 64, True, 5, 21, dev), make_model(4, 64, True, 5, 22, dev)
    kw = _kwargs(coarse, fine, 16, 16, 0.0)
    rays = T(I.ray_batch(100, seed=9), dev)
    od = (rays[:, 0:3].contiguous(), rays[:, 3:6].contiguous())
    tgt = torch.rand(100, 3, device=dev)
    K = np.array([[100., 0, 20], [0, 100., 20], [0, 0, 1]])
    common = dict(rays=od, near=2.0, far=6.0, ndc=False, use_viewdirs=True)
    l_one = R.render_loss(40, 40, K, tgt, chunk=128, **common, **kw)[0]
    l_many = R.render_loss(40, 40, K, tgt, chunk=32, **common, **kw)[0]
    assert abs(l_one.item() - l_many.item()) <= 2e-7 * l_one.item()
    g1 = torch.autograd.grad(l_one, list(fine.parameters()), allow_unused=True)
    g2 = torch.autograd.grad(l_many, list(fine.parameters()), allow_unused=True)
    for a, b in zip(g1, g2):
        if a is not None:
            assert torch.allclose(a, b, rtol=1e-4, atol=1e-7 * float(a.abs().max()) + 1e-12)


def test_graphed_fused_step_sees_the_eager_steps_numbers(dev):
    """GraphedStep over the fused step (render_loss, in-kernel streams with perturb = 1): every replay reads the generator's current
    (seed, offset) from device memory, so the recorded step replayed N times equals N eager steps from the same generator state
    bit for bit — fresh jitter every replay, none of it from a generator launch."""
    from consistentnerf_amd import run_nerf as R
    from consistentnerf_amd.graph import GraphedStep
    from consistentnerf_amd.optim import FusedAdam
    K = np.array([[100., 0, 20], [0, 100., 20], [0, 0, 1]])

    def build():
        coarse, fine = make_model(4, 128, True, 5, 93, dev), make_model(4, 128, True, 5, 94, dev)
        kw = _kwargs(coarse, fine, 16, 16, 1.0)
        opt = FusedAdam(list(coarse.parameters()) + list(fine.parameters()), lr=5e-4)

        def step_fn(ro, rd, tgt):
            loss = R.render_loss(40, 40, K, tgt, chunk=4096, rays=(ro, rd), near=2.0, far=6.0, ndc=False, use_viewdirs=True, **kw)[0]
            opt.zero_grad()
            R.backward(loss)
            opt.step()
            return loss
        return opt, step_fn
    nstep = 6
    gb = torch.Generator(device=dev).manual_seed(1)
    batches = []
    for i in range(nstep):
        r = T(I.ray_batch(96, seed=40 + i), dev)
        batches.append((r[:, 0:3].contiguous(), r[:, 3:6].contiguous(), torch.rand(96, 3, device=dev, generator=gb)))
    opt_e, step_e = build()
    opt_e.make_capturable()
    torch.manual_seed(2024)
    for _ in range(3):
        step_e(*batches[0])
    le = [step_e(*b).item() for b in batches]
    opt_g, step_g = build()
    torch.manual_seed(2024)
    gs = GraphedStep(step_g, opt_g, batches[0], warmup=3)
    lg = [float(gs(*b).detach().clone()) for b in batches]
    assert le == lg, (le, lg)
    assert torch.equal(opt_e.flat_param, opt_g.flat_param)
    # the same batch replayed twice draws fresh jitter
    a = float(gs(*batches[0]).detach().clone())
    b = float(gs(*batches[0]).detach().clone())
    assert a != b
