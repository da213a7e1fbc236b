"""CPU-side tests (no GPU): the C-ABI library loads and exports every symbol include/cnerf.h declares, the
host logic (tensor bookkeeping, sharding) is right, and the product path REFUSES to run without a GPU
(no CPU fallback).  No compute calls are made here."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from consistentnerf_amd import _lib, build
    if not os.path.exists(_lib.LIB_PATH):
        build.build(verbose=False)
    return _lib.load()


def test_header_symbols_all_exported(lib):
    from consistentnerf_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "cnerf.h")).read()
    declared = set(re.findall(r"\b(cnerf_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True).stdout
    exported = set(re.findall(r" T (cnerf_[a-z0-9_]+)", out))
    assert declared <= exported, declared - exported
    assert lib.cnerf_abi_version() == 6
    assert lib.cnerf_strerror(-2).decode().startswith("configuration")


def test_every_entry_point_rejects_null_and_empty_arguments(lib):
    """include/cnerf.h: "<0 for an argument error".  Every int-returning entry point called with null pointers and zero sizes
    returns a CNERF_E_* code before it touches the device (so this runs without a GPU) — no launch, no crash.  One child
    process for all of them: a regression that dereferences a null would end the child, not the test session."""
    code = r"""
import sys, ctypes as C
sys.path.insert(0, %r)
from consistentnerf_amd import _lib as L
lib = C.CDLL(L.LIB_PATH)
for name, (res, args) in L.SIGNATURES.items():
    if res is not C.c_int or not args:
        continue
    fn = getattr(lib, name); fn.restype = res; fn.argtypes = args
    kind = lambda a: getattr(a, "_type_", None)          # one-letter code for the scalar ctypes, else a pointer / char_p
    vals = [0.0 if kind(a) in ("f", "d") else (0 if isinstance(kind(a), str) and kind(a) in "iIlLqQhHbB" else None) for a in args]
    print(name, fn(*vals), flush=True)
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout[-300:], r.stderr[-300:])
    seen = dict(ln.split() for ln in r.stdout.strip().splitlines())
    from consistentnerf_amd import _lib
    import ctypes as C
    want = {n for n, (res, args) in _lib.SIGNATURES.items() if res is C.c_int and args}
    assert set(seen) == want and len(want) >= 40
    bad = {n: c for n, c in seen.items() if int(c) not in (-1, -2, -3)}
    assert not bad, bad


def test_tensor_bookkeeping_matches_reference_state_dict(lib):
    """cnerf_tensor_shape order/shapes == reference state_dict (minus the 3 scalars) for the BASELINE nets."""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import _inputs as I
    from consistentnerf_amd.ops import NetSpec
    from consistentnerf_amd.run_nerf_helpers import NeRF
    for D, W, vd, och in ((8, 256, True, 5), (4, 128, True, 4), (4, 128, False, 5), (8, 128, True, 5)):
        spec = NetSpec(D=D, W=W, use_viewdirs=vd, output_ch=och)
        shapes = spec.tensor_shapes()
        ref = [s for n, s in I.nerf_param_shapes(D, W, 63, 27 if vd else 0, och, vd) if n not in
               ("temp_rgb", "temp_depth", "depth_scale")]
        assert shapes == [tuple(s) for s in ref]
        m = NeRF(D=D, W=W, input_ch=63, output_ch=och, skips=[4], input_ch_views=27 if vd else 0, use_viewdirs=vd)
        assert [tuple(t.shape) for t in m.kernel_tensors()] == shapes
        assert m.spec() == spec if vd else m.spec().D == D
        assert list(m.state_dict().keys()) == [n for n, _ in I.nerf_param_shapes(D, W, 63, 27 if vd else 0, och, vd)]
    import ctypes as C
    assert lib.cnerf_packed_floats(C.byref(NetSpec(D=5).c())) == -1          # D == skip+1: reference mis-shapes
    assert lib.cnerf_packed_floats(C.byref(NetSpec(W=192).c())) == -1        # outside the compiled envelope


def test_no_cpu_fallback():
    from consistentnerf_amd import ops, CnerfError
    from consistentnerf_amd.run_nerf_helpers import sample_pdf
    with pytest.raises(CnerfError):
        ops.embed(torch.zeros(4, 3), 10)
    with pytest.raises(CnerfError):
        sample_pdf(torch.zeros(2, 63), torch.zeros(2, 62), 8, det=True)
    if not torch.cuda.is_available():
        from consistentnerf_amd import run_nerf
        with pytest.raises(CnerfError):
            run_nerf._default_device()
    src = "".join(open(os.path.join(ROOT, "consistentnerf_amd", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "consistentnerf_amd")) if f.endswith(".py"))
    assert "oracle" not in src.replace("no CPU oracle", ""), "product code must never import the oracle"


def test_shard_bounds():
    from consistentnerf_amd.distributed import shard_bounds
    for n in (4096, 4097, 7, 8):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests", "golden"))
import torch.distributed as dist
import _inputs as I
from consistentnerf_amd import distributed as D
from oracle import nerf_oracle as O
rank, world, _ = D.init_from_env("gloo")
torch.manual_seed(0)
# global batch (identical on every rank), masked depth/rgb losses with GLOBAL counts, flat-grad all-reduce
B = 64
sd = O.as_tensors(I.nerf_state_dict(4, 128, 10, 4, 4, True, seed=41), True)
net, cfg = O.NetCfg(4, 128, output_ch=4), O.RenderCfg(16, 0, 0.0)
rays = torch.from_numpy(I.ray_batch(B, seed=3)); rs = np.random.RandomState(1)
target = torch.from_numpy(rs.uniform(size=(B, 3)).astype(np.float32))
prior = torch.from_numpy(rs.uniform(2, 6, size=(B,)).astype(np.float32))
mask = torch.from_numpy((rs.uniform(size=(B,)) < 0.6).astype(np.float32))
def loss_on(sl, counts):
    out = O.render_rays(rays[sl], sd, None, net, cfg)
    m = mask[sl]
    # per-ray weights normalised by GLOBAL counts (what hardmask_losses does with counts=...)
    w = torch.where(m == 1, 1.0 / counts[0], 0.2 / counts[1])
    l = (w[:, None] * (out["rgb_map"] - target[sl]) ** 2).sum() / 3
    l = l + (((out["depth_map"] - prior[sl]) / 6.0) ** 2 * (m == 1)).sum() / counts[0]
    return l
params = [p for k, p in sd.items() if k not in ("temp_rgb", "temp_depth", "depth_scale")]
counts = D.global_mask_counts(D.shard_batch(mask))
assert counts.tolist() == [float((mask == 1).sum()), float((mask == 0).sum())]
lo, hi = D.shard_bounds(B)
g = torch.autograd.grad(loss_on(slice(lo, hi), counts), params, allow_unused=True)
flat = torch.cat([(x if x is not None else torch.zeros_like(p)).reshape(-1) for x, p in zip(g, params)])
D.allreduce_sum_(flat)
gref = torch.autograd.grad(loss_on(slice(0, B), counts), params, allow_unused=True)
fref = torch.cat([(x if x is not None else torch.zeros_like(p)).reshape(-1) for x, p in zip(gref, params)])
err = (flat - fref).abs().max().item() / fref.abs().max().item()
assert err < 1e-5, err
t = torch.ones(3); D.allreduce_mean_(t); assert torch.allclose(t, torch.ones(3))
# GradReducer: the flat gradient goes out in per-network slices as the networks' backward nodes finish
class FakeOpt:                       # the two members of FusedAdam the reducer uses (FusedAdam itself needs a GPU)
    def __init__(self, mods):
        self.params = [p for m in mods for p in m.parameters()]
        self.offs = np.cumsum([0] + [p.numel() for p in self.params]).tolist()
        self.flat_grad = torch.zeros(self.offs[-1])
        self.index = {id(p): i for i, p in enumerate(self.params)}
    def slice_of(self, params):
        idx = sorted(self.index[id(p)] for p in params)
        assert idx == list(range(idx[0], idx[-1] + 1))
        return self.offs[idx[0]], self.offs[idx[-1] + 1]
mods = [torch.nn.Linear(5, 3), torch.nn.Linear(4, 2)]
opt = FakeOpt(mods)
for mean in (True, False):
    red = D.GradReducer(opt, mods, mean=mean)
    assert red.bytes_per_step == 4 * opt.offs[-1] and [red.slices[id(m)] for m in mods] == [(0, 18), (18, 28)]
    mine = torch.arange(28, dtype=torch.float32) * (rank + 1)
    opt.flat_grad.copy_(mine)
    mods[1]._cnerf_pending = 2                     # a network with two backward nodes reports after the second one
    import consistentnerf_amd.run_nerf as RN
    RN._report_ready(mods[1], True); assert not red._done
    RN._report_ready(mods[1], True); assert red._done == {id(mods[1])}     # fine-first order; mods[0] never reports
    red.finish()
    want = torch.arange(28, dtype=torch.float32) * sum(r + 1 for r in range(world)) * ((1.0 / world) if mean else 1.0)
    assert torch.allclose(opt.flat_grad, want), (opt.flat_grad, want)
    assert not red._works and not red._done and mods[1]._cnerf_pending == 0
# the merged coarse+fine backward reports both networks at once: their adjacent slices leave as ONE message
red = D.GradReducer(opt, mods, mean=False)
opt.flat_grad.copy_(torch.arange(28, dtype=torch.float32) * (rank + 1))
mods[0]._cnerf_pending = mods[1]._cnerf_pending = 1
RN._report_ready_pair(mods[1], mods[0])
assert red.messages == 1 and red._done == {id(mods[0]), id(mods[1])}
red.finish()
assert red.messages == 1 and red.steps == 1
assert torch.allclose(opt.flat_grad, torch.arange(28, dtype=torch.float32) * sum(r + 1 for r in range(world)))
# a network evaluated twice (two pending nodes) next to one evaluated once: the pair report only releases what is final
opt.flat_grad.zero_()
mods[0]._cnerf_pending, mods[1]._cnerf_pending = 2, 1
RN._report_ready_pair(mods[1], mods[0])
assert red._done == {id(mods[1])} and red.messages == 2
red.finish()
assert red.messages == 3
# the two exchanges of a sharded `--ss_loss` step (run_nerf_view.ss_global_stats; VT:917-921 threshold rule, VT:941-966 means):
# all-reduce MIN of the ranks' minimum |z - D_ref| (carried as float bits in meta[4]), all-reduce SUM of the three ray counts
from consistentnerf_amd.run_nerf_view import ss_global_stats
amin_r = np.float32(0.75 - 0.25 * rank)
meta = torch.tensor([100 + rank, 2, 0, 0, int(np.float32(amin_r).view(np.int32)), 40 + rank, 0, 0], dtype=torch.int32)
amin_g, counts_fn = ss_global_stats(meta, 256 + 8 * rank, None)
assert float(amin_g) == 0.75 - 0.25 * (world - 1) and meta[4] == int(np.float32(amin_r).view(np.int32))      # (input untouched)
c3 = counts_fn(meta)
assert c3.tolist() == [float(sum(40 + r for r in range(world))), float(sum(256 + 8 * r for r in range(world))),
                       float(sum(100 + r for r in range(world)))], c3
D.barrier()
if rank == 0: print("DIST_OK", world, err)
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharded_step_equals_single(tmp_path):
    """N>1 path on CPU (gloo, world_size 2): sharded masked-loss gradients + one flat all-reduce == 1-rank."""
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", str(script), ROOT],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "DIST_OK 2" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_row_block_padding():
    """Inference sharding (§8e): blocks tile [0,H), every rank renders ceil(H/w) rows, padding repeats the last own row."""
    from consistentnerf_amd.distributed import row_block
    for H in (1, 7, 8, 756):
        for w in (1, 2, 3, 8):
            rows = -(-H // w)
            nxt = 0
            for r in range(w):
                lo, hi, idx = row_block(H, r, w)
                assert lo == nxt and len(idx) == rows
                nxt = hi
                own = hi - lo
                assert idx[:own].tolist() == list(range(lo, hi))
                assert all(int(i) == max(hi - 1, min(lo, H - 1)) for i in idx[own:])
                assert int(idx.max()) <= H - 1
            assert nxt == H


RENDER_WORKER = r'''
import os, sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests", "golden"))
import torch.distributed as dist
import _inputs as I
from consistentnerf_amd import distributed as D
from oracle import nerf_oracle as O
rank, world, _ = D.init_from_env("gloo")
H, W = 7, 5                                   # 7 rows over 2 ranks: 4 + 3, rank 1 pads one row
K = I.intrinsics(H, W, 9.0)
poses = [torch.from_numpy(I.camera_pose(th, -20.0, 3.0)) for th in (0.0, 40.0)]
calls = []
def render_fn(H_, W_, K_, chunk, rays, **kw):   # any per-ray function stands in for the renderer on CPU
    o, d = rays[0], rays[1]
    calls.append(tuple(o.shape))
    return [torch.sin(o * 1.7 + d * 3.1), (d * o).sum(-1)]
def get_rays_fn(H_, W_, K_, c2w):
    return O.get_rays(H_, W_, K_, c2w)
ft = []
res = D.render_path_sharded(poses, (H, W, 9.0), K, 64, {}, render_fn=render_fn, get_rays_fn=get_rays_fn, frame_times=ft)
lo, hi = D.shard_bounds(H)
assert calls == [(hi - lo, W, 3)] * 2, calls      # every rank renders ITS rows only (4 and 3 of 7); padding is output-side
assert len(ft) == 2
if rank == 0:
    rgbs, disps = res
    for i, c2w in enumerate(poses):
        o, d = O.get_rays(H, W, K, c2w[:3, :4])
        ref = render_fn(H, W, K, 64, torch.stack([o, d], 0))
        assert np.array_equal(rgbs[i], ref[0].numpy()) and np.array_equal(disps[i], ref[1].numpy())
    assert rgbs.shape == (2, H, W, 3) and disps.shape == (2, H, W)
else:
    assert res is None                             # render_path needs the frames on ONE host (R:157-159)
# a frame with fewer rows than ranks: the rank without rows contributes an all-padding block
res1 = D.render_path_sharded(poses[:1], (1, W, 9.0), I.intrinsics(1, W, 9.0), 64, {}, render_fn=render_fn, get_rays_fn=get_rays_fn)
if rank == 0:
    o, d = O.get_rays(1, W, I.intrinsics(1, W, 9.0), poses[0][:3, :4])
    assert np.array_equal(res1[0][0], render_fn(1, W, None, 64, torch.stack([o, d], 0))[0].numpy())
D.barrier()
if rank == 0: print("RENDER_OK", world)
dist.destroy_process_group()
'''


def test_two_rank_gloo_sharded_render_path(tmp_path):
    """render_path rows sharded over 2 ranks (gloo): each rank renders its own rows, ONE gather of the packed (edge-padded)
    blocks to rank 0 per frame, frame == unsharded."""
    script = tmp_path / "render_worker.py"
    script.write_text(RENDER_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script), ROOT],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "RENDER_OK 2" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_bench_roofline_constants_match_the_architecture():
    """bench.py's FLOP-per-ray-sample constants recomputed from the D=8/W=256/viewdirs layer list, and the
    committed PMC summaries it reads for `roofline.traffic` parse to plausible byte counts."""
    sys.path.insert(0, ROOT)
    import bench
    W, Wh, xe, de = 256, 128, 63, 27
    layers = [(W, xe)] + [(W, W)] * 4 + [(W, W + xe)] + [(W, W)] * 2 + [(Wh, W + de), (W, W), (1, W), (3, Wh)]
    fwd = sum(o * i for o, i in layers)
    assert fwd == bench.MAC_FWD == bench.MAC_WGRAD == 593408
    # dgrad skips inputs that need no gradient: gamma(x) into layer 0 and into the skip layer, gamma(d) into the view layer
    assert fwd - (W * xe + W * xe + Wh * de) == bench.MAC_DGRAD == 557696
    assert 2 * (bench.MAC_FWD + bench.MAC_DGRAD + bench.MAC_WGRAD) == 3489024
    t = bench.pmc_traffic("mlp_wgrad", 1048576)         # the merged coarse+fine launch bench.py's roofline names
    assert t is not None and 21e9 < t < 27e9            # 21.6 GB algorithmic reads + 0.3 GB of partials, gamma(x) / h7 read twice
    assert bench.pmc_traffic("mlp_fwd_train", 786432) > 8e9   # the 8.2 GB stash
    assert bench.pmc_traffic("nonexistent", 1) is None


def test_header_is_plain_c_and_the_c_example_links(tmp_path):
    """include/cnerf.h compiles as C99 with gcc (no C++, no torch types in any signature) and the plain-C user of the
    ABI (tests/c_abi/render_smoke.c; the GPU suite runs it) links against libcnerf_hip.so."""
    from consistentnerf_amd import _lib
    libdir = os.path.dirname(_lib.LIB_PATH)
    src = os.path.join(ROOT, "tests", "c_abi", "render_smoke.c")
    cc = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
          "-I" + os.path.join(ROOT, "include"), src, "-o", str(tmp_path / "render_smoke"), "-L/opt/rocm/lib", "-lamdhip64",
          "-L" + libdir, "-l:" + os.path.basename(_lib.LIB_PATH), "-Wl,-rpath,/opt/rocm/lib", "-Wl,-rpath," + libdir]
    r = subprocess.run(cc, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    hdr_only = tmp_path / "h.c"
    hdr_only.write_text('#include "cnerf.h"\nint main(void) { return sizeof(cnerf_render_cfg) == 24 ? 0 : 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"),
                        str(hdr_only), "-o", str(tmp_path / "h")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    assert subprocess.run([str(tmp_path / "h")]).returncode == 0


def test_no_inline_asm_valu_next_to_mfma(tmp_path):
    """The hazard recognizer cannot see a VALU instruction inside an asm statement, so it does not insert the wait states
    gfx950 needs between such a write and an MFMA reading it (nor between an MFMA and such a read).  scripts/isa_hazards.py
    scans the device ISA of every MFMA kernel for both patterns; a synthetic positive control proves it can find them."""
    root = os.path.join(os.path.dirname(__file__), "..")
    sys.path.insert(0, os.path.join(root, "scripts"))
    import isa_hazards
    ctl = tmp_path / "ctl.s"
    ctl.write_text("probe_k:\n"
                   "\t;;#ASMSTART\n\tv_cvt_pk_bf16_f32 v5, v12, v13\n\t;;#ASMEND\n"
                   "\tds_read_b128 v[10:13], v133 offset:5120\n"
                   "\tv_mfma_f32_32x32x16_bf16 a[0:15], v[6:9], v[2:5], a[0:15]\n"
                   "\t;;#ASMSTART\n\tv_pk_mul_f32 v[20:21], a[0:1], v[30:31]\n\t;;#ASMEND\n"
                   ".Lfunc_end0:\n")
    found = isa_hazards.check(str(ctl))
    assert len(found) == 2 and "2 required" in found[0] and "11 required" in found[1], found
    ok = tmp_path / "ok.s"
    ok.write_text("probe_k:\n"
                  "\t;;#ASMSTART\n\tv_cvt_pk_bf16_f32 v5, v12, v13\n\t;;#ASMEND\n"
                  "\ts_nop 1\n"
                  "\tv_mfma_f32_32x32x16_bf16 a[0:15], v[6:9], v[2:5], a[0:15]\n"
                  ".Lfunc_end0:\n")
    assert isa_hazards.check(str(ok)) == []
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    files = "mlp_fwd mlp_bwd wgrad mlp_fwd_bf"
    r = subprocess.run(["bash", os.path.join(root, "scripts", "isa_stats.sh"), str(tmp_path)],
                       env=dict(os.environ, FILES=files), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    found = []
    for f in files.split():
        found += isa_hazards.check(str(tmp_path / f"{f}.s"))
    assert not found, "\n".join(found[:10])
    # register-file facts DESIGN.md quotes (same toolchain as the build): the D=8/W=256 dgrad and inference forward keep
    # everything in registers, wgrad has no scratch, nothing exceeds the 512-register budget of one wave per SIMD
    meta = {}
    for f in ("mlp_fwd", "mlp_bwd", "wgrad"):
        txt = (tmp_path / f"{f}.s").read_text()
        for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size", txt, re.S):
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            meta[name] = {k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
                          for k in ("vgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size")}
    pick = lambda frag: next(v for k, v in meta.items() if frag in k)
    assert pick("mlp_dgrad_kILi8ELb1E")["vgpr_spill_count"] == 0 and pick("mlp_dgrad_kILi8ELb1E")["private_segment_fixed_size"] == 0
    assert pick("mlp_fwd_kILi8ELb1ELb0E")["vgpr_spill_count"] == 0
    assert pick("wgrad_kENS")["private_segment_fixed_size"] == 0 and pick("wgrad_kENS")["vgpr_spill_count"] == 0
    assert all(v["vgpr_count"] <= 512 for v in meta.values())
    assert pick("mlp_fwd_kILi8ELb1ELb1E")["sgpr_spill_count"] == 0 and pick("mlp_fwd_kILi8ELb1ELb1E")["vgpr_spill_count"] <= 8


def test_ticket_is_taken_after_the_partial_store_completed(tmp_path):
    """composite.hip's loss form: the per-workgroup partial (write-through store) must be complete before the workgroup's ticket
    atomic (a workgroup-scope release fence compiles to nothing on gfx950).  scripts/isa_ticket_check.py reads it off
    the device ISA; a synthetic positive control proves it can find the bad order."""
    root = os.path.join(os.path.dirname(__file__), "..")
    sys.path.insert(0, os.path.join(root, "scripts"))
    import isa_ticket_check
    bad = tmp_path / "bad.s"
    bad.write_text("k:\n\tglobal_store_dwordx2 v0, v[2:3], s[6:7] sc1\n\ts_lshr_b32 s3, s2, 6\n"
                   "\tglobal_atomic_add v2, v2, v3, s[62:63] sc0\n\ts_waitcnt vmcnt(0)\n.Lfunc_end0:\n")
    found = isa_ticket_check.check(str(bad), need_sites=0)
    assert len(found) == 1 and "possibly in flight" in found[0], found
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    r = subprocess.run(["bash", os.path.join(root, "scripts", "isa_stats.sh"), str(tmp_path)],
                       env=dict(os.environ, FILES="composite"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    # one publish site per compiled sample-count variant of the loss form (C = 1, 2, 3, 4, 8, 16)
    assert isa_ticket_check.check(str(tmp_path / "composite.s"), need_sites=6) == []


def test_pixel_permutation_oracle_is_a_permutation():
    """oracle/philox.py::permutation (the numpy restatement of csrc/sampler.hip's pixel draw; the GPU suite compares the kernel
    with it bit for bit): a bijection of [0, n) for every n incl. non-powers of two and the degenerate sizes, a function of
    (seed, offset) only, prefixes of one permutation, and uniform to a chi-square test."""
    from oracle import philox as P
    for n in (1, 2, 3, 4, 5, 16, 17, 255, 1000, 4097):
        p = P.permutation(11, 8, n, n)
        assert p.dtype == np.int64 and sorted(p.tolist()) == list(range(n)), n
    a, b = P.permutation(5, 12, 190512, 4096), P.permutation(5, 12, 190512, 512)
    assert np.array_equal(a[:512], b) and len(set(a.tolist())) == 4096 and a.max() < 190512
    assert not np.array_equal(a, P.permutation(5, 16, 190512, 4096)) and not np.array_equal(a, P.permutation(6, 12, 190512, 4096))
    cnt = np.zeros(32)
    for off in range(0, 160, 4):
        cnt += np.bincount(P.permutation(3, off, 190512, 4096) * 32 // 190512, minlength=32)
    chi2 = float(((cnt - cnt.mean()) ** 2 / cnt.mean()).sum() / 31)
    assert 0.3 < chi2 < 2.2, chi2


def test_engine_query_probe_selects_the_plain_route_without_the_private_symbol(monkeypatch):
    """run_nerf's direct-accumulate route and merged coarse+fine backward depend on torch._C._will_engine_execute_node (private).
    The import-time probe must accept this torch's symbol, reject a missing one and one that answers differently, and with the
    query unusable the routing helpers must select the plain autograd route (no direct accumulation, no level pairing)."""
    import torch
    import consistentnerf_amd.run_nerf as R
    fn, why = R._probe_engine_query()
    assert fn is not None and why is None and R._ENGINE_QUERY is not None
    real = torch._C._will_engine_execute_node
    monkeypatch.delattr(torch._C, "_will_engine_execute_node")
    fn, why = R._probe_engine_query()
    assert fn is None and "does not exist" in why
    monkeypatch.setattr(torch._C, "_will_engine_execute_node", lambda node: True, raising=False)   # never refuses
    fn, why = R._probe_engine_query()
    assert fn is None and "answered" in why
    monkeypatch.setattr(torch._C, "_will_engine_execute_node", lambda node: 1 / 0, raising=False)    # raises something else
    fn, why = R._probe_engine_query()
    assert fn is None and "ZeroDivisionError" in why
    monkeypatch.setattr(torch._C, "_will_engine_execute_node", real, raising=False)
    assert R._probe_engine_query()[0] is not None
    # routing with the query switched off
    monkeypatch.setattr(R, "_ENGINE_QUERY", None)
    p = torch.nn.Parameter(torch.ones(3))
    assert R._engine_accumulates(p) is False

    class N:     # stand-ins for two _MlpFn nodes with distinct models
        def __init__(self):
            self.stash, self.model = object(), object()

    class Raw:
        def __init__(self):
            self.grad_fn = N()
    a, b = Raw(), Raw()
    R._link_levels(a, b)
    assert not hasattr(a.grad_fn, "pair") and not hasattr(b.grad_fn, "pair")
    monkeypatch.setattr(R, "_ENGINE_QUERY", real)
    R._link_levels(a, b)
    assert a.grad_fn.pair is b.grad_fn.pair


def test_wgrad_range_plan_invariants():
    """csrc/wgrad.hip::plan_ranges through its host-only debug view (no GPU): for the merged coarse+fine launch at the C4 shard
    (512 rays), the C2 batch (4096 rays) and ragged / tiny sizes — every range is a whole number of 32-point slabs and the
    ranges of a GEMM cover its network's points; GEMMs writing one parameter tensor share their count; 64 ranges from 98 304
    points up, 32 below (the measured optimum: (64, 32) at the C4 shard, (64, 64) at C2); the count of a network does not
    depend on its partner in a merged launch (merged == separate, bit for bit); workgroups are launched longest first;
    counts stay within the partial buffer's capacity."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import wgrad_plan
    for B in list(range(1, 130)) + [512, 513, 1000, 4096, 32768]:
        jobs = wgrad_plan.plan(B)
        assert len(jobs) == 28
        t_prev = None
        by_tensor = {}
        for net, N, K, tiles, ns, ch, tensor in jobs:
            Mp = (B * (192 if net == 0 else 64) + 31) // 32 * 32
            assert ch % 32 == 0 and ns >= 1 and ns * ch >= Mp and (ns - 1) * ch < Mp, (B, net, ns, ch, Mp)
            assert ns <= max(1, min(128, Mp // 512)) and ch * 2600 * 4 < 2 ** 31
            by_tensor.setdefault((net, tensor), set()).add(ns)
            t = ch // 32 * (1024 * tiles + 560)
            assert t_prev is None or t <= t_prev, "grid order must be longest workgroup first"
            t_prev = t
        assert all(len(v) == 1 for v in by_tensor.values())
        n_f = {ns for net, *_, ns, ch, t in jobs if net == 0}
        n_c = {ns for net, *_, ns, ch, t in jobs if net == 1}
        assert len(n_f) == 1 and len(n_c) == 1
        n_f, n_c = n_f.pop(), n_c.pop()
        if B >= 512:
            # (ragged sizes drop an empty trailing range: 63 instead of 64 at 513 rays)
            assert 60 <= n_f <= 64 and (60 <= n_c <= 64 if B >= 1536 else 30 <= n_c <= 32), (B, n_f, n_c)
        # one network alone gets the same count as in the merged launch
        assert {j[4] for j in wgrad_plan.plan(B, S=(192,))} == {n_f} and {j[4] for j in wgrad_plan.plan(B, S=(64,))} == {n_c}
    assert len(wgrad_plan.plan(4096, S=(192,))) == 14


def test_rows_of_global_draws_the_whole_batch_and_slices():
    """run_nerf._rows_of_global (strong sharding): with `global_rows = (offset, total)` the random block is drawn for
    the WHOLE batch and sliced, so ranks holding the same generator state reproduce the single-rank stream row for row; without
    it the draw is just the shard's own."""
    import torch
    from consistentnerf_amd.run_nerf import _rows_of_global
    draw = lambda n, c: torch.rand(n, c)  # noqa: E731
    torch.manual_seed(5)
    whole = draw(12, 3)
    parts = []
    for lo, hi in ((0, 5), (5, 9), (9, 12)):
        torch.manual_seed(5)
        parts.append(_rows_of_global(draw, hi - lo, 3, (lo, 12)))
    assert torch.equal(torch.cat(parts), whole) and all(p.is_contiguous() for p in parts)
    torch.manual_seed(5)
    assert torch.equal(_rows_of_global(draw, 12, 3, None), whole)


def _run_bench(args, env_extra=None, timeout=600):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True,
                       timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return r.returncode, lines, [json.loads(ln) for ln in lines if ln.startswith("{")], r.stderr


def test_bench_gpus_n_launched_plainly_reports_missing_gpus_as_one_json_line():
    """`python bench.py --gpus N` with no torchrun around it must not die on an assertion.  Here (no GPU, or
    fewer than N): exactly ONE line on stdout, JSON, with an `error`, and a non-zero exit status — for RCCL and for gloo."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("host has the GPUs: the launch itself is covered by the gpu-marked test")
    for extra in ({}, {"CNERF_DIST_BACKEND": "gloo"} if torch.cuda.device_count() == 0 else {}):
        rc, lines, objs, err = _run_bench(["--gpus", "2", "--steps", "2", "--warmup", "1"], extra, timeout=120)
        assert rc != 0
        assert len(lines) == 1 and len(objs) == 1, (lines, err[-500:])
        assert objs[0]["n_gpus"] == 2 and objs[0]["value"] is None and "GPU" in objs[0]["error"]
        assert "Traceback" not in err


def test_psnr_twins_permutation_statistics_have_size_and_power():
    """scripts/psnr_parity.py::twins_statistics (round-4 criterion D): on synthetic PSNR tables whose HIP draws come from the
    oracle's own distribution the blocked permutation test does not reject; with a 2 dB systematic bias it does, at every
    milestone; the literal 4-seed paired form can never go below 1/16."""
    import importlib
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "scripts"))
    P = importlib.import_module("psnr_parity")
    ms = [25, 50, 100, 150]

    def table(bias, seed):
        rs = np.random.RandomState(seed)
        rows = []
        for s_ in range(4):
            base = rs.normal(25, 1, 4)
            rows.append({"seed": s_, "psnr_oracle": (base + rs.normal(0, 1, 4)).tolist(),
                         "psnr_oracle_1ulp": (base + rs.normal(0, 1, 4)).tolist(),
                         "psnr_hip_same_init": (base + bias).tolist(),
                         "psnr_hip_draws": [(base + rs.normal(0, 1, 4) + bias).tolist() for _ in range(6)],
                         "max_rel_loss_diff_up_to_hip_same_init": [1e-5, 1e-2, 1.0, 1.0]})
            rows[-1]["psnr_hip_same_init"][0] = rows[-1]["psnr_oracle"][0] + 1e-3 + bias
        return rows
    null = [P.twins_statistics(table(0.0, k), ms, n_perm=3000) for k in range(6)]
    assert sum(o["criterion_D_pass"] for o in null) >= 5                      # size: at most one false alarm in six
    alt = P.twins_statistics(table(2.0, 0), ms, n_perm=3000)
    assert not alt["criterion_D_pass"] and max(alt["T_bias"]["p_two_sided"]) < 0.0125
    assert min(alt["T_pair"]["p_one_sided"]) >= 1.0 / 16 and alt["T_pair"]["smallest_attainable_p"] == 1.0 / 16


def test_training_precision_selection(monkeypatch):
    """run_nerf.training_precision: "fp32" unless opted in; an explicit per-model setting is obeyed as is (an uncovered architecture
    then fails loudly in the launch); the process-wide default (CNERF_TRAIN_PRECISION) only applies to architectures the bf16x3
    kernels cover and leaves every other network on the exact-fp32 path."""
    from consistentnerf_amd import run_nerf as R
    from consistentnerf_amd.run_nerf_helpers import NeRF
    big = NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=27, use_viewdirs=True)
    novd = NeRF(D=8, W=256, input_ch=63, output_ch=5, skips=[4], input_ch_views=0, use_viewdirs=False)
    small = NeRF(D=2, W=64, input_ch=63, output_ch=4, skips=[4], input_ch_views=27, use_viewdirs=True)
    monkeypatch.setattr(R, "DEFAULT_TRAINING_PRECISION", "fp32")
    assert [R.training_precision(m) for m in (big, novd, small)] == ["fp32"] * 3
    monkeypatch.setattr(R, "DEFAULT_TRAINING_PRECISION", "bf16x3")
    assert [R.training_precision(m) for m in (big, novd, small)] == ["bf16x3", "fp32", "fp32"]
    novd.training_precision = "bf16x3"
    assert R.training_precision(novd) == "bf16x3"          # explicit: obeyed (the launch reports the unsupported architecture)
    big.training_precision = "fp32"
    assert R.training_precision(big) == "fp32"
    big.training_precision = "fp8"
    with pytest.raises(ValueError):
        R.training_precision(big)


def test_bench_launches_per_step_reads_the_last_full_step():
    """bench.launches_per_step: one training step = the dispatches between two consecutive `adam_k` of the ordered dispatch list;
    own kernels (anonymous namespace of libcnerf_hip.so) and ATen glue are told apart; fewer than two `adam_k`: nothing to report."""
    import bench
    own = lambda n: f"void (anonymous namespace)::{n}((anonymous namespace)::Args)"   # noqa: E731
    aten = "void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float>, std::array<char*, 1ul> >(int, at::native::FillFunctor<float>, std::array<char*, 1ul>)"
    seq = [own("pack_k"), own("adam_k"),                                                     # tail of step 0
           own("pack_rays_k"), aten, own("mlp_fwd_k<8, true, true>"), own("wgrad_k"), own("adam_k"),     # step 1
           own("pack_rays_k"), own("mlp_fwd_k<8, true, true>"), own("mlp_fwd_k<8, true, true>"), aten, aten, own("wgrad_k"),
           own("wgrad_reduce_k"), own("adam_k"),                                             # step 2 (the one reported)
           own("pack_rays_k")]                                                               # head of an unfinished step
    d = {10 * i + 3: n for i, n in enumerate(seq)}        # dispatch ids: ordered, not dense
    r = bench.launches_per_step(d)
    assert r["total"] == 8 and r["own"] == 6 and r["aten_and_runtime"] == 2
    assert r["kernels"]["mlp_fwd_k<8, true, true>"] == 2 and r["kernels"]["aten:FillFunctor"] == 2 and r["kernels"]["adam_k"] == 1
    assert bench.launches_per_step({1: own("adam_k"), 2: own("pack_k")}) is None


def test_philox_known_answers():
    """oracle/philox.py (the numpy restatement of csrc/rng.hpp's generator) against the known-answer vectors Random123 publishes
    for philox4x32-10 (kat_vectors: counter, key -> output) — the pin of the stream oracle the GPU tests compare the kernels with."""
    from oracle import philox as P
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0), (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, out in kat:
        assert tuple(int(x) for x in P.philox4x32_10(ctr, key)) == out
    u = P.uniform(7, 12, 512, 64, row0=100)
    assert u.dtype == np.float32 and u.min() >= 0.0 and u.max() < 1.0 and abs(u.mean() - 0.5) < 0.01
    # a shard's rows are the rows of the global stream; distinct offsets / seeds are distinct streams
    assert np.array_equal(P.uniform(7, 12, 612, 64)[100:], u)
    assert not np.array_equal(P.uniform(7, 13, 512, 64, row0=100), u) and not np.array_equal(P.uniform(8, 12, 512, 64, row0=100), u)


def test_philox_streams_are_uniform_and_uncorrelated():
    """Distribution of the in-kernel streams, tested on the numpy restatement (the kernels equal it bit for bit, tests/test_gpu_fused_step.py):
    chi-square uniformity on 256 bins, lag correlations along a row / down a column / across the two streams of one render_rays
    call (offset + 0 jitter, offset + 1 resampling) / across consecutive calls (offset + 4) — all at 5 sigma of their nulls."""
    from oracle import philox as P
    rows, cols = 4096, 64
    a = P.uniform(1234, 8, rows, cols).astype(np.float64)
    n = a.size
    hist = np.bincount((a.reshape(-1) * 256).astype(np.int64), minlength=256)
    chi2 = float(((hist - n / 256) ** 2 / (n / 256)).sum())
    assert abs(chi2 - 255) < 5 * np.sqrt(2 * 255), chi2
    c = a - 0.5

    def corr(x, y):
        return float((x * y).mean() / (1 / 12))
    b = P.uniform(1234, 9, rows, cols).astype(np.float64) - 0.5
    nxt = P.uniform(1234, 12, rows, cols).astype(np.float64) - 0.5
    other_seed = P.uniform(1235, 8, rows, cols).astype(np.float64) - 0.5
    for name, r, m in (("along a row", corr(c[:, :-1], c[:, 1:]), rows * (cols - 1)), ("down a column", corr(c[:-1], c[1:]), (rows - 1) * cols),
                       ("jitter vs resampling stream", corr(c, b), n), ("consecutive calls", corr(c, nxt), n),
                       ("adjacent seeds", corr(c, other_seed), n)):
        assert abs(r) < 5 / np.sqrt(m), (name, r)
    # 24-bit grid, like ATen's CPU torch.rand for float32
    assert np.all(a * 2 ** 24 == np.round(a * 2 ** 24))


def test_bench_line_stays_under_the_limit_in_the_worst_case():
    """the driver could not parse round 5's 20-27 KB stdout line.  bench.compact_line() is what goes to stdout
    now; whatever the legs put into the full result object (per-kernel tables, launch lists, prose) the line stays under
    bench.LINE_LIMIT bytes, keeps every contract key, `roofline` / `cpu_baseline` with their numbers, the flat leg_* scalars, and
    names the side file.  Worst case built here: every object the bench can attach, inflated well beyond anything it produced."""
    import json
    import bench
    prose = "x" * 4000
    rows = [{"kernel": f"mlp_kernel_with_a_long_template_name<{i}, true, true>", "points": 1048576, "launches": 200, "avg_ms": 8.87931234,
             "tflops": 140.151234, "frac": 0.89101234, "share_of_step": 0.33851234, "basis": prose} for i in range(12)]
    legs = {k: {"ms_per_step": 33.2, "roofline": {"frac": 0.87, "kernels": rows}, "launches_per_step": {"total": 15, "kernels": {r["kernel"]: 1 for r in rows}},
                "what": prose} for k in ("c4_shard", "c2_bf16x3", "c5", "c3", "c3_ss")}
    legs["hbm_kernels"] = [{"kernel": f"k{i}", "basis": prose, "gbps": 1234.5} for i in range(40)]
    out = {"metric": "train_ray_samples_per_sec", "value": 39972806.64650532, "unit": "ray-samples/s", "n_gpus": 8, "steps": 200, "warmup": 20,
           "ms_per_step": 26.232233560003806, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic", "hip_graph": False, "route": {"backward": "merged", "note": prose, "wgrad_launches_per_step": 1.0},
           "config": dict({"workload": "DTU scan8 3-view " + prose, "loss_entry": prose, "random_streams": prose, "rays_per_gpu": 4096,
                           "global_batch": 32768, "ray_samples_per_ray": 256, "device": "gfx950:sramecc+:xnack-", "cus": 256,
                           "parallelism": "ray-shard dp8, RCCL all-reduce of the flat fp32 grad", "final_loss": 0.164232},
                          **{f"leg_scalar_{i}": 1.23456789 * i for i in range(24)}),
           "roofline": {"bound": "mfma", "kernel": "mlp_wgrad (M=1048576 points)", "achieved": 140.15, "peak": 157.3, "unit": "TFLOP/s",
                        "frac": 0.891, "traffic": 24318790220, "traffic_source": prose, "avg_launch_ms": 8.8793, "whole_step_frac": 0.8866,
                        "kernels": rows, "kernels_measured": prose,
                        "pmc": {"traffic": 24318790220, "FETCH_SIZE_bytes_x2": 24010364646, "WRITE_SIZE_bytes": 308425574, "source": prose,
                                "launches": {"kernels": {r["kernel"]: 3 for r in rows}}}},
           "cpu_baseline": {"value": 81139.5, "unit": "ray-samples/s", "cores": 32, "kind": "port", "physical_cores": 128,
                            "value_physical_cores": 19479.5, "sample": prose, "inference_sample": prose, "sample_physical_cores": prose,
                            "single_thread_value": 29988.7, "inference_value": 203271.0},
           "dist": {"rccl_ranks": 8, "ranks": 8, "backend": "nccl", "messages_per_step": 1.0, "slice_bytes": [4766776] * 64,
                    "bytes_per_step": 4766776, "one_over_world": prose, "allreduce_exposed_ms": 0.1234, "measured": prose},
           "extra": legs}
    assert len(json.dumps(out)) > 200000
    line = bench.compact_line(out, "gpurun_out/bench_detail_8gpus.json")
    blob = json.dumps(line)
    assert len(blob) <= bench.LINE_LIMIT < 6144, len(blob)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "dist", "detail"):
        assert k in line, k
    assert "extra" not in line
    assert line["value"] == out["value"] and line["ms_per_step"] == out["ms_per_step"]
    rf = line["roofline"]
    assert rf["bound"] == "mfma" and rf["frac"] == 0.891 and rf["traffic"] == 24318790220 and rf["peak"] == 157.3 and rf["achieved"] == 140.15
    assert rf["whole_step_frac"] == 0.8866 and 1 <= len(rf["kernels"]) <= 4 and "basis" not in rf["kernels"][0]
    cb = line["cpu_baseline"]
    assert cb["value"] == 81139.5 and cb["cores"] == 32 and cb["kind"] == "port" and 0 < len(cb["sample"]) <= 200
    assert line["config"]["workload"].startswith("DTU scan8 3-view") and line["config"]["rays_per_gpu"] == 4096
    assert all(f"leg_scalar_{i}" in line["config"] for i in range(24))
    assert line["dist"]["ranks"] == 8 and len(line["dist"]["slice_bytes"]) <= 8
    # a realistic object passes through the first (least lossy) form: strings intact up to 200 characters, 4 kernel rows, route kept
    small = dict(out, config={"workload": "DTU scan8 3-view (synthetic)", "rays_per_gpu": 4096}, route={"backward": "merged"})
    small["roofline"] = dict(out["roofline"], traffic_source="rocprofv3 --pmc passes of this command", kernels_measured="timed region")
    small["cpu_baseline"] = dict(out["cpu_baseline"], sample="5 warm steps of 1024 rays", inference_sample="fwd", sample_physical_cores="3 steps")
    small["dist"] = dict(out["dist"], one_over_world="folded into adam_k", measured="HIP events")
    l2 = bench.compact_line(small, None)
    assert len(l2["roofline"]["kernels"]) == 4 and l2["route"] == {"backward": "merged"} and "detail" not in l2
    assert len(json.dumps(l2)) <= bench.LINE_LIMIT
    # and a degenerate object (an error line) still comes out
    assert bench.compact_line({"metric": "m", "value": None, "error": "boom"})["error"] == "boom"
