"""Lane-accurate numpy model of the MFMA data layouts the HIP kernels rely on (CPU only).

`mfma_32x32x2` below is the documented gfx950 semantics of v_mfma_f32_32x32x2_f32
(A: lane l holds A[i=l&31][k=l>>5]; B: lane l holds B[k=l>>5][j=l&31]; C/D: lane l, reg r holds
D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]).  The tests re-state, per lane, the address arithmetic of
csrc/geom.hip (panels, bias group), csrc/mlp_common.hpp (gemm_lds over the swizzled encoding tile, gemm_reg with
the previous layer's accumulators as the B operand, the ReLU sign-bit words), csrc/mlp_bwd.hip (transposed panels)
and csrc/wgrad.hip (point-contracted NT GEMM), run them through the model and compare with dense matmuls.  They pin the layout CONTRACT between the pack kernel and
the compute kernels; the GPU parity tests then pin the kernels themselves.
"""
import numpy as np

LANES = np.arange(64)
I31, HH = LANES & 31, LANES >> 5


def mfma_32x32x2(a, b, c):
    """a, b: [64] per-lane operands; c: [64,16] per-lane accumulators -> new c."""
    A = np.zeros((32, 2)); B = np.zeros((2, 32))
    A[I31, HH] = a
    B[HH, I31] = b
    D = A @ B                                   # [32,32]
    out = c.copy()
    for r in range(16):
        rows = (r & 3) + 8 * (r >> 2) + 4 * HH
        out[:, r] += D[rows, I31]
    return out


def pack_panel(Wm, col0, N, K):
    """geom.hip JOB_PANEL: P[kg][n][e] = W[n][col0 + 8kg + e], rows padded to 32, K to 8."""
    Np, KG = -(-N // 32) * 32, -(-K // 8)
    P = np.zeros((KG, Np, 8))
    for kg in range(KG):
        for e in range(8):
            k = 8 * kg + e
            if k < K:
                P[kg, :N, e] = Wm[:N, col0 + k]
    return P


def pack_panel_t(Wm, col0, N, K):
    """geom.hip JOB_PANEL_T: PT[ng][k][e] = W[8ng + e][col0 + k], rows (k) padded to 32."""
    Kp, NG = -(-K // 32) * 32, -(-N // 8)
    P = np.zeros((NG, Kp, 8))
    for ng in range(NG):
        for e in range(8):
            n = 8 * ng + e
            if n < N:
                P[ng, :K, e] = Wm[n, col0:col0 + K]
    return P


def hs_off(W, m, c):
    return m * W + ((c ^ (m & 15)) << 2)


def acc_to_dense(acc, nto):
    """C-layout -> dense [N, 32 points]."""
    out = np.zeros((32 * nto, 32))
    for t in range(nto):
        for r in range(16):
            out[32 * t + (r & 3) + 8 * (r >> 2) + 4 * HH, I31] = acc[t][:, r]
    return out


def test_mfma_model_is_a_matmul():
    rs = np.random.RandomState(0)
    A, B = rs.normal(size=(32, 2)), rs.normal(size=(2, 32))
    c = mfma_32x32x2(A[I31, HH], B[HH, I31], np.zeros((64, 16)))
    assert np.allclose(acc_to_dense([c], 1), A @ B)


def pack_bias_group(P, bias):
    """geom.hip: forward panels of biased layers carry one more group, P[KG][n][0] = bias[n]."""
    extra = np.zeros((1,) + P.shape[1:])
    extra[0, :len(bias), 0] = bias
    return np.concatenate([P, extra], 0)


def a_piece(P, kg, t):
    """a_load: lane (i, hh) reads 16 bytes at ((kg*Np + 32t + i)*8 + 4hh) floats of the panel."""
    flat, NP = P.reshape(-1), P.shape[1]
    base = ((kg * NP + 32 * t + I31) * 8 + 4 * HH)
    return np.stack([flat[o:o + 4] for o in base])


def bias_step(acc, P, KG, nto):
    """the extra MFMA per tile against the constant B operand (1 for half-wave 0, 0 for half-wave 1)."""
    one = (HH == 0).astype(np.float64)
    for t in range(nto):
        acc[t] = mfma_32x32x2(a_piece(P, KG, t)[:, 0], one, acc[t])
    return acc


def gemm_reg(P, X, nti, nto):
    """mlp_common.hpp gemm_reg: B(kg, j) = X[kg>>2][:, 4(kg&3) + j] — the C layout of the producing layer."""
    acc = [np.zeros((64, 16)) for _ in range(nto)]
    for kg in range(4 * nti):
        for t in range(nto):
            a4 = a_piece(P, kg, t)
            for j in range(4):
                acc[t] = mfma_32x32x2(a4[:, j], X[kg >> 2][:, 4 * (kg & 3) + j], acc[t])
    return acc


def enc_byte_off(m, hh, kg):
    """gemm_lds: byte address of chunk 2kg+hh of row m == lbase ^ (kg << 5)."""
    return (m * 256 + ((hh ^ (m & 15)) << 4)) ^ (kg << 5)


def test_forward_layer_chain():
    """layer 0 from the swizzled encoding tile (gemm_lds), then two layers with the accumulators as the B operand
    (gemm_reg): == W3 relu(W2 relu(W1 x + b1) + b2) + b3, no LDS round trip in between."""
    rs = np.random.RandomState(1)
    W = 64; nt = W // 32; K0 = 27; K0p = 32
    x = rs.normal(size=(K0, 32))                      # gamma(x)^T, 32 points
    W1, b1 = rs.normal(size=(W, K0 + 5)), rs.normal(size=W)   # uses columns 5.. (col0 offset like the skip layer)
    W2, b2 = rs.normal(size=(W, W)), rs.normal(size=W)
    W3, b3 = rs.normal(size=(32, W)), rs.normal(size=32)      # narrower output (view branch: NTO != NTI)
    T = np.zeros(32 * 64)
    for m in range(32):                                # encode(): element k of point m, tile rows are 64 floats
        for k in range(K0p):
            T[hs_off(64, m, k >> 2) + (k & 3)] = x[k, m] if k < K0 else 0.0
    for m in range(32):                                # the XOR addressing used by gemm_lds
        for hh in range(2):
            for kg in range(8):
                assert enc_byte_off(m, hh, kg) == 4 * hs_off(64, m, 2 * kg + hh)
    P1 = pack_bias_group(pack_panel(W1, 5, W, K0), b1)
    acc = [np.zeros((64, 16)) for _ in range(nt)]
    for kg in range(K0p // 8):
        b4 = np.stack([T[enc_byte_off(m, hh, kg) // 4: enc_byte_off(m, hh, kg) // 4 + 4] for m, hh in zip(I31, HH)])
        for t in range(nt):
            a4 = a_piece(P1, kg, t)
            for j in range(4):
                acc[t] = mfma_32x32x2(a4[:, j], b4[:, j], acc[t])
    acc = bias_step(acc, P1, K0p // 8, nt)
    h1 = np.maximum(W1[:, 5:] @ x + b1[:, None], 0)
    X = [np.maximum(a, 0) for a in acc]                # relu_bits: in place on the registers
    assert np.allclose(acc_to_dense(X, nt), h1)
    P2 = pack_bias_group(pack_panel(W2, 0, W, W), b2)
    Y = bias_step(gemm_reg(P2, X, nt, nt), P2, W // 8, nt)
    h2 = np.maximum(W2 @ h1 + b2[:, None], 0)
    Y = [np.maximum(a, 0) for a in Y]
    assert np.allclose(acc_to_dense(Y, nt), h2)
    P3 = pack_bias_group(pack_panel(W3, 0, 32, W), b3)
    V = bias_step(gemm_reg(P3, Y, nt, 1), P3, W // 8, 1)
    assert np.allclose(acc_to_dense(V, 1), W3 @ h2 + b3[:, None])


def test_relu_sign_bit_words():
    """relu_bits / mask_bits (mlp_common.hpp): per dword (tiles 2d, 2d+1) two fp32 bit streams acc = 2*acc + flag — even
    registers in the upper 16 bits, odd ones in the lower 16 (8 + 8 left-aligned when the dword holds a single tile);
    mask_bits consumes the word MSB first in the same element order."""
    rs = np.random.RandomState(5)
    for nt in (1, 2, 4, 8):
        x = rs.normal(size=(nt, 16)).astype(np.float32)
        nd = (nt + 1) // 2
        words = []
        for d in range(nd):
            acc = np.zeros(2, np.float32)
            tiles = [t for t in (2 * d, 2 * d + 1) if t < nt]
            for t in tiles:
                for r in range(0, 16, 2):
                    flag = (x[t, r:r + 2] > 0).astype(np.float32)        # clamp(x * inf)
                    acc = acc * np.float32(2) + flag                      # v_pk_fma_f32, exact below 2^24
            single = len(tiles) == 1
            words.append(((int(acc[0]) << (24 if single else 16)) | (int(acc[1]) << (16 if single else 0))) & 0xffffffff)
        g = np.ones((nt, 16))
        w = list(words)
        for d in range(nd):                                               # v_add_co (carry = top bit) + v_cndmask
            for par in range(2):
                for t in (2 * d, 2 * d + 1):
                    if t >= nt:
                        continue
                    for r in range(par, 16, 2):
                        carry = w[d] >> 31
                        w[d] = (w[d] << 1) & 0xffffffff
                        g[t, r] = g[t, r] if carry else 0.0
        assert np.array_equal(g, (x > 0).astype(np.float64))


def test_lds_swizzle_is_conflict_free():
    """ds_read_b128 is served in 16-lane groups; each group must touch 16 distinct 16-byte bank slots
    (bank slot = (byte_addr/16) % 16).  Same for the 8-lane groups of ds_write_b128."""
    groups_r = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for W in (64,):                                     # the encoding tiles: 32 points x 64 floats
        for c in range(W // 4):
            for g in groups_r:
                slots = {(hs_off(W, m, c) // 4) % 16 for m in g}
                assert len(slots) == 16
            for g0 in range(0, 32, 8):
                slots = {(hs_off(W, m, c) // 4) % 16 for m in range(g0, g0 + 8)}
                assert len(slots) == 8
        # bijective within a row
        for m in range(32):
            assert sorted(hs_off(W, m, c) for c in range(W // 4)) == [m * W + 4 * c for c in range(W // 4)]


def test_dgrad_transposed_panel():
    """mlp_bwd.hip: dIn^T[K x 32] = W^T . dZ^T with the transposed panel as the A operand and the gradient's
    C-layout registers as B."""
    rs = np.random.RandomState(2)
    N, K = 64, 96                                       # W is [N out, K in]; dgrad contracts over n
    Wm, dZ = rs.normal(size=(N, K + 3)), rs.normal(size=(N, 32))
    X = []
    for t in range(N // 32):                            # dZ in C layout: lane (m, hh), reg r <-> n = 32t + 8(r>>2) + 4hh + (r&3)
        a = np.zeros((64, 16))
        for r in range(16):
            a[:, r] = dZ[32 * t + (r & 3) + 8 * (r >> 2) + 4 * HH, I31]
        X.append(a)
    PT = pack_panel_t(Wm, 3, N, K)                     # [N/8][Kp][8]
    acc = gemm_reg(PT, X, N // 32, K // 32)
    assert np.allclose(acc_to_dense(acc, K // 32), Wm[:, 3:].T @ dZ)


def test_tile_major_storage_and_wgrad_lds_image():
    """common.hpp tile-major storage + wgrad.hip LDS image: a producing lane (m, hh) writes 16 bytes at
    tm_col(c0) + m*32 + hh*16 of its workgroup's tile row; the wgrad DMA copies each 1 KiB block to o*OCTF floats;
    lane (i, hh) of step st then reads column 32x + i of point 2st + hh at lbase + 16 st + 4x*OCTF — every half-wave
    on 32 distinct banks."""
    OCTF, rows = 264, 96
    rs = np.random.RandomState(7)
    A = rs.normal(size=(32, rows))                           # one 32-point tile row, logical [m][c]
    mem = np.zeros(32 * rows)
    for t in range(rows // 32):                              # producer: TileStores of a block starting at column 32t
        for q in range(4):
            for m in range(32):
                for hh in range(2):
                    byte = ((32 * t) >> 3) * 1024 + (4 * 0 + q) * 1024 + m * 32 + hh * 16
                    mem[byte // 4: byte // 4 + 4] = A[m, 32 * t + 8 * q + 4 * hh: 32 * t + 8 * q + 4 * hh + 4]
    assert np.array_equal(mem.reshape(rows // 8, 32, 8).transpose(1, 0, 2).reshape(32, rows), A)
    lds = np.zeros(32 * OCTF)
    for o in range(rows // 8):                               # DMA: block o -> slot o*OCTF, lane-linear
        lds[o * OCTF: o * OCTF + 256] = mem[o * 256: (o + 1) * 256]
    for st in range(16):
        for x in range(rows // 32):
            for hh in range(2):
                addr = [(i >> 3) * OCTF + hh * 8 + (i & 7) + 16 * st + 4 * x * OCTF for i in range(32)]
                assert len({a % 32 for a in addr}) == 32       # conflict-free half-wave
                assert np.array_equal(lds[addr], A[2 * st + hh, 32 * x: 32 * x + 32])


def test_wgrad_point_contraction():
    """wgrad.hip: dW[n][k] = sum_m X^T[n][m] Y^T[k][m]; both operands are 16-byte pieces along m taken
    at column 8*step + 4*hh, the 4 components feed 4 MFMAs."""
    rs = np.random.RandomState(3)
    M = 64
    XT, YT = rs.normal(size=(32, M)), rs.normal(size=(32, M))
    acc = np.zeros((64, 16))
    for step in range(M // 8):
        a4 = np.stack([XT[i, 8 * step + 4 * hh: 8 * step + 4 * hh + 4] for i, hh in zip(I31, HH)])
        b4 = np.stack([YT[i, 8 * step + 4 * hh: 8 * step + 4 * hh + 4] for i, hh in zip(I31, HH)])
        for j in range(4):
            acc = mfma_32x32x2(a4[:, j], b4[:, j], acc)
    assert np.allclose(acc_to_dense([acc], 1), XT @ YT.T)


# ---- opt-in bf16-plane inference forward (csrc/mlp_fwd_bf.hip) -------------------------------------------------------------
def bf16_rne(x):
    """round-to-nearest-even bf16 of float32 values, returned as float32 (pack_bf_k / v_cvt_pk_bf16_f32)."""
    u = np.asarray(x, np.float32).view(np.uint32)
    h = ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint32)
    return (h << 16).view(np.float32)


def split_planes(x, NP):
    out, r = [], np.asarray(x, np.float32)
    for _ in range(NP):
        p = bf16_rne(r)
        out.append(p)
        r = (r - p).astype(np.float32)        # exact: what this plane left over
    return out


def mfma_32x32x16(a, b, c):
    """v_mfma_f32_32x32x16_bf16: a, b [64, 8] per-lane operands (lane l: row / column l&31, k = 8 (l>>5) + e); c [64, 16]."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for e in range(8):
        A[I31, 8 * HH + e] = a[:, e]
        B[8 * HH + e, I31] = b[:, e]
    D = A @ B
    out = c.copy()
    for r in range(16):
        out[:, r] += D[(r & 3) + 8 * (r >> 2) + 4 * HH, I31]
    return out


def test_bf16_plane_layer_chain():
    """The layout contract of mlp_fwd_bf.hip: K-step s = (tile s>>1, half s&1) of a register-operand GEMM takes registers
    r = 8 half + e of the previous layer's C-layout tile as the 8 bf16 of lane (m, hh), and pack_bf_k stores the weights in
    the matching k order 32t + 16 half + 8(e>>2) + 4hh + (e&3); LDS-operand panels use k = 16s + 8hh + e.  Two chained
    layers through the lane model with 2 planes (3 cross terms) against dense fp64 matmuls, and the error ordering of
    1 / 2 / 3 planes."""
    rs = np.random.RandomState(3)
    W_, K0 = 64, 32                               # a 64-wide layer fed by a 32-channel "encoding", then 64 -> 64
    W0 = rs.normal(size=(W_, K0)).astype(np.float32) * 0.3
    W1 = rs.normal(size=(W_, W_)).astype(np.float32) * 0.2
    X = rs.normal(size=(32, K0)).astype(np.float32)          # 32 points
    ref1 = np.maximum(X.astype(np.float64) @ W0.T.astype(np.float64), 0) @ W1.T.astype(np.float64)
    errs = {}
    for NP in (1, 2, 3):
        prods = [(i, s_ - i) for s_ in range(NP - 1, -1, -1) for i in range(s_ + 1)]
        Wp0, Wp1 = split_planes(W0, NP), split_planes(W1, NP)
        NT = W_ // 32
        # layer 0: B operand from the "LDS tile": lane (m, hh), element e of K-step s = channel 16s + 8hh + e
        acc = [np.zeros((64, 16)) for _ in range(NT)]
        for s_ in range(K0 // 16):
            ch = 16 * s_ + 8 * HH[:, None] + np.arange(8)[None, :]
            bpl = split_planes(X[I31[:, None], ch], NP)
            for t in range(NT):
                apl = [Wp0[p][32 * t + I31[:, None], ch] for p in range(NP)]     # pack kind 0
                for i, j in prods:
                    acc[t] = mfma_32x32x16(apl[i], bpl[j], acc[t])
        # layer 1: B operand = relu of layer 0's accumulators, 8 registers at a time
        out = [np.zeros((64, 16)) for _ in range(NT)]
        for s_ in range(2 * NT):
            t_in, half = s_ >> 1, s_ & 1
            e = np.arange(8)[None, :]
            xs = np.maximum(acc[t_in][:, 8 * half:8 * half + 8], 0).astype(np.float32)
            bpl = split_planes(xs, NP)
            feat = 32 * t_in + 16 * half + 8 * (e >> 2) + 4 * HH[:, None] + (e & 3)   # pack kind 1
            # the C layout puts exactly that feature in register 8 half + e of lane (m, hh)
            r = 8 * half + e
            assert np.array_equal(feat, 32 * t_in + (r & 3) + 8 * (r >> 2) + 4 * HH[:, None])
            for t in range(NT):
                apl = [Wp1[p][32 * t + I31[:, None], feat] for p in range(NP)]
                for i, j in prods:
                    out[t] = mfma_32x32x16(apl[i], bpl[j], out[t])
        got = np.zeros((32, W_))
        for t in range(NT):
            for r in range(16):
                got[I31, 32 * t + (r & 3) + 8 * (r >> 2) + 4 * HH] = out[t][:, r]
        errs[NP] = np.abs(got - ref1).max() / np.abs(ref1).max()
    assert errs[1] < 3e-2 and errs[2] < 2e-4 and errs[3] < 2e-6, errs
    assert errs[1] > 20 * errs[2] > 400 * errs[3] / 20 or errs[3] < 1e-7, errs
