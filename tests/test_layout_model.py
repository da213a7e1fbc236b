"""Lane-accurate numpy model of the MFMA data layouts the HIP kernels rely on (CPU only).

`mfma_32x32x2` below is the documented gfx950 semantics of v_mfma_f32_32x32x2_f32
(A: lane l holds A[i=l&31][k=l>>5]; B: lane l holds B[k=l>>5][j=l&31]; C/D: lane l, reg r holds
D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31]).  The tests re-state, per lane, the address arithmetic of
csrc/geom.hip (panels), csrc/mlp_fwd.hip (gemm_seg / init_bias / park / LDS swizzle),
csrc/mlp_bwd.hip (transposed panels) and csrc/wgrad.hip (point-contracted NT GEMM), run them through
the model and compare with dense matmuls.  They pin the layout CONTRACT between the pack kernel and
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


def lds_store_tile(Hs, W, acc, nt):
    """park(): lane (m,hh) writes float4 (acc[t][4q..4q+3]) at chunk 8t+2q+hh of row m."""
    for t in range(nt):
        for q in range(4):
            for lane in LANES:
                m, hh = lane & 31, lane >> 5
                o = hs_off(W, m, 8 * t + 2 * q + hh)
                Hs[o:o + 4] = acc[t][lane, 4 * q:4 * q + 4]


def gemm_seg(acc, P, Hs, W, KG, nto):
    """mlp_fwd.hip gemm_seg: A = 16-byte piece of the panel, B = 16-byte chunk 2kg+hh of the LDS row."""
    flat = P.reshape(-1)
    NP = P.shape[1]
    for kg in range(KG):
        b4 = np.stack([Hs[hs_off(W, m, 2 * kg + hh):hs_off(W, m, 2 * kg + hh) + 4] for m, hh in zip(I31, HH)])
        for t in range(nto):
            base = ((kg * NP + 32 * t + I31) * 8 + 4 * HH)
            a4 = np.stack([flat[o:o + 4] for o in base])
            for j in range(4):
                acc[t] = mfma_32x32x2(a4[:, j], b4[:, j], acc[t])
    return acc


def init_bias(bias, nto):
    acc = [np.zeros((64, 16)) for _ in range(nto)]
    for t in range(nto):
        for q in range(4):
            for j in range(4):
                acc[t][:, 4 * q + j] = bias[32 * t + 8 * q + 4 * HH + j]
    return acc


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


def test_forward_layer_chain():
    """two layers through panels + swizzled LDS tile == relu(W2 relu(W1 x + b1) + b2)."""
    rs = np.random.RandomState(1)
    W = 64; nt = W // 32; K0 = 27; K0p = 32
    x = rs.normal(size=(K0, 32))                      # gamma(x)^T, 32 points
    W1, b1 = rs.normal(size=(W, K0 + 5)), rs.normal(size=W)   # uses columns 5.. (col0 offset like the skip layer)
    W2, b2 = rs.normal(size=(W, W)), rs.normal(size=W)
    Hs = np.zeros(32 * W)
    for m in range(32):                                # encode(): element k of point m
        for k in range(K0p):
            Hs[hs_off(W, m, k >> 2) + (k & 3)] = x[k, m] if k < K0 else 0.0
    acc = gemm_seg(init_bias(b1, nt), pack_panel(W1, 5, W, K0), Hs, W, K0p // 8, nt)
    h1 = np.maximum(W1[:, 5:] @ x + b1[:, None], 0)
    assert np.allclose(np.maximum(acc_to_dense(acc, nt), 0), h1)
    for t in range(nt):
        acc[t] = np.maximum(acc[t], 0)
    lds_store_tile(Hs, W, acc, nt)
    acc2 = gemm_seg(init_bias(b2, nt), pack_panel(W2, 0, W, W), Hs, W, W // 8, nt)
    assert np.allclose(acc_to_dense(acc2, nt), W2 @ h1 + b2[:, None])


def test_lds_swizzle_is_conflict_free():
    """ds_read_b128 is served in 16-lane groups; each group must touch 16 distinct 16-byte bank slots
    (bank slot = (byte_addr/16) % 16).  Same for the 8-lane groups of ds_write_b128."""
    groups_r = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
                [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    for W in (64, 128, 256):
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
    """mlp_bwd.hip: dIn^T[K x 32] = W^T . dZ^T with the transposed panel as the A operand."""
    rs = np.random.RandomState(2)
    N, K = 64, 96                                       # W is [N out, K in]; dgrad contracts over n
    Wm, dZ = rs.normal(size=(N, K + 3)), rs.normal(size=(N, 32))
    Wd = 96
    Hs = np.zeros(32 * Wd)
    for m in range(32):
        for n in range(N):
            Hs[hs_off(Wd, m, n >> 2) + (n & 3)] = dZ[n, m]
    PT = pack_panel_t(Wm, 3, N, K)                     # [N/8][Kp][8]
    acc = gemm_seg([np.zeros((64, 16)) for _ in range(K // 32)], PT, Hs, Wd, N // 8, K // 32)
    assert np.allclose(acc_to_dense(acc, K // 32), Wm[:, 3:].T @ dZ)


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
