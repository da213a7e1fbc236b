"""numpy restatement of the counter-based uniform streams of consistentnerf_amd/csrc/rng.hpp.  TEST INFRASTRUCTURE ONLY.

The reference draws the stratified jitter (R:376) and the resampling positions (H:227) with torch.rand; which generator produces
those numbers is not part of the reference's semantics (a CPU run and a CUDA run of the reference itself already differ), only
their distribution is: independent U[0, 1).  The product generates them inside the consuming kernels from Philox4x32-10 (Salmon,
Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11 — the Random123 generator, also the one behind torch's own
CUDA streams).  Here: the same function in numpy uint64 arithmetic, PINNED on Random123's published known-answer vectors
(tests/test_host.py::test_philox_known_answers); the GPU tests compare the kernels' streams with `uniform()` bit for bit.
"""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)
KEY_XOR = 0x636e6572665f726e      # "cnerf_rn": keeps the stream disjoint from torch.rand calls on the same seed


def philox4x32_10(ctr, key):
    """ctr: 4 arrays (or ints) of 32-bit words, key: 2 ints -> 4 uint32 arrays."""
    c = [np.asarray(x, dtype=np.uint64) & MASK for x in ctr]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0), p1 & MASK, (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1), p0 & MASK]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return [x.astype(np.uint32) for x in c]


def uniform(seed, offset, rows, cols, row0=0):
    """[rows, cols] float32: element (row0 + r, c) of the global stream `offset` of `seed` (rng.hpp CnRngDev::uniform)."""
    e = (np.uint64(row0) + np.arange(rows, dtype=np.uint64))[:, None] * np.uint64(cols) + np.arange(cols, dtype=np.uint64)[None, :]
    key = (int(seed) ^ KEY_XOR) & 0xFFFFFFFFFFFFFFFF
    off = int(offset) & 0xFFFFFFFFFFFFFFFF
    x0 = philox4x32_10([e & MASK, e >> np.uint64(32), off & 0xFFFFFFFF, off >> 32], [key & 0xFFFFFFFF, key >> 32])[0]
    return ((x0 >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


PIXEL_KEY_XOR = 0x636e6572665f7078     # "cnerf_px": the pixel-draw permutation's key (csrc/sampler.hip)


def permutation(seed, offset, n, count):
    """The first `count` values pi(0), ..., pi(count - 1) of the keyed pseudo-random permutation pi of [0, n) that
    csrc/sampler.hip::perm_index draws training pixels with (V:1503 `np.random.choice(n, N_rand, replace=False)`: any prefix of a
    permutation is a draw without replacement).  pi: an alternating Feistel network on bits = max(2, ceil(log2 n)) bits — halves of
    a = bits // 2 (high) and b = bits - a (low) bits; rounds r = 0, 2, 4, 6: L ^= F(r, R) & mask_a then R ^= F(r + 1, L) & mask_b,
    F(r, v) = word 0 of Philox4x32-10 on counter (v, r, offset) under key seed ^ PIXEL_KEY_XOR — cycle-walked until the value is < n."""
    n, count = int(n), int(count)
    assert 0 <= count <= n < (1 << 31)
    if n <= 1:
        return np.zeros(count, dtype=np.int64)
    bits = 2
    while (1 << bits) < n:
        bits += 1
    a = bits // 2
    b = bits - a
    ma, mb = np.uint64((1 << a) - 1), np.uint64((1 << b) - 1)
    key = (int(seed) ^ PIXEL_KEY_XOR) & 0xFFFFFFFFFFFFFFFF
    off = int(offset) & 0xFFFFFFFFFFFFFFFF
    kk = [key & 0xFFFFFFFF, key >> 32]

    def F(r, v):
        return philox4x32_10([v & MASK, np.full_like(v, r), off & 0xFFFFFFFF, off >> 32], kk)[0].astype(np.uint64)

    x = np.arange(count, dtype=np.uint64)
    todo = np.ones(count, dtype=bool)
    while todo.any():
        v = x[todo]
        L, R = v >> np.uint64(b), v & mb
        for r in range(0, 8, 2):
            L = L ^ (F(r, R) & ma)
            R = R ^ (F(r + 1, L) & mb)
        v = (L << np.uint64(b)) | R
        x[todo] = v
        todo[todo] = v >= np.uint64(n)
    return x.astype(np.int64)
