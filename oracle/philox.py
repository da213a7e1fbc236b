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
