"""Philox4x32-10 in numpy and the dropout mask built on it (TEST INFRASTRUCTURE, see oracle/__init__.py).

The product's dropout (csrc/elementwise.hip re2e_dropout) keeps element i when word (i & 3) of
Philox4x32-10(counter = {i >> 2 (low 32 bits), i >> 34, call, 0}, key = {seed low, seed high}) is >= floor(p * 2^32) and scales the
kept values by 1 / (1 - p).  Algorithm: Salmon et al., "Parallel random numbers: as easy as 1, 2, 3" (SC'11), the 10-round
4x32 variant with multipliers 0xD2511F53 / 0xCD9E8D57 and Weyl key increments 0x9E3779B9 / 0xBB67AE85 -- pinned below
against the known-answer vectors published with Random123 (kat_vectors: philox4x32-10)."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(counter, key):
    """counter: (n, 4) uint32, key: (2,) uint32 -> (n, 4) uint32"""
    c = [counter[:, i].astype(np.uint64) for i in range(4)]
    k0, k1 = int(key[0]), int(key[1])
    mask = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        n0 = (p1 >> np.uint64(32)) ^ c[1] ^ np.uint64(k0)
        n1 = p1 & mask
        n2 = (p0 >> np.uint64(32)) ^ c[3] ^ np.uint64(k1)
        n3 = p0 & mask
        c = [n0 & mask, n1, n2 & mask, n3]
        k0, k1 = (k0 + W0) & 0xFFFFFFFF, (k1 + W1) & 0xFFFFFFFF
    return np.stack(c, 1).astype(np.uint32)


def dropout_mask(n, p, seed, call):
    """float32 array of n entries: 0 or 1/(1-p), exactly the mask re2e_dropout applies for (seed, call)."""
    n4 = (n + 3) // 4
    q = np.arange(n4, dtype=np.uint64)
    ctr = np.stack([(q & np.uint64(0xFFFFFFFF)).astype(np.uint32), (q >> np.uint64(32)).astype(np.uint32),
                    np.full(n4, call, np.uint32), np.zeros(n4, np.uint32)], 1)
    words = philox4x32_10(ctr, np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], np.uint32)).reshape(-1)[:n]
    t = float(p) * 4294967296.0
    thr = np.uint32(0xFFFFFFFF) if t >= 4294967295.0 else np.uint32(int(t))
    return np.where(words >= thr, np.float32(1.0) / (np.float32(1.0) - np.float32(p)), np.float32(0.0)).astype(np.float32)
