"""Philox-4x32-10 restatement (TEST INFRASTRUCTURE ONLY - see oracle/__init__.py).

Follows include/philox.h:13-73 of the reference (standard Random123 Philox4x32 with 10
rounds: multipliers 0xD2511F53 / 0xCD9E8D57, Weyl key steps 0x9E3779B9 / 0xBB67AE85) and
the dropout indexing of include/softmax.h:50-51,96-114.
"""
import numpy as np

_M_A = np.uint64(0xD2511F53)
_M_B = np.uint64(0xCD9E8D57)
_W_A = np.uint32(0x9E3779B9)
_W_B = np.uint32(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


def _round(c0, c1, c2, c3, k0, k1):
    # include/philox.h:19-32  philox_single_round
    p0 = _M_A * c0.astype(np.uint64)
    p1 = _M_B * c2.astype(np.uint64)
    hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
    lo0 = (p0 & _MASK32).astype(np.uint32)
    hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
    lo1 = (p1 & _MASK32).astype(np.uint32)
    return hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0


def philox4x32_10(counter, key):
    """counter: 4 uint32 arrays (broadcastable), key: 2 uint32 scalars/arrays.
    Returns 4 uint32 arrays.  include/philox.h:38-53 (9 rounds with key bump + 1)."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint32) for c in counter]
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(key[0])
    k1 = np.uint32(key[1])
    with np.errstate(over="ignore"):
        for _ in range(9):
            c0, c1, c2, c3 = _round(c0, c1, c2, c3, k0, k1)
            k0 = np.uint32((int(k0) + int(_W_A)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W_B)) & 0xFFFFFFFF)
        c0, c1, c2, c3 = _round(c0, c1, c2, c3, k0, k1)
    return c0, c1, c2, c3


def dropout_threshold(p_dropout):
    """include/softmax.h:51: uint32((1.0f - p) * 4294967295.0f) in fp32 arithmetic.
    NB 4294967295.0f rounds to 2^32 in fp32; a product >= 2^32 saturates (CUDA cvt)."""
    t = np.float32(np.float32(1.0) - np.float32(p_dropout)) * np.float32(4294967295.0)
    t = float(t)
    if t >= 4294967295.0:
        return np.uint32(0xFFFFFFFF)
    return np.uint32(int(t))


def dropout_keep_mask(seed, offset, p_dropout, rows, n_cols, row0=0, col0=0, n_glob=None):
    """Boolean keep-mask [rows, n_cols] for query rows row0.. and key cols col0..

    include/softmax.h:97-104: flat = (GLOBAL_ROW_OFFSET + row) * GLOBAL_N + col;
    state = init_philox(seed, dropout_offset + (flat >> 2)); lane = flat & 3;
    keep iff r <= thr.  Dense: no batch/head term (same mask for every (b, h))."""
    if n_glob is None:
        n_glob = n_cols
    i = (np.arange(rows, dtype=np.uint64) + np.uint64(row0))[:, None]
    j = (np.arange(n_cols, dtype=np.uint64) + np.uint64(col0))[None, :]
    flat = i * np.uint64(n_glob) + j
    ctr = np.uint64(offset) + (flat >> np.uint64(2))
    c0 = (ctr & _MASK32).astype(np.uint32)
    c1 = (ctr >> np.uint64(32)).astype(np.uint32)
    z = np.zeros_like(c0)
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    r = philox4x32_10((c0, c1, z, z), (seed & 0xFFFFFFFF, seed >> 32))
    lane = (flat & np.uint64(3)).astype(np.int64)
    rr = np.choose(lane, r)
    return rr <= dropout_threshold(p_dropout)
