"""CPU: the slot arithmetic of the varlen flat work list (csrc/fa_common.h: flat_owner / decode_work_flat),
restated in numpy.  Invariants the kernels rely on: slot starts are strictly increasing, every sequence owns at
least ceil(len / block) slots and at most one more, every slot has exactly one owner, and the grid size is
floor(total / block) + batch."""
import numpy as np
import pytest


def starts(cu, block):
    b = np.arange(len(cu) - 1)
    return cu[:-1] // block + b


def owner(f, cu, block):
    """What flat_owner computes with its ballots: the last sequence whose first slot is <= f."""
    st = starts(cu, block)
    b = int(np.sum(st <= f)) - 1
    return b, f - int(st[b]) if b >= 0 else -1


@pytest.mark.parametrize("block", [128, 256])
@pytest.mark.parametrize("seed", range(6))
def test_flat_slots_cover_every_block_exactly_once(block, seed):
    rng = np.random.default_rng(seed)
    B = int(rng.integers(1, 200))
    lens = rng.integers(0, 5 * block, size=B)
    lens[rng.integers(0, B, size=max(1, B // 8))] = 0
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    total = int(cu[-1])
    F = total // block + B
    st = starts(cu, block)
    assert np.all(np.diff(st) >= 1)                                  # strictly increasing
    ends = np.concatenate([st[1:], [F]])
    need = -(-lens // block)
    assert np.all(ends - st >= need) and np.all(ends - st <= need + 1)
    seen = set()
    for f in range(F):
        b, blk = owner(f, cu, block)
        assert 0 <= b < B and 0 <= blk < ends[b] - st[b]
        if blk < need[b]:
            seen.add((b, blk))
    assert len(seen) == int(need.sum())                              # every real block has a slot
