"""CPU: the block id -> (unit, item) arithmetic of the dense grids (csrc/fa_common.h: unit_grid / decode_unit_item), restated in
Python.  Invariants the kernels rely on: every (unit, item) is owned by exactly one block id; ids past the real work are marked
invalid; units of full rounds of eight sit on XCD unit % 8 with all of their items; the 1-7 units of the last round use all eight
XCDs with run lengths that differ by at most one item (id % 8 is where a block is observed to land)."""
import pytest


def unit_grid(units, per_unit):
    full8, tail = units & ~7, units & 7
    return full8 * per_unit + (8 * ((tail * per_unit + 7) // 8) if tail else 0)


def decode_unit_item(i, units, per_unit):
    full8 = units & ~7
    head_ids = full8 * per_unit
    if i < head_ids:
        xcd, j = i & 7, i >> 3
        ul = j // per_unit
        return ul * 8 + xcd, j - ul * per_unit, True
    tail_items = (units - full8) * per_unit
    i2 = i - head_ids
    x = i2 & 7
    lo, hi = (tail_items * x) >> 3, (tail_items * (x + 1)) >> 3
    lin = lo + (i2 >> 3)
    valid = lin < hi
    u = lin // per_unit
    return full8 + u, lin - u * per_unit, valid


@pytest.mark.parametrize("units", [1, 2, 3, 4, 5, 7, 8, 9, 12, 13, 16, 23, 24, 128])
@pytest.mark.parametrize("per_unit", [1, 2, 3, 7, 16, 33, 128])
def test_every_item_has_exactly_one_block(units, per_unit):
    grid = unit_grid(units, per_unit)
    seen = {}
    per_xcd = [0] * 8
    for i in range(grid):
        u, it, ok = decode_unit_item(i, units, per_unit)
        if not ok:
            continue
        assert 0 <= u < units and 0 <= it < per_unit
        assert (u, it) not in seen
        seen[(u, it)] = i & 7
        per_xcd[i & 7] += 1
    assert len(seen) == units * per_unit
    assert grid - len(seen) < 8                                          # at most seven idle blocks, none of them a whole XCD's share
    full8 = units & ~7
    for (u, it), x in seen.items():
        if u < full8:
            assert x == u % 8                                            # full rounds: the unit's items share one XCD (one L2)
    # the last round's items: spread over the XCDs in runs that differ by at most the rounding of the cut
    tail = [x for (u, it), x in seen.items() if u >= full8]
    if tail:
        cnt = [tail.count(x) for x in range(8)]
        chunk = -(-len(tail) // 8)
        assert max(cnt) == chunk and sum(cnt) == len(tail)
        assert max(cnt) - min(cnt) <= 1                                   # no XCD empty while others hold two items
        # an XCD's run is contiguous in (unit, item): it touches at most two units when a unit has >= chunk items
        for x in range(8):
            us = {u for (u, it), xx in seen.items() if u >= full8 and xx == x}
            if per_unit >= chunk:
                assert len(us) <= 2


def test_full_rounds_keep_the_round_4_grid():
    """BASELINE config 2 (128 units) and every multiple of eight: the grid and the placement of rounds 1-4."""
    for units, per_unit in ((128, 16), (8, 64), (64, 8)):
        assert unit_grid(units, per_unit) == 8 * (units // 8) * per_unit
        for i in range(0, unit_grid(units, per_unit), 37):
            u, it, ok = decode_unit_item(i, units, per_unit)
            j = i >> 3
            assert ok and u == (j // per_unit) * 8 + (i & 7) and it == j % per_unit
