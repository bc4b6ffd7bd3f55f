"""The grid selection of Reprojector::reprojectMap on the device (hso_gpu_reproject_select) against a sequential
restatement of src/reprojector.cpp:253-306 (the branch and the three passes), :352-429 (reprojectCell) and :556-612
(reprojectCellAll) written the reference's way: lists per cell, stable sort on first visit, erase while walking."""
import numpy as np
import pytest


def select_reference(cell, quality, flags, cell_order, n_cells, max_fts):
    n = len(cell)
    out, n_matches = [], 0
    matched = lambda i: (flags[i] & 3) == 1
    if n == 0 or max_fts <= 0:
        return out, 0, 0
    if n < max_fts + 50:                               # reprojectCellAll, :556-612
        for i in range(n):
            out.append([i, False])
            if flags[i] & 2 or not matched(i):
                continue
            out[-1][1] = True
            n_matches += 1
            if n_matches >= max_fts:
                break
        return out, n_matches, 0
    cells = [[] for _ in range(n_cells)]
    for i in range(n):
        cells[cell[i]].append(i)
    state = {"m": 0}

    def visit(c, is_2nd, is_3rd):                      # reprojectCell, :352-429
        L = cells[c]
        if not L:
            return False
        if not is_2nd:
            L.sort(key=lambda i: -int(quality[i]))     # list.sort is stable, like std::list::sort
        success = 0
        while L:
            i = L.pop(0)
            out.append([i, False])
            if flags[i] & 2 or not matched(i):
                continue
            out[-1][1] = True
            if not is_3rd:
                return True
            success += 1
            state["m"] += 1
            if success >= 3 or state["m"] >= max_fts:
                return True
        return False

    passes = 1
    for k in range(n_cells):                           # :268-278
        if visit(cell_order[k], False, False):
            state["m"] += 1
        if state["m"] >= max_fts:
            break
    if state["m"] < max_fts:                           # :281-293
        passes = 2
        for k in range(n_cells - 1, 0, -1):
            if visit(cell_order[k], True, False):
                state["m"] += 1
            if state["m"] >= max_fts:
                break
        if state["m"] < max_fts:                       # :296-305
            passes = 3
            for k in range(n_cells):
                visit(cell_order[k], True, True)
                if state["m"] >= max_fts:
                    break
    return out, state["m"], passes


def make_frame(rng, n, n_cells, p_match, p_deleted, clustered):
    if clustered:
        centres = rng.integers(0, n_cells, size=max(n_cells // 6, 1))
        cell = centres[rng.integers(0, len(centres), size=n)]
    else:
        cell = rng.integers(0, n_cells, size=n)
    ptype = rng.integers(1, 5, size=n)                 # TEMPORARY .. GOOD
    ftype = rng.integers(0, 3, size=n)
    deleted = rng.random(n) < p_deleted
    ptype[deleted] = 0
    quality = (ptype << 4) | ftype
    flags = (rng.random(n) < p_match).astype(np.uint8) | (deleted.astype(np.uint8) << 1)
    return cell.astype(np.int32), quality.astype(np.uint8), flags.astype(np.uint8)


def test_reference_restatement_basics():
    """hand cases of the sequential walk: the better type goes first, ties keep projection order, a deleted point costs a trial"""
    cell = np.zeros(60, np.int32); cell[:] = np.arange(60) % 3
    quality = np.full(60, (3 << 4) | 1, np.uint8); quality[4] = (4 << 4) | 0   # candidate 4 (cell 1) is TYPE_GOOD
    flags = np.ones(60, np.uint8); flags[1] = 3                                 # candidate 1 (cell 1) deleted
    out, m, passes = select_reference(cell, quality, flags, [0, 1, 2], 3, 2)
    assert out[0] == [0, True] and out[1] == [4, True] and m == 2 and passes == 1   # cell 1: the GOOD point jumps the queue


@pytest.mark.gpu
def test_reproject_select_matches_the_sequential_walk(gpu_ctx):
    rng = np.random.default_rng(5)
    frames, order_sets = [], []
    n_cells = 570                                       # 30 x 19 cells: EuRoC with 25-pixel cells
    cell_order = rng.permutation(n_cells).astype(np.int32)
    cases = [(180, 200, 0.8, 0.02, False),              # fewer than max_fts + 50: reprojectCellAll
             (240, 200, 0.95, 0.0, False),              # reprojectCellAll that meets the budget early
             (900, 200, 0.85, 0.03, False),             # pass 1 meets the budget
             (900, 600, 0.7, 0.05, False),              # passes 1 + 2
             (900, 800, 0.5, 0.05, True),               # clustered cells: pass 3 with up to three per cell
             (2500, 2000, 0.9, 0.02, True),             # 2000-feature budget, pass 3 cut inside a cell
             (400, 300, 0.0, 0.1, False),               # nothing matches: every candidate examined
             (0, 200, 0.5, 0.0, False)]                 # an empty frame among the others
    fb, cells, quals, flgs = [0], [], [], []
    for n, _, pm, pd, cl in cases:
        c, q, f = make_frame(rng, n, n_cells, pm, pd, cl)
        cells.append(c); quals.append(q); flgs.append(f); fb.append(fb[-1] + n)
    seen_passes = set()
    for budget in sorted({b for _, b, *_ in cases}):
        got, counts = gpu_ctx.reproject_select(fb, np.concatenate(cells), np.concatenate(quals), np.concatenate(flgs), cell_order, budget)
        for k, (n, _, *_rest) in enumerate(cases):
            ref, m, passes = select_reference(cells[k], quals[k], flgs[k], cell_order, n_cells, budget)
            assert [tuple(r) for r in ref] == got[k], (k, budget)
            assert (counts[k, 0], counts[k, 1], counts[k, 2]) == (len(ref), m, passes), (k, budget)
            seen_passes.add(passes)
    assert seen_passes == {0, 1, 2, 3}


@pytest.mark.gpu
def test_reproject_select_rejects_bad_tables(gpu_ctx):
    with pytest.raises(RuntimeError):
        gpu_ctx.reproject_select([0, 2], [0, 7], [16, 16], [1, 1], [0, 1, 2], 10)          # cell out of range
    with pytest.raises(RuntimeError):
        gpu_ctx.reproject_select([0, 2], [0, 1], [16, 16], [1, 1], [0, 1, 1], 10)          # cell_order not a permutation
