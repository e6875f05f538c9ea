"""CPU model of the lane schedules k_scan_region's deep list uses for the reference's sequential float32 sum (pigo.go:137): the
DPP shift-and-add chains of seqsum_step (wave_shr:1 over 64 lanes), rowsum_step (row_shr:1 inside rows of 16 lanes: the quad pass
with four windows x 16 trees) and the 32-lane segments of the two-window quad pass (15 x row_shr on rows {0, 2}, one row_bcast:15
into rows {1, 3}, 15 x row_shr on rows {1, 3}).  Each step is modelled exactly as the instruction behaves -- every enabled lane
computes float32(x[source lane] + leaf[lane]) at once, lanes without a source lane keep their value -- and the result must be,
bit for bit, the prefix sums the reference's loop builds: ((acc + l0) + l1) + ...  This pins the SCHEDULE (which lane is final
after which step); the hardware run is pinned by tests/test_gpu_parity.py::test_region_deep_list_quad_pass."""
import numpy as np


def _step(x, leaf, src, enabled):
    """One v_add_f32_dpp: lanes with a valid source lane (src[i] >= 0) that are enabled take x[src[i]] + leaf[i], simultaneously."""
    new = x.copy()
    for i in range(64):
        if enabled[i] and src[i] >= 0:
            new[i] = np.float32(x[src[i]] + leaf[i])
    return new


def _sequential(acc, leaf):
    out, s = np.empty(len(leaf), np.float32), np.float32(acc)
    for j, v in enumerate(leaf):
        s = np.float32(s + v)  # out += treePred[...]
        out[j] = s
    return out


ROW_SHR = [i - 1 if i % 16 else -1 for i in range(64)]       # row_shr:1 -- a row's lane 0 has no source lane
WAVE_SHR = [i - 1 for i in range(64)]                         # wave_shr:1 -- lane 0 has no source lane (-1)
ROW_BCAST15 = [(i // 16) * 16 - 1 if i >= 16 else -1 for i in range(64)]  # row_bcast:15 -- lane 15 of the row below, to every lane of the row


def _rows(mask):
    return [bool(mask >> (i // 16) & 1) for i in range(64)]


def test_wave_chain_of_64_trees():
    rng = np.random.default_rng(1)
    for _ in range(50):
        leaf = (rng.standard_normal(64) * rng.choice([1e-3, 1.0, 50.0])).astype(np.float32)
        acc = np.float32(rng.standard_normal() * 3)
        x = leaf.copy()
        x[0] = np.float32(acc + leaf[0])
        for k in range(63):
            x = _step(x, leaf, WAVE_SHR, [True] * 64)
            assert np.array_equal(x[:k + 2].view(np.uint32), _sequential(acc, leaf)[:k + 2].view(np.uint32))  # lanes 0..k+1 are final
        assert np.array_equal(x.view(np.uint32), _sequential(acc, leaf).view(np.uint32))


def test_quad_pass_four_windows_of_16_trees():
    rng = np.random.default_rng(2)
    for _ in range(50):
        leaf = (rng.standard_normal(64) * 10).astype(np.float32)
        acc = (rng.standard_normal(4) * 3).astype(np.float32)
        x = leaf.copy()
        for w in range(4):
            x[16 * w] = np.float32(acc[w] + leaf[16 * w])
        for _k in range(15):
            x = _step(x, leaf, ROW_SHR, [True] * 64)
        for w in range(4):
            want = _sequential(acc[w], leaf[16 * w:16 * w + 16])
            assert np.array_equal(x[16 * w:16 * w + 16].view(np.uint32), want.view(np.uint32))


def test_quad_pass_two_windows_of_32_trees():
    rng = np.random.default_rng(3)
    for _ in range(50):
        leaf = (rng.standard_normal(64) * 10).astype(np.float32)
        acc = (rng.standard_normal(2) * 3).astype(np.float32)
        x = leaf.copy()
        for w in range(2):
            x[32 * w] = np.float32(acc[w] + leaf[32 * w])
        for _k in range(15):
            x = _step(x, leaf, ROW_SHR, _rows(0x5))       # trees 1..15 of both windows
        x = _step(x, leaf, ROW_BCAST15, _rows(0xa))       # tree 16: lane 0 of rows 1 and 3 is final, their other lanes are not yet
        for _k in range(15):
            x = _step(x, leaf, ROW_SHR, _rows(0xa))       # trees 17..31
        for w in range(2):
            want = _sequential(acc[w], leaf[32 * w:32 * w + 32])
            assert np.array_equal(x[32 * w:32 * w + 32].view(np.uint32), want.view(np.uint32))
