"""Host-side checks of two index formulas the HIP kernels rely on (no GPU): they are restated here in numpy from the comments in
instantvnr_amd/csrc/grid_device.h (gather_corners_brick, F = 2) and render.hip (march_kernel, result slots)."""
import numpy as np


def test_brick_column_division_by_seven_is_exact_for_every_bricked_level():
    """grid_device.h: bx = (x * 9363) >> 16 replaces x / 7 for the 8 x 2 x 2 bricks with a repeated column; build_brick_image keeps
    levels of 13 000 grid points per axis or more hashed, so the identity must hold below that (it first fails at x = 13 108)"""
    x = np.arange(0, 13000 + 2, dtype=np.uint64)
    assert np.array_equal((x * 9363) >> 16, x // 7)
    bad = np.arange(13000, 70000, dtype=np.uint64)
    first_bad = int(bad[((bad * 9363) >> 16) != bad // 7][0])
    assert first_bad >= 13000 + 2


def test_ghost_column_bricks_hold_every_x_pair_in_one_line():
    """entry (x, y, z) of a level lives in brick (x // 7, y >> 1, z >> 1) at column x % 7, and column 7 repeats column 0 of the +x
    neighbour: so the pair (x, x + 1) is always columns (w, w + 1) of ONE brick, for every x of a level with res + 1 grid points"""
    for res in (16, 21, 97, 1024):
        x = np.arange(0, res, dtype=np.int64)          # left corner of a cell; right corner x + 1 <= res
        bx, w = x // 7, x % 7
        assert np.all(w + 1 <= 7)                       # the right corner is column w + 1 <= 7 of the same brick
        # the repeated column holds the same grid point as column 0 of the next brick
        assert np.array_equal((bx * 7 + 7)[w == 6], x[w == 6] + 1)
        assert bx.max() < res // 7 + 1                  # bricks per row as LevelInfo.pad1 stores them


def test_result_slots_are_unique_and_fit_the_arena():
    """render.hip: sample j of the ray in lane l of 64-ray group g is slot (g * n_iters + j) * 64 + l; the arena of a ray part holds
    n_local * n_iters slots (n_local a multiple of 64)"""
    for n_local, n_iters in ((64, 1), (640, 16), (4096, 24), (1984, 32)):
        g, j, l = np.meshgrid(np.arange(n_local // 64), np.arange(n_iters), np.arange(64), indexing="ij")
        slot = (g * n_iters + j) * 64 + l
        assert slot.max() < n_local * n_iters
        assert np.unique(slot).size == slot.size
        # the 64 rays of a group read / write sample j as one run of 64 neighbouring slots
        assert np.all(np.diff(slot, axis=2) == 1)
