"""Deterministic synthetic inputs for BASELINE.json's configs.

Pure numpy; no reference code, no oracle.  World convention follows the
reference (tile.c:547-592): chunk = 256 world units, nav cell = 4 wu, world X
grows to the LEFT, map centred on the origin:
    map_pos = (W*128, 0, -H*128);  cell (R, C) covers
    x in (map_pos.x - 4*(C+1), map_pos.x - 4*C],  z in [map_pos.z + 4*R, map_pos.z + 4*(R+1)).
"""
import numpy as np

COST_IMPASSABLE = 0xFF
CHUNK_WU = 256.0
CELL_WU = 4.0


def map_pos(w, h):
    return np.array([w * CHUNK_WU / 2.0, 0.0, -h * CHUNK_WU / 2.0], np.float32)


def cost_grid(w, h, seed=1234, frac_impassable=0.20):
    """[h*64, w*64] u8 cost grid: seeded axis-aligned rectangles until ~20 % of cells are
    impassable (0xff), everything else cost 1; 1-cell map border passable; only the largest
    4-connected island kept (others filled impassable)."""
    from scipy import ndimage
    rng = np.random.RandomState(seed)
    R, Cc = h * 64, w * 64
    imp = np.zeros((R, Cc), bool)
    target = frac_impassable * R * Cc
    while imp.sum() < target:
        rh, rw = rng.randint(2, 13), rng.randint(2, 13)
        r0, c0 = rng.randint(1, R - rh - 1), rng.randint(1, Cc - rw - 1)
        imp[r0:r0 + rh, c0:c0 + rw] = True
    imp[0, :] = imp[-1, :] = False
    imp[:, 0] = imp[:, -1] = False
    lab, n = ndimage.label(~imp)          # 4-connectivity by default
    if n > 1:
        sizes = ndimage.sum(~imp, lab, index=np.arange(1, n + 1))
        keep = 1 + int(np.argmax(sizes))
        imp |= (lab != keep)
    return np.where(imp, COST_IMPASSABLE, 1).astype(np.uint8)


def to_chunks(grid):
    """[h*64, w*64] -> [h, w, 64, 64] (the N_CopyCostBasePacked layout, nav.c:2432)."""
    R, Cc = grid.shape
    h, w = R // 64, Cc // 64
    return np.ascontiguousarray(grid.reshape(h, 64, w, 64).transpose(0, 2, 1, 3))


def from_chunks(planes):
    h, w = planes.shape[:2]
    return np.ascontiguousarray(planes.transpose(0, 2, 1, 3).reshape(h * 64, w * 64))


def cell_centre(w, h, R, C):
    mp = map_pos(w, h)
    x = mp[0] - CELL_WU * (np.asarray(C, np.float32) + 0.5)
    z = mp[2] + CELL_WU * (np.asarray(R, np.float32) + 0.5)
    return np.stack([x, z], -1).astype(np.float32)


def passable_cells(grid, blockers=None):
    ok = grid != COST_IMPASSABLE
    if blockers is not None:
        ok &= blockers == 0
    return np.argwhere(ok)


def destinations(grid, k, seed=42):
    """K distinct passable global cells (R, C), uniform."""
    rng = np.random.RandomState(seed)
    cells = passable_cells(grid)
    idx = rng.choice(len(cells), size=k, replace=False)
    return cells[idx]


def agents(grid, n, k_flocks, seed=7, radius=1.0, max_speed=20.0, hz=20):
    """N agents at random passable cell centres + U(-1.5,1.5) jitter, round-robin flocks."""
    rng = np.random.RandomState(seed)
    h, w = grid.shape[0] // 64, grid.shape[1] // 64
    cells = passable_cells(grid)
    idx = rng.randint(0, len(cells), size=n)
    pos = cell_centre(w, h, cells[idx, 0], cells[idx, 1])
    pos += rng.uniform(-1.5, 1.5, size=pos.shape).astype(np.float32)
    vel = rng.normal(0.0, 0.35, size=pos.shape).astype(np.float32)
    return {
        "pos": pos.astype(np.float32),
        "vel": vel,
        "radius": np.full(n, radius, np.float32),
        "max_speed": np.full(n, max_speed, np.float32),
        "speed": np.full(n, max_speed, np.float32),
        "flock": (np.arange(n) % k_flocks).astype(np.int32),
        "hz": hz,
    }
