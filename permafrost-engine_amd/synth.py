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
    count = 0                                   # == imp.sum(), kept incrementally (big maps)
    while count < target:
        rh, rw = rng.randint(2, 13), rng.randint(2, 13)
        r0, c0 = rng.randint(1, R - rh - 1), rng.randint(1, Cc - rw - 1)
        blk = imp[r0:r0 + rh, c0:c0 + rw]
        count += blk.size - int(blk.sum())
        blk[:] = True
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


def agents(grid, n, k_flocks, seed=7, radius=1.0, max_speed=20.0, hz=20, blockers=None, cols=None,
           rows=None, crowd_cells=0):
    """N agents at random passable (and unblocked) cell centres + U(-1.5,1.5) jitter, round-robin
    flocks.  cols = (c0, c1) / rows = (r0, r1): only cells of the global columns [c0, c1) and rows
    [r0, r1) (one region of the map).  crowd_cells > 0: the crowded variant -- every flock starts
    packed into the passable cells within `crowd_cells` (Chebyshev) of a random centre, so that the
    neighbour caps of the movement tick (128 / 32 + 32) bind and ClearPath runs in its R^3 regime."""
    rng = np.random.RandomState(seed)
    h, w = grid.shape[0] // 64, grid.shape[1] // 64
    cells = passable_cells(grid, blockers)
    if cols is not None and (cols[0] > 0 or cols[1] < grid.shape[1]):
        cells = cells[(cells[:, 1] >= cols[0]) & (cells[:, 1] < cols[1])]
    if rows is not None and (rows[0] > 0 or rows[1] < grid.shape[0]):
        cells = cells[(cells[:, 0] >= rows[0]) & (cells[:, 0] < rows[1])]
    if crowd_cells > 0:
        flock_of = np.arange(n) % k_flocks
        idx = np.zeros(n, np.int64)
        for f in range(k_flocks):
            c = cells[rng.randint(len(cells))]
            near = np.flatnonzero((np.abs(cells[:, 0] - c[0]) <= crowd_cells) & (np.abs(cells[:, 1] - c[1]) <= crowd_cells))
            mine = np.flatnonzero(flock_of == f)
            idx[mine] = near[rng.randint(0, len(near), size=len(mine))]
    else:
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


# ------------------------------------------------------------------------------------------
# synthetic request streams (what the reference's host-side planner would hand the kernels)
# ------------------------------------------------------------------------------------------
def local_islands(grid, blockers=None):
    """Per-chunk 4-connected components of passable cells (cost != 0xff and blockers == 0);
    u16 ids local to each chunk, ISLAND_NONE (0xffff) elsewhere.  Any consistent labelling
    serves the kernels, which only test ids for equality (field.c:1151,1195)."""
    from scipy import ndimage
    ok = grid != COST_IMPASSABLE
    if blockers is not None:
        ok &= blockers == 0
    h, w = grid.shape[0] // 64, grid.shape[1] // 64
    out = np.full(grid.shape, 0xFFFF, np.uint16)
    for cr in range(h):
        for cc in range(w):
            sl = (slice(cr * 64, cr * 64 + 64), slice(cc * 64, cc * 64 + 64))
            lab, n = ndimage.label(ok[sl])
            o = out[sl]
            o[lab > 0] = (lab[lab > 0] - 1).astype(np.uint16)
    return out


def portals(grid):
    """Maximal runs of mutually passable facing cells on every shared chunk edge (the portal
    notion of n_link_chunks, nav.c:470-556).  Returns a list of dicts with both sides."""
    ok = grid != COST_IMPASSABLE
    h, w = grid.shape[0] // 64, grid.shape[1] // 64
    out = []

    def runs(mask):
        idx = np.flatnonzero(mask)
        if idx.size == 0:
            return []
        brk = np.flatnonzero(np.diff(idx) > 1)
        starts = np.r_[idx[0], idx[brk + 1]]
        ends = np.r_[idx[brk], idx[-1]]
        return list(zip(starts.tolist(), ends.tolist()))

    for cr in range(h):
        for cc in range(w):
            if cr + 1 < h:      # bottom edge of (cr,cc) / top edge of (cr+1,cc)
                both = ok[cr * 64 + 63, cc * 64:cc * 64 + 64] & ok[cr * 64 + 64, cc * 64:cc * 64 + 64]
                for a, b in runs(both):
                    out.append(dict(a=(cr, cc), a_ep=(63, a, 63, b), b=(cr + 1, cc), b_ep=(0, a, 0, b)))
            if cc + 1 < w:      # right edge of (cr,cc) / left edge of (cr,cc+1)
                both = ok[cr * 64:cr * 64 + 64, cc * 64 + 63] & ok[cr * 64:cr * 64 + 64, cc * 64 + 64]
                for a, b in runs(both):
                    out.append(dict(a=(cr, cc), a_ep=(a, 63, b, 63), b=(cr, cc + 1), b_ep=(a, 0, b, 0)))
    return out


REQ_FIELDS = ("layer", "type", "faction_id", "flags", "enemies", "chunk_r", "chunk_c", "tile_r",
              "tile_c", "port_r0", "port_c0", "port_r1", "port_c1", "next_r0", "next_c0",
              "next_r1", "next_c1", "next_chunk_r", "next_chunk_c", "port_iid", "next_iid")


def whole_map_requests(grid, dests, liid=None):
    """For every destination cell: one chunk-field request per chunk of the map -- TARGET_TILE
    in the destination chunk, TARGET_PORTAL (towards the neighbouring chunk that is one step
    closer in chunk-BFS distance, through the portal nearest to the straight line) elsewhere
    ('each destination expands to all chunks of the map').
    Returns a dict of equal-length int arrays keyed by REQ_FIELDS, plus 'dest' (index into dests)."""
    from collections import deque
    h, w = grid.shape[0] // 64, grid.shape[1] // 64
    if liid is None:
        liid = local_islands(grid)
    plist = portals(grid)
    by_pair = {}
    for p in plist:
        by_pair.setdefault((p["a"], p["b"]), []).append((p["a_ep"], p["b_ep"]))
        by_pair.setdefault((p["b"], p["a"]), []).append((p["b_ep"], p["a_ep"]))
    cols = {k: [] for k in REQ_FIELDS + ("dest",)}

    def push(**kw):
        for k in REQ_FIELDS:
            cols[k].append(kw.get(k, 0))
        cols["dest"].append(di)

    for di, (R, Cc) in enumerate(np.asarray(dests)):
        dchunk = (int(R) // 64, int(Cc) // 64)
        dist = {dchunk: 0}
        q = deque([dchunk])
        while q:
            cur = q.popleft()
            for d in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                nb = (cur[0] + d[0], cur[1] + d[1])
                if nb in dist or not (0 <= nb[0] < h and 0 <= nb[1] < w):
                    continue
                if (nb, cur) not in by_pair:
                    continue
                dist[nb] = dist[cur] + 1
                q.append(nb)
        for cr in range(h):
            for cc in range(w):
                ch = (cr, cc)
                if ch == dchunk:
                    push(type=1, faction_id=0xF, chunk_r=cr, chunk_c=cc,
                         tile_r=int(R) % 64, tile_c=int(Cc) % 64)
                    continue
                if ch not in dist:
                    continue
                best = None
                for d in ((-1, 0), (1, 0), (0, -1), (0, 1)):
                    nb = (cr + d[0], cc + d[1])
                    if dist.get(nb, 1 << 30) == dist[ch] - 1 and (ch, nb) in by_pair:
                        for ep, nep in by_pair[(ch, nb)]:
                            mr, mc = (ep[0] + ep[2]) / 2.0, (ep[1] + ep[3]) / 2.0
                            gr, gc = cr * 64 + mr, cc * 64 + mc
                            score = abs(gr - R) + abs(gc - Cc)
                            if best is None or score < best[0]:
                                best = (score, nb, ep, nep)
                _, nb, ep, nep = best
                push(type=0, faction_id=0xF, chunk_r=cr, chunk_c=cc,
                     port_r0=ep[0], port_c0=ep[1], port_r1=ep[2], port_c1=ep[3],
                     next_r0=nep[0], next_c0=nep[1], next_r1=nep[2], next_c1=nep[3],
                     next_chunk_r=nb[0], next_chunk_c=nb[1],
                     port_iid=int(liid[cr * 64 + ep[0], cc * 64 + ep[1]]),
                     next_iid=int(liid[nb[0] * 64 + nep[0], nb[1] * 64 + nep[1]]))
    return {k: np.asarray(v, np.int64) for k, v in cols.items()}


# ------------------------------------------------------------------------------------------
# the same request stream as the REFERENCE'S PLANNER emits it (committed fixtures)
# ------------------------------------------------------------------------------------------
def planner_requests(grid, dests):
    """The whole-map request stream of (grid, dests) as permafrost-engine's own planner chose it
    (n_request_path, nav.c:1774: its portals, its island ids) -- from the fixtures under data/ that
    tests/tools/make_requests.py generated with the reference build -- or None when there is no fixture for
    exactly this map and these destinations (then: whole_map_requests, the numpy stand-in).  Same dict of
    columns as whole_map_requests."""
    import glob
    import hashlib
    import os
    h = hashlib.sha1()
    h.update(np.ascontiguousarray(grid).tobytes())
    h.update(np.ascontiguousarray(dests, np.int64).tobytes())
    key = h.hexdigest()
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
    for path in sorted(glob.glob(os.path.join(here, "requests_*.npz"))):
        d = np.load(path)
        if str(d["key"]) != key:
            continue
        cols = {k: (d[k].astype(np.int64) if k in d.files else np.zeros(len(d["dest"]), np.int64)) for k in REQ_FIELDS}
        cols["dest"] = d["dest"].astype(np.int64)
        return cols
    return None


def planner_los(grid, dests):
    """The LOS-field chain of (grid, dests) as the reference planner builds it (N_LOSFieldCreate next to every
    flow field of a path, each from the LOS field of the chunk before it: nav.c:1843, :2026-2039) -- from the
    fixtures tests/tools/make_requests.py generated (los_cfg*.npz) -- or None.  Columns in creation order (a
    field's predecessor always comes first): dest, chunk_r, chunk_c, prev_dr, prev_dc (0, 0: the destination
    chunk's own field, no predecessor)."""
    import glob
    import hashlib
    import os
    h = hashlib.sha1()
    h.update(np.ascontiguousarray(grid).tobytes())
    h.update(np.ascontiguousarray(dests, np.int64).tobytes())
    key = h.hexdigest()
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
    for path in sorted(glob.glob(os.path.join(here, "los_*.npz"))):
        d = np.load(path)
        if str(d["key"]) == key:
            return {k: d[k].astype(np.int64) for k in ("dest", "chunk_r", "chunk_c", "prev_dr", "prev_dc")}
    return None
