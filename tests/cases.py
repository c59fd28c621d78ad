"""Shared builders of parity cases: requests + expected results from the reference oracle
(oracle/_ref, i.e. the reference's own code).  Test infrastructure only."""
import numpy as np

from oracle import pfref
from permafrost_engine_amd import synth


def ref_nav_for(w, h, seed=1234, blockers=None, frac=0.20):
    grid = synth.cost_grid(w, h, seed=seed, frac_impassable=frac)
    nav = pfref.RefNav(synth.to_chunks(grid))
    if blockers is not None:
        nav.set_blockers(blockers)
    return grid, nav


def random_blockers(grid, seed, frac=0.03):
    """u16 refcounts on ~frac of the passable cells, in chunk layout."""
    rng = np.random.RandomState(seed)
    b = np.zeros(grid.shape, np.uint16)
    m = (grid != 255) & (rng.rand(*grid.shape) < frac)
    b[m] = rng.randint(1, 4, size=int(m.sum()))
    return synth.to_chunks(b)


def tile_requests(grid, k, seed):
    """k TARGET_TILE requests on random cells (passable or not: an impassable target must
    give an all-NONE field, field.c:1118-1124)."""
    rng = np.random.RandomState(seed)
    reqs = np.zeros(k, pfref.FIELD_REQ_DTYPE)
    R = rng.randint(0, grid.shape[0], size=k)
    Cc = rng.randint(0, grid.shape[1], size=k)
    reqs["type"] = pfref.TARGET_TILE
    reqs["faction_id"] = pfref.FACTION_ID_NONE
    reqs["chunk_r"], reqs["tile_r"] = R // 64, R % 64
    reqs["chunk_c"], reqs["tile_c"] = Cc // 64, Cc % 64
    return reqs


def planner_requests(nav, grid, pairs, seed):
    """The chunk-field request stream the reference planner (n_request_path, nav.c:1774)
    emits for `pairs` random src->dst queries, with the fields it produced."""
    rng = np.random.RandomState(seed)
    h, w = grid.shape[0] // 64, grid.shape[1] // 64
    cells = synth.passable_cells(grid)
    all_reqs, all_before, all_after = [], [], []
    for _ in range(pairs):
        a, b = cells[rng.randint(len(cells))], cells[rng.randint(len(cells))]
        src = synth.cell_centre(w, h, a[0], a[1])
        dst = synth.cell_centre(w, h, b[0], b[1])
        nav.request_path(src, dst, clear_cache=True)
        reqs, before, after = nav.trace()
        all_reqs.append(reqs); all_before.append(before); all_after.append(after)
    return np.concatenate(all_reqs), np.concatenate(all_before), np.concatenate(all_after)


def ref_fields(nav, reqs, before=None, want_integ=True):
    n = len(reqs)
    dirs = np.zeros((n, 64, 64), np.uint8)
    integ = np.zeros((n, 64, 64), np.float32)
    for i in range(n):
        d, g = nav.field_update(reqs[i], inout=before[i] if (before is not None and reqs[i]["inout"]) else None,
                                want_integ=want_integ)
        dirs[i] = d
        if want_integ:
            integ[i] = g
    return dirs, integ


# ---------------------------------------------------------------------------------------------
# movement worlds
# ---------------------------------------------------------------------------------------------
def make_agents(grid, n, k_flocks, seed, clustered=True, sigma=40.0):
    """Agent snapshot as a dict of arrays named after navhip_world members.  Clustered worlds put
    every flock in a blob so that neighbour caps (32/32/128) actually bind."""
    rng = np.random.RandomState(seed)
    h, w = grid.shape[0] // 64, grid.shape[1] // 64
    # Keep agents >= 3 cells (12 wu) off the map edge: nullify_impass_components (movement.c:1839)
    # probes pos +- 4 wu, and for an off-map probe the reference reads an uninitialised
    # tile_desc (the assert at nav.c:4062 is compiled out under NDEBUG) -- undefined behaviour
    # there is not a parity target.
    inner = np.zeros(grid.shape, bool)
    inner[3:-3, 3:-3] = True
    cells = np.argwhere((grid != 255) & inner)
    flock = (np.arange(n) % k_flocks).astype(np.int32)
    if clustered:
        centres = cells[rng.randint(0, len(cells), size=k_flocks)]
        want = centres[flock] + rng.normal(0, sigma / 4.0, size=(n, 2))
        want = np.clip(np.rint(want), 3, np.array(grid.shape) - 4).astype(int)
        # snap to the nearest passable cell in a small window
        ok = (grid != 255) & inner
        R, Cc = want[:, 0].copy(), want[:, 1].copy()
        for i in range(n):
            if not ok[R[i], Cc[i]]:
                for rad in range(1, 8):
                    r0, r1 = max(R[i] - rad, 0), min(R[i] + rad + 1, grid.shape[0])
                    c0, c1 = max(Cc[i] - rad, 0), min(Cc[i] + rad + 1, grid.shape[1])
                    sub = np.argwhere(ok[r0:r1, c0:c1])
                    if len(sub):
                        R[i], Cc[i] = r0 + sub[0, 0], c0 + sub[0, 1]
                        break
    else:
        idx = rng.randint(0, len(cells), size=n)
        R, Cc = cells[idx, 0], cells[idx, 1]
    pos = synth.cell_centre(w, h, R, Cc) + rng.uniform(-1.5, 1.5, size=(n, 2)).astype(np.float32)
    vel = rng.normal(0.0, 0.45, size=(n, 2)).astype(np.float32)
    state = np.zeros(n, np.uint8)
    u = rng.rand(n)
    state[u < 0.10] = 2          # STATE_ARRIVED (still: static neighbours, no work item)
    state[(u >= 0.10) & (u < 0.13)] = 7   # STATE_TURNING
    state[(u >= 0.13) & (u < 0.16)] = 4   # STATE_WAITING
    vel[state == 2] = 0
    dests = synth.destinations(grid, k_flocks, seed=seed + 1)
    return {
        "pos_xz": pos.astype(np.float32), "vel_xz": vel,
        "radius": rng.choice([1.0, 1.0, 1.5, 2.5], size=n).astype(np.float32),
        "max_speed": rng.choice([20.0, 15.0, 25.0], size=n).astype(np.float32),
        "speed": np.full(n, 20.0, np.float32),
        "flags": np.full(n, 1 << 3, np.uint32),
        "state": state,
        "has_dest_los": (rng.rand(n) < 0.3).astype(np.uint8),
        "flock": flock,
        "flock_target_xz": synth.cell_centre(w, h, dests[:, 0], dests[:, 1]),
        "dest_cells": dests,
    }


def ref_move_for(nav, world, hz=20):
    """Load `world` into the reference's movement module; returns (RefMove, dest_ids)."""
    k = len(world["flock_target_xz"])
    dest_ids = []
    nav_first = True
    for f in range(k):
        members = np.flatnonzero(world["flock"] == f)
        src = world["pos_xz"][members[0]] if len(members) else world["flock_target_xz"][f]
        ok, did = nav.request_path(src, world["flock_target_xz"][f], clear_cache=nav_first)
        nav_first = False
        dest_ids.append(did)
    nav.trace()
    mv = pfref.RefMove(nav, world["pos_xz"], world["vel_xz"], world["radius"], world["max_speed"],
                       world["speed"], world["flags"], world["state"].astype(np.int32), world["flock"],
                       world["has_dest_los"], world["flock_target_xz"], np.array(dest_ids, np.uint32),
                       hz=hz)
    return mv, dest_ids
