"""BASELINE.json's full sizes (configs[2]: 1024x1024 map, 64 whole-map flow fields = 16 384 chunk
fields, 100 000 agents) through size-independent properties + a sampled slice against the oracle
restatement (the whole job would take the CPU oracle minutes)."""
import numpy as np
import pytest

from oracle import navoracle
from tests import cases

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4

DR = np.array([0, -1, -1, -1, 0, 0, 1, 1, 1])       # row step of enum flow_dir (nav.h:94-104)
DC = np.array([0, -1, 0, 1, -1, 1, -1, 0, 1])


@pytest.fixture(scope="module")
def job(navlib):
    synth = cases.synth
    W, K, N = 16, 64, 100_000
    grid = synth.cost_grid(W, W, seed=1234)
    liid = synth.local_islands(grid)
    dests = synth.destinations(grid, K, seed=42)
    cols = synth.whole_map_requests(grid, dests, liid)
    reqs = cases.cols_to_reqs(cols, navlib.FIELD_REQ_DTYPE)
    ctx = navlib.NavContext(W, W)
    ctx.upload_plane(0, navlib.PLANE_COST_BASE, synth.to_chunks(grid))
    ctx.upload_plane(0, navlib.PLANE_BLOCKERS, np.zeros((W, W, 64, 64), np.uint16))
    ctx.upload_plane(0, navlib.PLANE_LOCAL_ISLANDS, synth.to_chunks(liid))
    yield dict(W=W, K=K, N=N, grid=grid, liid=liid, dests=dests, cols=cols, reqs=reqs, ctx=ctx)
    ctx.close()


def test_full_size_fields_properties(navlib, job):
    ctx, reqs, grid = job["ctx"], job["reqs"], job["grid"]
    n = len(reqs)
    assert n == 64 * 256
    ctx.set_field_kernel(0)
    dirs, integ = ctx.N_FlowFieldUpdate(reqs, want_integ=True)
    # (1) two independent algorithms (bit-parallel wave BFS vs LDS relaxation) agree everywhere
    ctx.set_field_kernel(1)
    dirs_g, integ_g = ctx.N_FlowFieldUpdate(reqs, want_integ=True)
    ctx.set_field_kernel(0)
    assert np.array_equal(dirs, dirs_g) and np.array_equal(integ, integ_g)
    # (2) reached set == passable cells 4-connected to a seed; impassable cells never reached
    chunks = cases.synth.to_chunks(grid)
    cost = chunks[reqs["chunk_r"], reqs["chunk_c"]]
    fin = np.isfinite(integ)
    assert not fin[cost == 255].any()
    assert fin.reshape(n, -1).any(1).all(), "every planner-style request must have a seed"
    # (3) descent: every reached non-seed cell points at an in-chunk neighbour whose value is the
    #     minimum of its admissible neighbours, i.e. strictly smaller (unit costs: exactly 1 or 2 less)
    r, c = np.meshgrid(np.arange(64), np.arange(64), indexing="ij")
    for blk in range(0, n, 2048):
        d = dirs[blk:blk + 2048].astype(np.int64)
        g = integ[blk:blk + 2048]
        rr, cc = r[None] + DR[d], c[None] + DC[d]
        inside = (rr >= 0) & (rr < 64) & (cc >= 0) & (cc < 64)
        nb = np.take_along_axis(g.reshape(len(d), -1), (np.clip(rr, 0, 63) * 64 + np.clip(cc, 0, 63)).reshape(len(d), -1),
                                axis=1).reshape(d.shape)
        walk = np.isfinite(g) & (g > 0)
        assert inside[walk].all()
        drop = g[walk] - nb[walk]
        diag = (DR[d] != 0) & (DC[d] != 0)
        assert (drop[~diag[walk]] == 1).all() and (drop[diag[walk]] == 2).all()
        # seeds: NONE in the destination chunk, the portal direction elsewhere; unreached: NONE
        seeds = np.isfinite(g) & (g == 0)
        is_tile = (reqs["type"][blk:blk + 2048] == navlib.TARGET_TILE)[:, None, None]
        assert (d[seeds & is_tile] == 0).all() and (d[seeds & ~is_tile] != 0).all()
        assert (d[~np.isfinite(g)] == 0).all()
    # (4) idempotence of the in-place mode: rebuilding over the result changes nothing
    sub = slice(0, 1024)
    r2 = reqs[sub].copy()
    r2["flags"] |= navlib.REQ_INOUT
    again, _ = ctx.N_FlowFieldUpdate(r2, inout=dirs[sub])
    assert np.array_equal(again, dirs[sub])
    # (5) a sampled slice against the oracle restatement (Dijkstra with a binary heap)
    onav = navoracle.OracleNav(chunks, np.zeros_like(chunks, np.uint16), cases.synth.to_chunks(job["liid"]))
    pick = np.random.RandomState(0).choice(n, size=384, replace=False)
    exp_d, exp_g = onav.build_fields(reqs[pick].view(navoracle.FIELD_REQ_DTYPE), want_integ=True)
    assert np.array_equal(dirs[pick], exp_d) and np.array_equal(integ[pick], exp_g)


def test_full_size_agent_step_properties(navlib, job):
    ctx, grid, W, K, N = job["ctx"], job["grid"], job["W"], job["K"], job["N"]
    synth = cases.synth
    # fields of the whole job on the host so the oracle can sample the same pool
    dirs, _ = ctx.N_FlowFieldUpdate(job["reqs"])
    cols = job["cols"]
    slot_tbl = -np.ones((K, W * W), np.int32)
    slot_tbl[cols["dest"], cols["chunk_r"] * W + cols["chunk_c"]] = np.arange(len(cols["dest"]))
    ag = synth.agents(grid, N, K, seed=7)
    offs, members = navlib.flock_csr(ag["flock"], K)
    arrays = {
        "pos_xz": ag["pos"], "vel_xz": ag["vel"], "radius": ag["radius"], "max_speed": ag["max_speed"],
        "speed": ag["speed"], "flags": np.full(N, navlib.ENTITY_FLAG_MOVABLE, np.uint32),
        "state": np.zeros(N, np.uint8), "has_dest_los": np.zeros(N, np.uint8), "flock": ag["flock"],
        "flock_target_xz": synth.cell_centre(W, W, job["dests"][:, 0], job["dests"][:, 1]),
        "flock_offsets": offs, "flock_members": members,
        "flock_field_slot": slot_tbl, "field_pool": dirs.reshape(len(dirs), 4096), "vdes_xz": None,
    }
    out = ctx.agent_step(arrays)
    vel = out["vel_xz"]
    # (1) speed cap: |v| <= max_speed / hz (movement.c:3464), a hair of rounding allowed
    assert (np.linalg.norm(vel.astype(np.float64), axis=1) <= 1.0 + 1e-6).all()
    assert np.isfinite(vel).all()
    # (2) accepted moves land on pathable cells, rejected ones stay put
    moved = (out["status"] & navlib.ST_MOVED) != 0
    assert np.array_equal(out["new_pos_xz"][~moved], ag["pos"][~moved])
    np_ = out["new_pos_xz"][moved]
    C = np.clip(((W * 128.0 - np_[:, 0]) / 4.0).astype(int), 0, W * 64 - 1)
    R = np.clip(((np_[:, 1] + W * 128.0) / 4.0).astype(int), 0, W * 64 - 1)
    assert (grid[R, C] != 255).all()
    assert moved.mean() > 0.5
    # (3) shard invariance: two slab calls == one call (the multi-GPU partition, movement.c:3759)
    w, keep = navlib.make_world(W, W, arrays)
    import ctypes as C_
    halves = np.zeros((N, 2), np.float32)
    so = navlib.StepOut()
    so.vel_xz = halves.ctypes.data                  # both slab calls write into the same array
    for b, e in ((0, N // 2), (N // 2, N)):
        w.work_begin, w.work_end = b, e
        rc = navlib.lib().navhip_agent_step(ctx._h, C_.byref(w), C_.byref(so))
        assert rc == 0
    assert np.array_equal(halves.view(np.uint32), vel.view(np.uint32))
    # (4) a sampled slab against the oracle restatement on the SAME full snapshot
    onav = navoracle.OracleNav(synth.to_chunks(grid), np.zeros((W, W, 64, 64), np.uint16),
                               synth.to_chunks(job["liid"]))
    b, e = 41_000, 42_500
    exp = onav.agent_step(arrays, work=(b, e), nthreads=8)
    d = np.linalg.norm(vel[b:e].astype(np.float64) - exp["vel_xz"][b:e], axis=1)
    rel = d / np.maximum(np.linalg.norm(exp["vel_xz"][b:e].astype(np.float64), axis=1), 1e-3)
    assert (rel <= REL_TOL).all(), rel.max()
    assert np.array_equal(out["status"][b:e], exp["status"][b:e])
    dv = np.abs(out["vdes_xz"][b:e] - exp["vdes_xz"][b:e]).max()
    assert dv <= 1e-6
