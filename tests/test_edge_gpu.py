"""Edge cases of the hot path on the GPU: empty inputs, tick rates other than 20 Hz, entity flags
(garrisoned / combat-held / air), size layers, agents without a flock, the largest map the
reference's 6-bit chunk ids allow."""
import numpy as np
import pytest

from oracle import navoracle, pfref
from tests import cases

pytestmark = pytest.mark.gpu
REL_TOL = 1e-4


def _vel_err(a, b):
    d = np.linalg.norm(a.astype(np.float64) - b.astype(np.float64), axis=1)
    return d / np.maximum(np.linalg.norm(b.astype(np.float64), axis=1), 1e-3)


def test_empty_inputs_are_fine(navlib):
    ctx = navlib.NavContext(2, 2)
    grid = cases.synth.cost_grid(2, 2, seed=1)
    ctx.upload_plane(0, navlib.PLANE_COST_BASE, cases.synth.to_chunks(grid))
    dirs, integ = ctx.N_FlowFieldUpdate(navlib.make_reqs(0), want_integ=True)
    assert dirs.shape == (0, 64, 64)
    empty = {"pos_xz": np.zeros((0, 2), np.float32), "vel_xz": np.zeros((0, 2), np.float32),
             "radius": np.zeros(0, np.float32), "max_speed": np.zeros(0, np.float32),
             "speed": np.zeros(0, np.float32), "flags": np.zeros(0, np.uint32),
             "state": np.zeros(0, np.uint8), "has_dest_los": np.zeros(0, np.uint8),
             "flock": np.zeros(0, np.int32), "flock_target_xz": np.zeros((0, 2), np.float32),
             "flock_offsets": np.zeros(1, np.int32), "flock_members": np.zeros(0, np.int32),
             "vdes_xz": np.zeros((0, 2), np.float32)}
    out = ctx.agent_step(empty)
    assert out["vel_xz"].shape == (0, 2)
    c, ids = ctx.spatial_query(np.zeros((3, 2), np.float32), np.zeros((0, 2), np.float32), 10.0, 8)
    assert len(c) == 0
    assert ctx.N_LOSFieldCreate(np.zeros(0, navlib.LOS_REQ_DTYPE)).shape == (0, 64, 64)
    ctx.N_BlockersUpdate(np.zeros(0, navlib.CIRCLE_DTYPE))
    # a malformed request is an error, not a crash
    bad = navlib.make_reqs(1)
    bad["type"], bad["chunk_r"] = navlib.TARGET_TILE, 7
    with pytest.raises(navlib.NavHipError):
        ctx.N_FlowFieldUpdate(bad)
    ctx.close()


@pytest.mark.skipif(not pfref.available(), reason="needs the reference build (oracle/_ref)")
@pytest.mark.parametrize("hz", [10, 5, 1])
def test_other_tick_rates_match_reference(navlib, hz):
    grid, nav = cases.ref_nav_for(4, 4, seed=21)
    world = cases.make_agents(grid, 700, 3, seed=hz, clustered=True)
    world["vel_xz"] *= 20.0 / hz                       # velocities are per tick
    mv, _ = cases.ref_move_for(nav, world, hz=hz)
    exp = mv.velocity(None)
    vdes = mv.vdes()
    ctx = navlib.NavContext(4, 4)
    for plane, k in ((navlib.PLANE_COST_BASE, 0), (navlib.PLANE_BLOCKERS, 1), (navlib.PLANE_LOCAL_ISLANDS, 3)):
        ctx.upload_plane(0, plane, nav.plane(k))
    a = cases.step_arrays(world, vdes, [mv.flock_order(f) for f in range(3)])
    w, keep = navlib.make_world(4, 4, a, hz=hz)
    out = ctx.agent_step(a, hz=hz)
    moving = ~np.isin(world["state"], (2, 4))
    err = _vel_err(out["vel_xz"][moving], exp[moving])
    assert (err <= REL_TOL).all(), err.max()
    ctx.close()
    pfref.RefMove.unload()


def test_flags_layers_and_flockless_agents(navlib):
    """Garrisoned entities vanish from neighbour queries (position.c:100), combat-held ones stand
    still (movement.c:3404), air units only see air units and use the air layers, big radii use the
    3x3/5x5 layers (entity.c:554), enemy seekers have no flock (movement.c:3414)."""
    grid = cases.synth.cost_grid(4, 4, seed=21)
    chunks = cases.synth.to_chunks(grid)
    n, k = 900, 2
    world = cases.make_agents(grid, n, k, seed=3, clustered=True)
    rng = np.random.RandomState(5)
    flags = np.full(n, 1 << 3, np.uint32)
    u = rng.rand(n)
    flags[u < 0.08] |= 1 << 18                         # GARRISONED
    flags[(u >= 0.08) & (u < 0.14)] |= 1 << 21         # COMBAT_HELD
    flags[(u >= 0.14) & (u < 0.30)] |= 1 << 15         # AIR
    flags[(u >= 0.30) & (u < 0.34)] &= ~np.uint32(1 << 3)   # not movable: ignored as a neighbour
    world["flags"] = flags
    world["radius"] = rng.choice([1.0, 2.5, 5.5, 11.0], size=n).astype(np.float32)
    seekers = (u >= 0.34) & (u < 0.42) & (world["state"] == 0)
    world["state"][seekers] = 3                        # STATE_SEEK_ENEMIES
    world["flock"][seekers] = -1
    # different obstacles on every layer so that picking the wrong layer shows
    layers = [0, 1, 2, 8, 9, 10]
    blk = {l: cases.random_blockers(grid, seed=100 + l, frac=0.03) for l in layers}
    onav = navoracle.OracleNav(chunks)
    ctx = navlib.NavContext(4, 4)
    for l in layers:
        onav.set_layer(l, cost=chunks, blockers=blk[l])
        ctx.upload_plane(l, navlib.PLANE_COST_BASE, chunks)
        ctx.upload_plane(l, navlib.PLANE_BLOCKERS, blk[l])
    vdes = rng.normal(0, 1, (n, 2)).astype(np.float32)
    vdes /= np.linalg.norm(vdes, axis=1, keepdims=True)
    lists = [np.flatnonzero(world["flock"] == f) for f in range(k)]
    a = cases.step_arrays(world, vdes, lists)
    exp = onav.agent_step(a)
    out = ctx.agent_step(a)
    err = _vel_err(out["vel_xz"], exp["vel_xz"])
    assert (err <= REL_TOL).all(), err.max()
    assert np.array_equal(out["status"], exp["status"])
    held = (flags & (1 << 21)) != 0
    assert np.all(out["vel_xz"][held] == 0)
    assert ((flags & (1 << 15)) != 0).sum() > 50 and seekers.sum() > 20
    ctx.close()


def test_largest_map_and_corner_chunks(navlib):
    """64 x 64 chunks (the reference's 6-bit chunk ids, nav.c:841-848): fields in the four corner
    chunks, bit-exact against the restatement."""
    W = 64
    rng = np.random.RandomState(0)
    grid = np.where(rng.rand(W * 64, W * 64) < 0.18, 255, 1).astype(np.uint8)
    chunks = cases.synth.to_chunks(grid)
    ctx = navlib.NavContext(W, W)
    ctx.upload_plane(0, navlib.PLANE_COST_BASE, chunks)
    reqs = navlib.make_reqs(8)
    reqs["type"] = navlib.TARGET_TILE
    corners = [(0, 0), (0, 63), (63, 0), (63, 63)] * 2
    for i, (cr, cc) in enumerate(corners):
        sub = np.argwhere(chunks[cr, cc] != 255)
        r, c = sub[rng.randint(len(sub))]
        reqs["chunk_r"][i], reqs["chunk_c"][i], reqs["tile_r"][i], reqs["tile_c"][i] = cr, cc, r, c
    dirs, integ = ctx.N_FlowFieldUpdate(reqs, want_integ=True)
    onav = navoracle.OracleNav(chunks)
    ed, ei = onav.build_fields(reqs.view(navoracle.FIELD_REQ_DTYPE), want_integ=True)
    assert np.array_equal(dirs, ed) and np.array_equal(integ, ei)
    with pytest.raises(navlib.NavHipError):
        navlib.NavContext(65, 1)
    ctx.close()


def test_flock_sizes_around_the_cohesion_batch_and_tile_boundaries(navlib):
    """k_cohesion queues a flock's members 256 at a time, evaluates them in batches of 16 split over
    four lanes and carries the tail into the next tile: flocks of 1, 2, 15..17, 31..33, 63..65,
    255..257, 300 and 513 members (some with idle members in between) hit every boundary.  Several
    steps on one context, so that the lane grouping carried between steps is used as well."""
    grid = cases.synth.cost_grid(4, 4, seed=21)
    chunks = cases.synth.to_chunks(grid)
    sizes = [1, 2, 15, 16, 17, 31, 32, 33, 63, 64, 65, 255, 256, 257, 300, 513]
    n, k = sum(sizes), len(sizes)
    world = cases.make_agents(grid, n, k, seed=11, clustered=False)
    rng = np.random.RandomState(17)
    world["flock"] = np.repeat(np.arange(k), sizes).astype(np.int32)[rng.permutation(n)]
    world["state"][:] = 0                               # all point seeking ...
    world["state"][rng.rand(n) < 0.15] = 2              # ... except some arrived (idle) members
    onav = navoracle.OracleNav(chunks)
    ctx = navlib.NavContext(4, 4)
    ctx.upload_plane(0, navlib.PLANE_COST_BASE, chunks)
    vdes = rng.normal(0, 1, (n, 2)).astype(np.float32)
    vdes /= np.linalg.norm(vdes, axis=1, keepdims=True)
    lists = [rng.permutation(np.flatnonzero(world["flock"] == f)) for f in range(k)]
    for step in range(3):
        a = cases.step_arrays(world, vdes, lists)
        exp = onav.agent_step(a)
        out = ctx.agent_step(a)
        assert np.array_equal(out["vpref_xz"].view(np.uint32), exp["vpref_xz"].view(np.uint32)), step
        assert np.array_equal(out["vel_xz"].view(np.uint32), exp["vel_xz"].view(np.uint32)), step
        world["pos_xz"] = exp["new_pos_xz"].copy()
        world["vel_xz"] = exp["vel_xz"].copy()
    assert np.abs(exp["vel_xz"]).max() > 0
    ctx.close()
