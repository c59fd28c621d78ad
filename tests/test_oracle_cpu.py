"""Pinning the oracle.  The reference holds no golden vectors for this path, so
the C restatement (oracle/navoracle.c) is pinned against OUTPUTS OF THE REFERENCE ITSELF:
  * live, against oracle/_ref/libpfref.so (the reference's own nav/field/clearpath/movement
    translation units compiled in place) -- these tests need /root/reference or a prebuilt _ref;
  * against the committed fixtures tests/golden/*.npz that tests/tools/make_golden.py produced from
    _ref -- these run anywhere.
Integer work is compared bit for bit; float velocities are compared bit for bit too (same
compiler, same expression order), far inside BASELINE.json's 1e-4 relative bound."""
import os

import numpy as np
import pytest

from oracle import navoracle, pfref
from tests import cases

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
needs_ref = pytest.mark.skipif(not (pfref.available() or os.path.isdir("/root/reference")),
                               reason="oracle/_ref not built and /root/reference absent")


def _o_reqs(ref_reqs):
    out = np.zeros(len(ref_reqs), navoracle.FIELD_REQ_DTYPE)
    out["faction_id"] = 0xF
    for name in ("layer", "type", "faction_id", "chunk_r", "chunk_c", "tile_r", "tile_c",
                 "port_r0", "port_c0", "port_r1", "port_c1", "next_r0", "next_c0", "next_r1",
                 "next_c1", "next_chunk_r", "next_chunk_c", "port_iid", "next_iid"):
        out[name] = ref_reqs[name]
    out["flags"] = np.where(ref_reqs["inout"] != 0, 1, 0)
    return out


# ---------------------------------------------------------------------------------------------
# live against the reference
# ---------------------------------------------------------------------------------------------
@needs_ref
@pytest.mark.parametrize("seed,w,blk", [(11, 2, False), (1234, 4, False), (77, 3, True)])
def test_fields_restatement_matches_reference(seed, w, blk):
    grid = cases.synth.cost_grid(w, w, seed=seed)
    blockers = cases.random_blockers(grid, seed=3) if blk else None
    grid, nav = cases.ref_nav_for(w, w, seed=seed, blockers=blockers)
    reqs_t = cases.tile_requests(grid, 24, seed=5)
    reqs_p, before, _after = cases.planner_requests(nav, grid, pairs=12, seed=9)
    reqs = np.concatenate([reqs_t, reqs_p])
    before = np.concatenate([np.zeros((len(reqs_t), 64, 64), np.uint8), before])
    n_plan = len(reqs)
    reqs, before = cases.with_inplace(reqs, before, seed=seed)
    exp_dirs, exp_integ = cases.ref_fields(nav, reqs, before)
    onav = cases.oracle_nav_from_ref(nav)
    dirs, integ = onav.build_fields(_o_reqs(reqs), inout=before, want_integ=True)
    assert np.array_equal(dirs, exp_dirs)
    assert np.array_equal(integ, exp_integ)
    # the planner's own after-images are what N_FlowFieldUpdate left in the cache
    assert np.array_equal(dirs[len(reqs_t):n_plan], _after)
    assert (reqs["type"] == 0).sum() > 8 and (reqs["inout"] != 0).sum() >= 8


@needs_ref
def test_spatial_query_restatement_matches_reference():
    rng = np.random.RandomState(4)
    pos = np.concatenate([rng.uniform(-510, 510, size=(2500, 2)),
                          rng.normal([100, -200], 6.0, size=(700, 2)),
                          rng.normal([-300, 250], 14.0, size=(700, 2)),
                          [[-512.0, -512.0], [512.0, 512.0], [511.99, -3.0]]]).astype(np.float32)
    q = np.concatenate([pos[::9], [[100, -200], [-300, 250], [-512, 512], [0, 0]]]).astype(np.float32)
    bounds = (-512.0, 512.0, -512.0, 512.0)
    for r, cap in ((30.0, 128), (10.0, 512), (10.0, 7), (1400.0, 256)):
        ec, ei = pfref.spatial_query(bounds, pos, q, r, cap)
        gc, gi = navoracle.spatial_query(4, 4, pos, q, r, cap)
        assert np.array_equal(ec, gc), (r, cap)
        for k in range(len(q)):
            assert np.array_equal(ei[k, :ec[k]], gi[k, :gc[k]]), (r, cap, k)


@needs_ref
@pytest.mark.parametrize("seed,max_dyn,max_stat,spread", [(1, 6, 3, 9.0), (2, 32, 32, 9.5),
                                                         (3, 12, 0, 5.0), (4, 0, 12, 5.0), (5, 3, 3, 2.5)])
def test_clearpath_restatement_matches_reference(seed, max_dyn, max_stat, spread):
    nq = 300 if max_dyn < 32 else 40
    ent, des, dyn, nd, stat, ns = cases.cp_problems(seed, nq, max_dyn, max_stat, spread)
    exp = np.zeros((nq, 2), np.float32)
    for i in range(nq):
        exp[i] = pfref.clearpath_new_velocity(ent[i], des[i], dyn[i, :nd[i]], stat[i, :ns[i]])
    got = navoracle.clearpath(ent, des, dyn, nd, stat, ns)
    assert np.array_equal(got.view(np.uint32), exp.view(np.uint32))      # bit for bit, NaNs included


@needs_ref
@pytest.mark.parametrize("clustered,n,k,blk", [(False, 900, 4, False), (True, 800, 3, False),
                                               (True, 900, 2, True)])
def test_velocity_step_restatement_matches_reference(clustered, n, k, blk):
    grid = cases.synth.cost_grid(4, 4, seed=21)
    blockers = cases.random_blockers(grid, seed=8, frac=0.02) if blk else None
    grid, nav = cases.ref_nav_for(4, 4, seed=21, blockers=blockers)
    world = cases.make_agents(grid, n, k, seed=31 + n, clustered=clustered)
    mv, dest_ids = cases.ref_move_for(nav, world)
    exp_vel = mv.velocity(None)
    vdes = mv.vdes()
    onav = cases.oracle_nav_from_ref(nav)
    arrays = cases.step_arrays(world, vdes, [mv.flock_order(f) for f in range(k)])
    out = onav.agent_step(arrays)
    moving = ~np.isin(world["state"], (2, 4))
    assert np.array_equal(out["vel_xz"][moving].view(np.uint32), exp_vel[moving].view(np.uint32))
    assert np.all(out["vel_xz"][~moving] == 0)
    # individual steering terms and the preferred velocity
    ps = np.flatnonzero(np.isin(world["state"], (0, 5, 6)))
    for uid in ps[:40]:
        ea, ec, es = mv.forces(int(uid), vdes[uid])
        ga, gc, gs = onav.forces(arrays, int(uid), vdes[uid])
        assert np.array_equal(ea, ga) and np.array_equal(ec, gc) and np.array_equal(es, gs), uid
        assert np.array_equal(mv.vpref(int(uid), vdes[uid]), out["vpref_xz"][uid]), uid
    # position accept test vs N_PositionPathable / N_PositionBlocked
    for uid in np.flatnonzero(moving)[:150]:
        v = exp_vel[uid]
        npos = world["pos_xz"][uid] + v
        on_blocked = nav.position_blocked(world["pos_xz"][uid])
        acc = (np.linalg.norm(v) > 0) and nav.position_pathable(npos) and \
            (on_blocked or not nav.position_blocked(npos))
        assert bool(out["status"][uid] & 1) == bool(acc), uid
    # threads split the slab without changing anything
    out4 = onav.agent_step(arrays, nthreads=4)
    assert np.array_equal(out4["vel_xz"].view(np.uint32), out["vel_xz"].view(np.uint32))
    pfref.RefMove.unload()


@needs_ref
def test_flow_sampling_restatement_matches_reference():
    grid, nav = cases.ref_nav_for(4, 4, seed=21)
    world = cases.make_agents(grid, 600, 3, seed=77, clustered=False)
    mv, dest_ids = cases.ref_move_for(nav, world)
    mv.velocity(None)
    exp_vel = mv.velocity(None)
    vdes = mv.vdes()
    slots, pool = cases.cached_field_table(nav, dest_ids, 4, 4)
    arrays = cases.step_arrays(world, None, [mv.flock_order(f) for f in range(3)])
    arrays["flock_field_slot"], arrays["field_pool"] = slots, pool
    out = cases.oracle_nav_from_ref(nav).agent_step(arrays)
    ps = np.isin(world["state"], (0, 5, 6))
    clean = ps & ((out["status"] & 0x06) == 0)
    assert clean.sum() > 250
    assert np.array_equal(out["vdes_xz"][clean], vdes[clean])
    assert np.array_equal(out["vel_xz"][clean], exp_vel[clean])
    pfref.RefMove.unload()


# ---------------------------------------------------------------------------------------------
# against the committed fixtures (generated from the reference by tests/tools/make_golden.py)
# ---------------------------------------------------------------------------------------------
def _gold(name):
    path = os.path.join(GOLD, name)
    assert os.path.exists(path), "missing fixture %s (run tests/tools/make_golden.py)" % name
    return np.load(path)


@pytest.mark.parametrize("name", ["fields_3x3", "fields_5x2"])
def test_golden_fields(name):
    g = _gold(name + ".npz")
    onav = navoracle.OracleNav(g["cost"], g["blockers"], g["local_islands"])
    dirs, integ = onav.build_fields(g["reqs"].view(navoracle.FIELD_REQ_DTYPE).reshape(-1),
                                    inout=g["before"], want_integ=True)
    assert np.array_equal(dirs, g["dirs"])
    assert np.array_equal(integ, g["integ"])


@pytest.mark.parametrize("name", ["agents_4x4", "agents_5x2"])
def test_golden_spatial_and_clearpath(name):
    g = _gold(name + ".npz")
    h, w = g["cost"].shape[:2]
    for key in ("r30", "r10", "wide"):
        c, ids = navoracle.spatial_query(w, h, g["pos_xz"], g["sq_query"], float(g["sq_" + key + "_range"]),
                                         int(g["sq_" + key + "_cap"]))
        ec, ei = g["sq_" + key + "_counts"], g["sq_" + key + "_ids"]
        assert np.array_equal(c, ec)
        for k in range(len(c)):
            assert np.array_equal(ids[k, :c[k]], ei[k, :ec[k]]), (key, k)
    got = navoracle.clearpath(g["cp_ent"], g["cp_des"], g["cp_dyn"], g["cp_nd"], g["cp_stat"], g["cp_ns"])
    assert np.array_equal(got.view(np.uint32), g["cp_out"].view(np.uint32))


@pytest.mark.parametrize("name", ["agents_4x4", "agents_5x2"])
def test_golden_velocity_step(name):
    g = _gold(name + ".npz")
    onav = navoracle.OracleNav(g["cost"], g["blockers"], g["local_islands"])
    arrays = {k: g[k] for k in ("pos_xz", "vel_xz", "radius", "max_speed", "speed", "flags", "state",
                                "has_dest_los", "flock", "flock_target_xz", "flock_offsets",
                                "flock_members")}
    arrays["vdes_xz"] = g["vdes_xz"]
    out = onav.agent_step(arrays)
    moving = ~np.isin(g["state"], (2, 4))
    assert np.array_equal(out["vel_xz"][moving].view(np.uint32), g["ref_vel"][moving].view(np.uint32))
    # sampling the reference's cached fields on the fly gives the reference's desired directions
    arrays["vdes_xz"] = None
    arrays["flock_field_slot"], arrays["field_pool"] = g["flock_field_slot"], g["field_pool"]
    out2 = onav.agent_step(arrays)
    clean = np.isin(g["state"], (0, 5, 6)) & ((out2["status"] & 0x06) == 0)
    assert clean.sum() > 150
    assert np.array_equal(out2["vdes_xz"][clean], g["ref_vdes_sampled"][clean])


# ---------------------------------------------------------------------------------------------
# dynamic obstacles: N_BlockersIncref / Decref + local-island relabel
# ---------------------------------------------------------------------------------------------
def _random_circles(grid, n, seed, air_frac=0.0, max_radius=9.0):
    rng = np.random.RandomState(seed)
    h, w = grid.shape[0] // 64, grid.shape[1] // 64
    c = np.zeros(n, navoracle.CIRCLE_DTYPE)
    c["x"] = rng.uniform(-w * 128.0 + 0.5, w * 128.0 - 0.5, n)
    c["z"] = rng.uniform(-h * 128.0 + 0.5, h * 128.0 - 0.5, n)
    c["radius"] = rng.choice([0.5, 1.0, 2.0, 3.25, 4.0, 6.0, max_radius], size=n)
    c["faction_id"] = rng.randint(0, 4, n)
    c["flags"] = np.where(rng.rand(n) < air_frac, 1 << 15, 0)
    c["delta"] = 1
    # a few on the map edges / corners, and exact tile-boundary positions
    c["x"][0], c["z"][0] = w * 128.0, -h * 128.0
    c["x"][1], c["z"][1] = -w * 128.0, h * 128.0
    c["x"][2], c["z"][2] = 4.0 * 7, -4.0 * 9
    return c


@needs_ref
def test_blockers_and_local_islands_restatement_matches_reference():
    grid = cases.synth.cost_grid(2, 2, seed=13)
    nav = pfref.RefNav(cases.synth.to_chunks(grid), layer_mask=0xff)        # ground + water layers
    onav = navoracle.OracleNav(cases.synth.to_chunks(grid))
    for layer in range(8):
        onav.set_layer(layer, cost=nav.plane(pfref.PLANE_COST, layer),
                       blockers=np.zeros((2, 2, 64, 64), np.uint16),
                       local_islands=nav.plane(pfref.PLANE_LOCAL_ISLANDS, layer),
                       factions=np.zeros((2, 2, 15, 64, 64), np.uint8))
    circles = _random_circles(grid, 90, seed=2, max_radius=22.0)
    circles["radius"][5], circles["radius"][6] = 100.0, 75.0     # the 1024-tile scratch caps bind
    undo = circles[::3].copy()
    undo["delta"] = -1
    for batch in (circles, undo):
        for c in batch:
            nav.blockers_circle(float(c["x"]), float(c["z"]), float(c["radius"]), int(c["faction_id"]),
                                int(c["flags"]), incref=(c["delta"] > 0))
        onav.blockers_circles(batch)
        for layer in range(8):
            assert np.array_equal(onav.plane(layer, "blockers"), nav.plane(pfref.PLANE_BLOCKERS, layer)), layer
            assert np.array_equal(onav.plane(layer, "factions"), nav.plane(pfref.PLANE_FACTIONS, layer)), layer
    assert onav.plane(3, "blockers").sum() > onav.plane(0, "blockers").sum() > 0     # contours on 7x7
    nav.flush_dirty()                                        # n_update_dirty_local_islands
    for layer in range(8):
        assert np.array_equal(onav.local_islands(layer), nav.plane(pfref.PLANE_LOCAL_ISLANDS, layer)), layer


@needs_ref
def test_formation_arms_restatement_matches_reference():
    grid, nav = cases.ref_nav_for(4, 4, seed=21)
    world = cases.make_agents(grid, 900, 3, seed=5, clustered=True)
    world["state"], form = cases.formation_inputs(world, seed=6)
    mv, dest_ids = cases.ref_move_for(nav, world)
    mv.set_formation(form["form_ready"], form["cell_pos_xz"], form["form_cohesion_xz"],
                     form["form_align_xz"], form["form_drag_xz"])
    rng = np.random.RandomState(1)
    vdes = rng.normal(0, 1, (900, 2)).astype(np.float32)
    vdes /= np.linalg.norm(vdes, axis=1, keepdims=True)
    exp_vel = mv.velocity(vdes)
    arrays = cases.step_arrays(world, vdes, [mv.flock_order(f) for f in range(3)])
    arrays.update(form)
    out = cases.oracle_nav_from_ref(nav).agent_step(arrays)
    moving = ~np.isin(world["state"], (2, 4))
    assert np.isin(world["state"], (1, 8)).sum() > 250
    assert (out["status"] & 0x80).sum() == 0
    assert np.array_equal(out["vel_xz"][moving].view(np.uint32), exp_vel[moving].view(np.uint32))
    # without the formation arrays the same agents are reported unsupported, not guessed
    for k in form:
        arrays[k] = None
    out2 = cases.oracle_nav_from_ref(nav).agent_step(arrays)
    assert ((out2["status"] & 0x80) != 0).sum() == np.isin(world["state"], (1, 8)).sum()
    pfref.RefMove.unload()


# ---------------------------------------------------------------------------------------------
# repair builds: N_FlowFieldUpdateToNearestPathable / N_FlowFieldUpdateIslandToNearest
# ---------------------------------------------------------------------------------------------
def repair_cases(nav, grid, seed, n_each=10):
    """(requests, existing fields, expected fields from the reference) for both repair builds on a
    map with blockers: existing fields are real planner fields of the chunk."""
    rng = np.random.RandomState(seed)
    h, w = grid.shape[0] // 64, grid.shape[1] // 64
    cost = nav.plane(pfref.PLANE_COST)
    blk = nav.plane(pfref.PLANE_BLOCKERS)
    li = nav.plane(pfref.PLANE_LOCAL_ISLANDS)
    base_reqs, before, after = cases.planner_requests(nav, grid, pairs=14, seed=seed)
    reqs, exist, exp = [], [], []
    # (A) start tiles that are impassable or blocked
    bad = np.argwhere((cost == 255) | (blk > 0))
    for k in rng.choice(len(bad), n_each, replace=False):
        cr, cc, r, c = (int(x) for x in bad[k])
        pick = np.flatnonzero((base_reqs["chunk_r"] == cr) & (base_reqs["chunk_c"] == cc))
        existing = after[pick[0]] if len(pick) else rng.randint(0, 9, (64, 64)).astype(np.uint8)
        rq = np.zeros(1, pfref.FIELD_REQ_DTYPE)[0]
        rq["type"], rq["chunk_r"], rq["chunk_c"], rq["tile_r"], rq["tile_c"] = 2, cr, cc, r, c
        rq["faction_id"] = 15
        reqs.append((rq, 0, 0))
        exist.append(existing)
        exp.append(nav.field_nearest_pathable(cr, cc, r, c, existing))
    # (B) move the frontier of real planner fields to every local island of their chunk
    order = rng.permutation(len(base_reqs))
    nb = 0
    for k in order:
        rq = base_reqs[k]
        ids = np.unique(li[rq["chunk_r"], rq["chunk_c"]])
        ids = ids[ids != 0xFFFF]
        for iid in ids[:3]:
            reqs.append((rq.copy(), 8, int(iid)))
            exist.append(after[k])
            exp.append(nav.field_island_to_nearest(rq, int(iid), after[k]))
            nb += 1
        if nb >= 2 * n_each:
            break
    return reqs, np.stack(exist), np.stack(exp)


def repair_reqs_to(dtype, reqs):
    out = np.zeros(len(reqs), dtype)
    for i, (rq, flag, iid) in enumerate(reqs):
        for name in ("layer", "type", "faction_id", "chunk_r", "chunk_c", "tile_r", "tile_c",
                     "port_r0", "port_c0", "port_r1", "port_c1", "next_r0", "next_c0", "next_r1",
                     "next_c1", "next_chunk_r", "next_chunk_c", "port_iid", "next_iid"):
            out[name][i] = rq[name]
        out["flags"][i] = flag
        out["aux_iid"][i] = iid
    return out


@needs_ref
@pytest.mark.parametrize("seed", [3, 8])
def test_repair_builds_restatement_matches_reference(seed):
    grid = cases.synth.cost_grid(3, 3, seed=40 + seed, frac_impassable=0.3)
    blk = cases.random_blockers(grid, seed=seed, frac=0.06)
    grid, nav = cases.ref_nav_for(3, 3, seed=40 + seed, blockers=blk, frac=0.3)
    reqs, exist, exp = repair_cases(nav, grid, seed)
    onav = cases.oracle_nav_from_ref(nav)
    onav.set_layer(0, islands=nav.plane(pfref.PLANE_ISLANDS))
    got, _ = onav.build_fields(repair_reqs_to(navoracle.FIELD_REQ_DTYPE, reqs), inout=exist)
    bad = [i for i in range(len(reqs)) if not np.array_equal(got[i], exp[i])]
    assert not bad, "repair builds differ: %s" % [(i, int(reqs[i][0]["type"]), reqs[i][1]) for i in bad[:6]]
    changed = sum(int(not np.array_equal(exp[i], exist[i])) for i in range(len(reqs)))
    assert changed > len(reqs) // 3            # the repairs really rewrote something


# ---------------------------------------------------------------------------------------------
# LOS fields (order-dependent wavefront: the restatement carries the reference's binary heap)
# ---------------------------------------------------------------------------------------------
@needs_ref
@pytest.mark.parametrize("seed,blk", [(2, False), (7, True)])
def test_los_restatement_matches_reference(seed, blk):
    grid = cases.synth.cost_grid(3, 3, seed=60 + seed, frac_impassable=0.25)
    blockers = cases.random_blockers(grid, seed=seed, frac=0.04) if blk else None
    grid, nav = cases.ref_nav_for(3, 3, seed=60 + seed, blockers=blockers, frac=0.25)
    reqs, prevs, exps = cases.los_chains(nav, grid, n_dests=5, seed=seed)
    got = cases.oracle_nav_from_ref(nav).build_los(cases.los_reqs_to(navoracle.LOS_REQ_DTYPE, reqs), prevs)
    bad = [i for i in range(len(reqs)) if not np.array_equal(got[i], exps[i])]
    assert not bad, "LOS fields differ: %s" % bad[:8]
    assert (exps & 1).mean() > 0.01 and (exps & 2).any()       # something visible, some lines drawn
    assert sum(1 for r in reqs if r["prev_dr"] or r["prev_dc"]) > 10


# ---------------------------------------------------------------------------------------------
# region fields (cell / group arrival, zone)
# ---------------------------------------------------------------------------------------------
@needs_ref
@pytest.mark.parametrize("seed,blk", [(4, False), (9, True)])
def test_region_fields_restatement_matches_reference(seed, blk):
    grid = cases.synth.cost_grid(3, 3, seed=80 + seed, frac_impassable=0.2)
    blockers = cases.random_blockers(grid, seed=seed, frac=0.03) if blk else None
    grid, nav = cases.ref_nav_for(3, 3, seed=80 + seed, blockers=blockers)
    reqs, S, O, inout, exp = cases.region_cases(nav, grid, seed)
    got = cases.oracle_nav_from_ref(nav).build_region_fields(
        cases.region_reqs_to(navoracle.REGION_REQ_DTYPE, reqs), S, O, inout=inout)
    bad = [i for i in range(len(reqs)) if not np.array_equal(got[i], exp[i])]
    assert not bad, "region fields differ: %s" % [(i, reqs[i]["out_mode"]) for i in bad[:8]]
    assert (exp != 0).mean() > 0.2


@needs_ref
@pytest.mark.parametrize("seed", [1, 2])
def test_attacking_path_fields_restatement_matches_reference(seed):
    grid, nav, reqs, enemies, exp_dirs, exp_integ = cases.faction_cases(seed)
    onav = cases.oracle_nav_from_ref(nav)
    onav.set_layer(0, factions=nav.plane(pfref.PLANE_FACTIONS))
    o = _o_reqs(reqs)
    o["faction_id"], o["enemies"] = 0, enemies
    dirs, integ = onav.build_fields(o, want_integ=True)
    assert np.array_equal(dirs, exp_dirs) and np.array_equal(integ, exp_integ)
    # the faction really matters: a neutral request (FACTION_ID_NONE) sees more blocked tiles
    o2 = o.copy()
    o2["faction_id"], o2["enemies"] = 0xF, 0
    d2, _ = onav.build_fields(o2)
    assert not np.array_equal(d2, exp_dirs)


@needs_ref
@pytest.mark.parametrize("w,h", [(5, 2), (2, 5)])
def test_non_square_map_restatement_matches_reference(w, h):
    """Width and height the right way round: fields and the velocity step on 5x2 / 2x5 chunks."""
    grid, nav = cases.ref_nav_for(w, h, seed=300 + w)
    reqs_t = cases.tile_requests(grid, 16, seed=6)
    reqs_p, before, _after = cases.planner_requests(nav, grid, pairs=12, seed=10)
    reqs = np.concatenate([reqs_t, reqs_p])
    before = np.concatenate([np.zeros((len(reqs_t), 64, 64), np.uint8), before])
    exp_dirs, exp_integ = cases.ref_fields(nav, reqs, before)
    onav = cases.oracle_nav_from_ref(nav)
    dirs, integ = onav.build_fields(_o_reqs(reqs), inout=before, want_integ=True)
    assert np.array_equal(dirs, exp_dirs) and np.array_equal(integ, exp_integ)
    world = cases.make_agents(grid, 700, 3, seed=5 + h, clustered=False)
    mv, dest_ids = cases.ref_move_for(nav, world)
    exp_vel = mv.velocity(None)
    vdes = mv.vdes()
    arrays = cases.step_arrays(world, vdes, [mv.flock_order(f) for f in range(3)])
    out = onav.agent_step(arrays)
    moving = ~np.isin(world["state"], (2, 4))
    assert np.array_equal(out["vel_xz"][moving].view(np.uint32), exp_vel[moving].view(np.uint32))
    pfref.RefMove.unload()
