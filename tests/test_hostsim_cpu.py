"""The thread-per-agent bodies of the movement step (csrc/agent_thread.h) compiled for the host
(tests/hostsim) against the reference build: visiting order and caps of the combined neighbour walk,
separation sums, the priority ladder, admissibility, the sequential ClearPath search and the
position accept -- everything in them that is logic rather than a GPU intrinsic.  Runs without a GPU;
the same source is what the device kernels call (the -m gpu tests check those end to end)."""
import numpy as np
import pytest

from oracle import pfref
from tests import cases, hostsim

pytestmark = pytest.mark.skipif(not pfref.available(), reason="oracle/_ref (the reference build) not present")


@pytest.fixture(scope="module")
def navlib():
    from permafrost_engine_amd import navhip
    return navhip


@pytest.mark.parametrize("seed,max_dyn,max_stat,spread", [(1, 2, 2, 9.0), (2, 4, 0, 6.0), (3, 0, 4, 5.0),
                                                         (4, 3, 1, 2.5), (5, 1, 1, 4.0), (6, 8, 8, 9.0),
                                                         (7, 32, 32, 9.5)])
def test_serial_clearpath_search_matches_reference(seed, max_dyn, max_stat, spread):
    nq = 600 if max_dyn < 32 else 80
    ent, des, dyn, nd, stat, ns = cases.cp_problems(seed, nq, max_dyn, max_stat, spread)
    got, found = hostsim.clearpath_light(ent, des, dyn, nd, stat, ns)
    n_checked = 0
    for i in range(nq):
        if not found[i]:
            continue            # the step hands these to the wave path (remove_furthest + retry)
        exp = pfref.clearpath_new_velocity(ent[i], des[i], dyn[i, :nd[i]], stat[i, :ns[i]])
        both_nan = np.isnan(exp) & np.isnan(got[i])
        assert np.array_equal(np.where(both_nan, 0, got[i]).view(np.uint32),
                              np.where(both_nan, 0, exp).view(np.uint32)), (i, nd[i], ns[i], got[i], exp)
        n_checked += 1
    assert n_checked > nq * (0.8 if max_dyn < 32 else 0.3)


DISP_FULL = 7      # agent_thread.h: DISP_DONE, ROW0..3, WAVE, HEAVY, FULL (the irregular gather)

def _world(navlib, clustered, n, k, blk, seed=21, garrison=False, arrival=False):
    grid = cases.synth.cost_grid(4, 4, seed=seed)
    blockers = cases.random_blockers(grid, seed=8, frac=0.02) if blk else None
    grid, nav = cases.ref_nav_for(4, 4, seed=seed, blockers=blockers)
    world = cases.make_agents(grid, n, k, seed=31 + n, clustered=clustered)
    if garrison:
        g = np.random.RandomState(5).rand(n) < 0.02
        world["flags"] = np.where(g, world["flags"] | navlib.ENTITY_FLAG_GARRISONED, world["flags"]).astype(np.uint32)
    return grid, nav, world


@pytest.mark.parametrize("clustered,n,k,blk,garrison", [(False, 1500, 4, False, False), (True, 1200, 3, False, False),
                                                        (True, 1500, 2, True, False), (False, 1500, 4, False, True)])
def test_thread_step_matches_reference(navlib, clustered, n, k, blk, garrison):
    grid, nav, world = _world(navlib, clustered, n, k, blk, garrison=garrison)
    mv, dest_ids = cases.ref_move_for(nav, world)
    exp_vel = mv.velocity(None)
    vdes = mv.vdes()
    order = [mv.flock_order(f) for f in range(k)]
    a = cases.step_arrays(world, vdes, order)
    moving = ~np.isin(world["state"], (2, 4))
    coh = np.zeros((n, 2), np.float32)
    for uid in np.flatnonzero(np.isin(world["state"], (0, 5, 6))):
        coh[uid] = mv.forces(int(uid), vdes[uid])[1]
    out = hostsim.agent_step(navlib, 4, 4, nav.plane(0), nav.plane(1), a, coh)
    disp = out["disp"]
    computed = moving & (disp < DISP_FULL)
    assert computed.sum() > 0.6 * moving.sum(), np.bincount(disp[moving])
    # ClearPath neighbour lists (counts) against find_neighbours
    for uid in np.flatnonzero(moving & (disp != DISP_FULL))[:300]:
        dyn, stat = mv.neighbours(int(uid))
        assert (len(dyn), len(stat)) == tuple(out["counts"][uid]), uid
    # preferred velocity of the point-seek agents, then the final velocities, bit for bit
    ps = np.flatnonzero(np.isin(world["state"], (0, 5, 6)) & (disp != DISP_FULL))
    for uid in ps[:80]:
        ev = mv.vpref(int(uid), vdes[uid])
        assert np.array_equal(out["vpref_xz"][uid].view(np.uint32), ev.view(np.uint32)), ("vpref", uid)
    bad = np.flatnonzero(computed & ~(out["vel_xz"].view(np.uint32) == exp_vel.view(np.uint32)).all(1))
    assert len(bad) == 0, (bad[:10], disp[bad[:10]], out["vel_xz"][bad[:3]], exp_vel[bad[:3]])
    assert np.all(out["vel_xz"][~moving] == 0)
    if garrison:
        assert (disp[moving] == DISP_FULL).sum() > 0          # garrisoned neighbours -> the irregular list
    # position accept
    for uid in np.flatnonzero(computed)[:200]:
        v = exp_vel[uid]
        npos = world["pos_xz"][uid] + v
        on_blocked = nav.position_blocked(world["pos_xz"][uid])
        acc = (np.linalg.norm(v) > 0) and nav.position_pathable(npos) and (on_blocked or not nav.position_blocked(npos))
        if world["flags"][uid] & navlib.ENTITY_FLAG_GARRISONED:
            acc = False
        assert bool(out["status"][uid] & 1) == bool(acc), uid
    pfref.RefMove.unload()


def test_thread_step_with_arrival_state_matches_reference(navlib):
    """G_Arrival_SeekTarget / G_Arrival_NeighbourSettling (the reference's own arrival.c in the harness):
    committed units seek their slot, settling neighbours are static obstacles."""
    n, k = 1500, 4
    grid, nav, world = _world(navlib, False, n, k, False)
    sink, aflags = cases.arrival_inputs(world, seed=3)
    mv, dest_ids = cases.ref_move_for(nav, world)
    base_vel = mv.velocity(None)
    vdes = mv.vdes()
    mv.set_arrival(sink, aflags)
    exp_vel = mv.velocity(vdes)
    moving = ~np.isin(world["state"], (2, 4))
    assert (exp_vel[moving] != base_vel[moving]).any(1).sum() > 50      # the inputs matter
    a = cases.step_arrays(world, vdes, [mv.flock_order(f) for f in range(k)])
    a["arrival_sink_xz"], a["arrival_flags"] = sink, aflags
    coh = np.zeros((n, 2), np.float32)
    for uid in np.flatnonzero(np.isin(world["state"], (0, 5, 6))):
        coh[uid] = mv.forces(int(uid), vdes[uid])[1]
    out = hostsim.agent_step(navlib, 4, 4, nav.plane(0), nav.plane(1), a, coh)
    computed = moving & (out["disp"] < DISP_FULL)
    assert computed.sum() > 0.6 * moving.sum()
    for uid in np.flatnonzero(moving)[:300]:
        dyn, stat = mv.neighbours(int(uid))
        assert (len(dyn), len(stat)) == tuple(out["counts"][uid]), uid
    bad = np.flatnonzero(computed & ~(out["vel_xz"].view(np.uint32) == exp_vel.view(np.uint32)).all(1))
    assert len(bad) == 0, (bad[:10], out["vel_xz"][bad[:3]], exp_vel[bad[:3]])
    pfref.RefMove.unload()
