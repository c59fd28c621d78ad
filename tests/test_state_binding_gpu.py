"""The state half of the reference's movement tick through bindings/permafrost/move_hip.c (move_hip_state_work ->
navhip_state_pass + the settle calls; csrc/state_kernels.hip) against the reference's own entity_compute_update: the arrival
arm and the arms added at the end of round 4 -- arrival zones, formation flags, wait counters, turning, enter-range.  Sorted behind the older GPU tests on purpose: this
part of the library had not run on a GPU when it was written."""
import numpy as np
import pytest

from oracle import pfref
from tests import cases

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not pfref.available(), reason="oracle/_ref (the reference build) is not present")]


def test_state_updates_through_the_binding():
    """fork_join_state_updates through bindings/permafrost/move_hip.c: the snapshot tables + the two
    destination-only nav queries of arrived() per flock -> ONE navhip_state_update for the slab
    (move_hip_state_work), then move_hip_update_work = move_update_work with the switch's outcome taken from
    the device for every unit it decided.  Next state and blocker flag of every unit == the reference's own
    entity_compute_update."""
    grid, nav, world, new_vel, vdes = cases.state_world()
    n = len(world["state"])
    mv, _ = cases.ref_move_for(nav, world)
    try:
        ref_state, ref_flags = mv.state_update(new_vel, vdes)
        assert nav.hip_init(), "no MI355X visible"
        got = mv.state_update_hip(new_vel, vdes)
        assert got is not None
        st, fl, dv = got
        assert np.array_equal(st, ref_state) and np.array_equal(fl, ref_flags)
        decided = (dv & 0x80) == 0
        moving = np.isin(world["state"], (0, 1))
        # the device decided most units, every branch among them
        assert decided.sum() > 0.7 * n          # (not: WAITING / TURNING units, the fifth on the other nav layer)
        assert (decided & moving & (st == 2)).sum() > 100 and (decided & moving & (st == 4)).sum() > 10
        assert (decided & moving & (fl == 0)).sum() > 100
        stats = mv.hip_state_stats()
        assert stats[0] == decided.sum() and stats[1] == n - decided.sum() and stats[2] == 1
        # a slab of the work items
        part = mv.state_update_hip(new_vel, vdes, begin=500, end=1700)
        assert np.array_equal(part[0][500:1700], ref_state[500:1700]) and np.array_equal(part[1][500:1700], ref_flags[500:1700])
        # the host loops forked over worker threads: the same answers
        mv.hip_threads(4, min_items=64)
        try:
            for _ in range(2):
                st2, fl2, dv2 = mv.state_update_hip(new_vel, vdes)
                assert np.array_equal(st2, ref_state) and np.array_equal(fl2, ref_flags) and np.array_equal(dv2, dv)
        finally:
            mv.hip_threads(1)
    finally:
        pfref.RefNav.hip_shutdown()
        pfref.RefMove.unload()


def test_flock_queries_follow_blocker_changes():
    """the binding keeps arrived()'s two per-flock queries between state passes -- M_NavClosestPathable
    and the destination's closest island tiles.  Both READ THE BLOCKERS (n_tile_blocked, nav.c:235; n_closest_island_tiles with
    ignore_blockers = false, :4725), and blockers change exactly where a flock arrives.  Between two passes a unit settles on
    a flock's destination (N_BlockersIncref around the target): the answers of that flock change, and the second pass --
    through the binding, the kept answers keyed on the blocker generation -- equals the reference's own
    entity_compute_update again.  (Keyed on target and nav-data epoch only, the device decided on stale answers.)"""
    grid, nav, world, new_vel, vdes = cases.state_world()
    # where N_ClosestPathable sends each such flock once its destination is blocked: forty units of the flock wait there,
    # on the move, 8-12 units from the target -- not arrived while the target is free, arrived once it is taken
    flocks = (0, 2, 3, 5)
    rng = np.random.RandomState(3)
    placed = []
    for f in flocks:
        x, z = (float(v) for v in world["flock_target_xz"][f])
        free = nav.closest_pathable((x, z))
        assert free is not None and abs(free[0] - x) < 1e-3 and abs(free[1] - z) < 1e-3      # (the target itself, while it is free)
        nav.blockers_circle(x, z, 7.0, incref=True)
        spot = nav.closest_pathable((x, z))
        nav.blockers_circle(x, z, 7.0, incref=False)
        assert spot is not None and np.hypot(spot[0] - x, spot[1] - z) > 4.0
        who = np.flatnonzero(world["flock"] == f)[:40]
        world["pos_xz"][who] = (np.array(spot, np.float32) + rng.uniform(-0.6, 0.6, (len(who), 2))).astype(np.float32)
        world["state"][who] = 0
        world["radius"][who] = 1.0
        world["flags"][who] = np.uint32(1 << 3)
        new_vel[who] = 0
        vdes[who] = (1.0, 0.0)
        placed.append(who)
        # (no ARRIVED unit within reach of them: "a flock mate that touches us has arrived" must not decide for them)
        near = np.hypot(world["pos_xz"][:, 0] - spot[0], world["pos_xz"][:, 1] - spot[1]) < 25.0
        world["state"][near & (world["state"] == 2)] = 0
    placed = np.concatenate(placed)
    mv, _ = cases.ref_move_for(nav, world)
    try:
        assert nav.hip_init(), "no MI355X visible"
        pfref.RefNav.hip_mode(True, 1)
        ref_state, ref_flags = mv.state_update(new_vel, vdes)
        st, fl, _ = mv.state_update_hip(new_vel, vdes)
        assert np.array_equal(st, ref_state) and np.array_equal(fl, ref_flags)
        st, fl, _ = mv.state_update_hip(new_vel, vdes)           # (nothing changed: the kept answers serve)
        assert np.array_equal(st, ref_state) and np.array_equal(fl, ref_flags)
        # units settle on the destinations: blockers under every target
        for f in flocks:
            x, z = (float(v) for v in world["flock_target_xz"][f])
            nav.blockers_circle(x, z, 7.0, incref=True)
        nav.flush_dirty()
        assert pfref.RefNav.hip_blockers_flush()
        ref_state2, ref_flags2 = mv.state_update(new_vel, vdes)
        differ = (ref_state2 != ref_state) | (ref_flags2 != ref_flags)
        assert differ[placed].sum() >= 100, "the blockers under the targets must change decisions, or the test shows nothing"
        st2, fl2, _ = mv.state_update_hip(new_vel, vdes)
        assert np.array_equal(st2, ref_state2) and np.array_equal(fl2, ref_flags2)
        # ... and when the units leave again
        for f in flocks:
            x, z = (float(v) for v in world["flock_target_xz"][f])
            nav.blockers_circle(x, z, 7.0, incref=False)
        nav.flush_dirty()
        assert pfref.RefNav.hip_blockers_flush()
        st3, fl3, _ = mv.state_update_hip(new_vel, vdes)
        assert np.array_equal(st3, ref_state) and np.array_equal(fl3, ref_flags)
    finally:
        pfref.RefNav.hip_mode(False)
        pfref.RefNav.hip_shutdown()
        pfref.RefMove.unload()


def test_arrival_settle_through_the_binding():
    """The same pass with ACTIVE arrival zones on two flocks (struct arrival_state with a footprint, slots and fill
    ranks): their units take the G_Arrival_ShouldSettle arm of entity_compute_update (movement.c:2443-2451).  The
    binding gathers them, counts their settled neighbours and runs the rule on the device (navhip_settled_count,
    navhip_arrival_settle), and the heading gate of every unit in one navhip_heading_gate: next state and blocker
    flag of every unit == the reference's own, and what the device's rule leaves in each unit's arrival state ==
    what the reference's leaves there."""
    grid, nav, world, new_vel, vdes = cases.state_world()
    n = len(world["state"])
    rng = np.random.RandomState(31)
    zones = {}
    for f, (fill, active_row, num_rows) in ((0, (0.8, 1, 3)), (3, (0.95, 2, 3))):
        t = world["flock_target_xz"][f]
        cell = (int((t[1] + 4 * 128.0) // 4), int((4 * 128.0 - t[0]) // 4))
        zones[f] = cases.arrival_zone_at(grid, cell, 8, rng, fill, active_row, num_rows)
        # the flock's units: round the zone, a share of them on slots
        m = np.flatnonzero(world["flock"] == f)
        world["pos_xz"][m] = (zones[f]["centre_xz"] + rng.normal(0, 22.0, (len(m), 2))).astype(np.float32)
        on = m[rng.rand(len(m)) < 0.3]
        world["pos_xz"][on] = zones[f]["slots_xz"][rng.randint(len(zones[f]["slots_xz"]), size=len(on))] \
            + rng.normal(0, 0.7, (len(on), 2)).astype(np.float32)
    world["pos_xz"] = np.clip(world["pos_xz"], -4 * 128.0 + 14, 4 * 128.0 - 14).astype(np.float32)
    in_zone = np.isin(world["flock"], list(zones)) & (world["radius"] < 5.0)
    sink = world["pos_xz"] + rng.normal(0, 12.0, (n, 2)).astype(np.float32)
    for f, z in zones.items():
        m = np.flatnonzero(world["flock"] == f)
        sink[m] = z["slots_xz"][rng.randint(len(z["slots_xz"]), size=len(m))]
    units = {"substate": rng.randint(0, 4, n).astype(np.uint8), "sink_valid": (rng.rand(n) < 0.7).astype(np.uint8),
             "sink_xz": sink.astype(np.float32), "order_pos_xz": (world["pos_xz"] + rng.normal(0, 3.5, (n, 2))).astype(np.float32),
             "progress_anchor_xz": (world["pos_xz"] + rng.normal(0, 1.4, (n, 2))).astype(np.float32),
             "progress_anchored": (rng.rand(n) < 0.7).astype(np.uint8), "stuck": rng.randint(0, 14, n).astype(np.int32)}
    mv, _ = cases.ref_move_for(nav, world)
    try:
        for f, z in zones.items():
            assert mv.set_arrival_zone(f, z) == len(z["tiles"])
        mv.set_arrival_units(units)
        ref_state, ref_flags = mv.state_update(new_vel, vdes)
        ref_after = mv.get_arrival_units()
        changed = (ref_after["substate"] != units["substate"]) | (ref_after["stuck"] != units["stuck"])
        assert changed[in_zone].sum() > 100 and not changed[~np.isin(world["flock"], list(zones))].any()
        mv.set_arrival_units(units)
        assert nav.hip_init(), "no MI355X visible"
        st, fl, dv = mv.state_update_hip(new_vel, vdes)
        assert np.array_equal(st, ref_state) and np.array_equal(fl, ref_flags)
        after = mv.get_arrival_units()
        for k in after:
            assert np.array_equal(after[k], ref_after[k]), k
        decided, settled, differ, gate_host = mv.hip_settle_stats()
        moving = np.isin(world["state"], (0, 1)) & ((world["flags"] & (1 << 18)) == 0)
        assert decided == (in_zone & moving).sum() > 300
        assert settled > 30 and differ == 0 and gate_host == 0
        # the units of the zones are the device's now
        assert ((dv[in_zone & moving] & 0x80) == 0).all()
        assert (in_zone & moving & (st == 2)).sum() >= settled and (in_zone & moving & (st == 0)).sum() > 50
    finally:
        pfref.RefNav.hip_shutdown()
        pfref.RefMove.unload()


def test_formation_and_wait_arms_through_the_binding():
    """The state pass with formation flags and wait counters in play (movement.c:2423-2437, :2630-2668): members on
    the move are no longer the host's -- the arrival arm answers for them and navhip_state_update_aux overrides where
    the flags decide --, ARRIVING_TO_CELL, WAITING, TURNING and ENTER_ENTITY_RANGE units are decided on the device, UPDATE_SET_MOVING, UPDATE_SET_DEST and
    UPDATE_SET_TARGET_DIR reach the patch.  Every unit's next state and flags == entity_compute_update's, the wait
    counters the device returns == the ones the reference leaves in movestate."""
    grid, nav, world, new_vel, vdes = cases.state_world()
    n = len(world["state"])
    rng = np.random.RandomState(12)
    world["state"] = world["state"].copy()
    u = rng.rand(n)
    world["state"][u < 0.10] = 8
    world["state"][(u >= 0.10) & (u < 0.18)] = 1
    world["state"][(u >= 0.18) & (u < 0.26)] = 4
    fstate = ((rng.rand(n) < 0.45) * 1 | (rng.rand(n) < 0.7) * 2 | (rng.rand(n) < 0.7) * 4 | (rng.rand(n) < 0.5) * 8
              | (rng.rand(n) < 0.5) * 16).astype(np.uint8)
    ticks = rng.choice([1, 1, 2, 3, 40], n).astype(np.int32)
    prev = rng.choice([0, 1, 3, 5], n).astype(np.uint8)
    # STATE_TURNING units (:2606-2628): the rotation against movestate.target_dir, half of them within the 5 degrees
    world["state"][(u >= 0.26) & (u < 0.34)] = 7
    ang = rng.uniform(-np.pi, np.pi, n)
    off = np.where(rng.rand(n) < 0.5, rng.uniform(-4.5, 4.5, n), rng.uniform(6, 180, n) * rng.choice([-1, 1], n))
    target_dir = pfref.RefMove.dir_quat(np.stack([np.cos(ang), np.sin(ang)], 1))
    ent_rot = pfref.RefMove.dir_quat(np.stack([np.cos(ang + np.deg2rad(off)), np.sin(ang + np.deg2rad(off))], 1))
    # STATE_ENTER_ENTITY_RANGE units (:2569-2604): a target among the neighbours (or none), a range, where the target stood
    world["state"][(u >= 0.34) & (u < 0.46)] = 6
    er = np.flatnonzero(world["state"] == 6)
    tgt = np.full(n, -1, np.int32)
    for i in er:
        if rng.rand() < 0.9:
            d = np.linalg.norm(world["pos_xz"] - world["pos_xz"][i], axis=1)
            d[i] = np.inf
            tgt[i] = np.argsort(d)[rng.randint(1, 60)]
    t_range = rng.choice([0.0, 5.0, 20.0, 60.0], n).astype(np.float32)
    t_prev = (world["pos_xz"][np.maximum(tgt, 0)] + rng.normal(0, 4.0, (n, 2))).astype(np.float32)
    mv, _ = cases.ref_move_for(nav, world)
    try:
        mv.set_state_aux(fstate, ticks, prev)
        mv.set_turning(ent_rot, target_dir)
        mv.set_range_targets(tgt, t_range, t_prev)
        ref_state, ref_flags = mv.state_update(new_vel, vdes)
        ref_ticks = mv.get_wait_ticks()
        assert (ref_flags & 4).sum() > 50 and (ref_flags & 8).sum() > 10 and (ref_flags & 16).sum() > 20
        assert ((world["state"] == 6) & (ref_state == 4)).sum() > 30 and ((world["state"] == 6) & (ref_state == 2)).sum() > 10
        assert ((world["state"] == 7) & (ref_state == 2)).sum() > 50 and ((world["state"] == 7) & (ref_state == 7)).sum() > 50
        mv.set_state_aux(fstate, ticks, prev)
        assert nav.hip_init(), "no MI355X visible"
        st, fl, dv = mv.state_update_hip(new_vel, vdes)
        assert np.array_equal(st, ref_state) and np.array_equal(fl, ref_flags)
        assert np.array_equal(mv.get_wait_ticks(), ref_ticks) and mv.hip_wait_differ() == 0
        decided = (dv & 0x80) == 0
        garr = (world["flags"] & (1 << 18)) != 0
        assert decided[np.isin(world["state"], (4, 6, 7, 8)) | garr].all()
        assert decided.sum() > 0.9 * n          # (not: the units on another nav layer than their flock's tables)
        assert (decided & (fl == 4)).sum() > 50 and (decided & (fl == 9) & (st == 7)).sum() > 10
        assert (decided & np.isin(world["state"], (0, 1)) & (st == 8)).sum() > 50
    finally:
        pfref.RefNav.hip_shutdown()
        pfref.RefMove.unload()


@pytest.mark.parametrize("hz", [10, 5, 1])
def test_state_pass_at_lower_movement_rates(hz):
    """Below 20 Hz entity_compute_update tests the first INTERPOLATED position of an accepted move (movement.c:2368-2377).
    The binding hands movestate.next_pos / .step to the gate kernel, which makes that position: the device decides the
    units it decides at 20 Hz (round 4 left every MOVING / WAITING unit to the host at these rates), and every unit's next
    state and flags equal the reference's."""
    grid, nav, world, new_vel, vdes = cases.state_world()
    n = len(world["state"])
    rng = np.random.RandomState(70 + hz)
    from_xz = (world["pos_xz"] + rng.normal(0, 0.3, (n, 2))).astype(np.float32)
    step = rng.choice([1.0 / (20 // hz), 1.0, 0.9995, 0.5], n).astype(np.float32)
    mv, _ = cases.ref_move_for(nav, world, hz=hz)
    try:
        mv.set_interp(from_xz, step)
        ref_state, ref_flags = mv.state_update(new_vel, vdes)
        assert nav.hip_init(), "no MI355X visible"
        mv.set_interp(from_xz, step)
        st, fl, dv = mv.state_update_hip(new_vel, vdes)
        assert np.array_equal(st, ref_state) and np.array_equal(fl, ref_flags)
        host = (dv & 0x80) != 0
        garr = (world["flags"] & (1 << 18)) != 0
        big = world["radius"] >= 5.0                     # (another nav layer than their flock's tables: the host's, as at 20 Hz)
        assert not host[np.isin(world["state"], (0, 1, 4)) & ~garr & ~big].any() and not host[garr].any()
        assert not host[np.isin(world["state"], (2, 3))].any()
        assert host.mean() < 0.15                     # (the fifth on the other nav layer that fall through + TURNING)
    finally:
        pfref.RefNav.hip_shutdown()
        pfref.RefMove.unload()


def test_surround_units_through_the_binding():
    """STATE_SURROUND_ENTITY through move_hip_state_work: the binding asks the two unit-query questions per surround unit
    (M_NavObjAdjacentFrom; M_NavClosestReachableAdjacentPosFrom from pos + new velocity and from pos), the device runs the
    switch (navhip_state_aux_in.surround_*), move_hip_update_work turns NAVHIP_SU_SURROUND_DEST into UPDATE_SET_DEST with
    the device's position.  Every unit's next state, flags and -- for the surround units -- next_dest and the
    surround_*_prev the reference's own switch stores, against entity_compute_update."""
    grid, nav, world, new_vel, vdes = cases.state_world(seed=9)
    n = len(world["state"])
    rng = np.random.RandomState(33)
    world["state"] = world["state"].copy()
    small = world["radius"] < 5.0
    su = np.flatnonzero((rng.rand(n) < 0.3) & small & (world["flags"] & (1 << 18) == 0))
    world["state"][su] = 5
    world["vel_xz"] = world["vel_xz"].copy()
    world["vel_xz"][su[rng.rand(len(su)) < 0.4]] = 0
    tgt = np.full(n, -1, np.int32)
    for i in su:
        r = rng.rand()
        if r < 0.08:
            continue
        d = np.linalg.norm(world["pos_xz"] - world["pos_xz"][i], axis=1)
        d[i] = np.inf
        order = np.argsort(d)
        tgt[i] = order[0] if r < 0.25 else order[rng.randint(3, 200)]
    t_prev = world["pos_xz"][np.maximum(tgt, 0)].copy()
    moved = rng.rand(n) < 0.5
    t_prev[moved] += rng.normal(0, 3.0, (moved.sum(), 2)).astype(np.float32)
    n_prev = (world["pos_xz"] + rng.normal(0, 6.0, (n, 2))).astype(np.float32)
    mv, _ = cases.ref_move_for(nav, world)
    try:
        mv.set_surround(tgt, t_prev, n_prev)
        ref_state, ref_flags = mv.state_update(new_vel, vdes)
        ref_tprev, ref_nprev, ref_dest = mv.get_surround()
        assert nav.hip_init(), "no MI355X visible"
        mv.set_surround(tgt, t_prev, n_prev)                # (the reference's switch has stored into movestate: start over)
        st, fl, dv = mv.state_update_hip(new_vel, vdes)
        got_tprev, got_nprev, got_dest = mv.get_surround()
        assert np.array_equal(st, ref_state) and np.array_equal(fl, ref_flags)
        is_su = world["state"] == 5
        assert not (dv[is_su] & 0x80).any() and is_su.sum() > 500          # every surround unit decided on the device
        sd = is_su & ((fl & 16) != 0)
        assert sd.sum() > 100 and np.array_equal(got_dest[sd], ref_dest[sd])
        assert np.array_equal(got_nprev, ref_nprev) and np.array_equal(got_tprev, ref_tprev)
        assert mv.hip_surround_differ() == 0
        assert (is_su & (st == 2)).sum() > 30 and (is_su & (st == 5) & (fl == 0)).sum() > 10
    finally:
        pfref.RefNav.hip_shutdown()
        pfref.RefMove.unload()


def test_state_pass_on_the_resident_snapshot_of_the_velocity_pass():
    """The tick's two fork-joins back to back (navigation_tick_task, movement.c:4263-4280) through the binding: the
    velocity pass leaves its snapshot and its results on the device, the state pass that follows runs on them
    (navhip_state_pass_resident: only movestate.next_rot and the flag / counter inputs travel, through page-locked
    memory).  Every unit's next state and flags == the host-buffer pass's == entity_compute_update on the velocities the
    device produced; a state pass without a velocity pass in front of it falls back to the host-buffer form."""
    import os
    grid, nav, world, new_vel, vdes = cases.state_world()
    n = len(world["state"])
    # (the host emulator steps ~30 agents a second: a slab of the work items there, all of them on the GPU)
    m = 500 if os.path.basename(os.environ.get("NAVHIP_LIB", "")) == "_navhip_emu.so" else n
    rng = np.random.RandomState(3)
    fstate = ((rng.rand(n) < 0.3) * 1 | (rng.rand(n) < 0.7) * 2 | (rng.rand(n) < 0.5) * 4 | (rng.rand(n) < 0.5) * 8).astype(np.uint8)
    ticks = rng.choice([1, 2, 40], n).astype(np.int32)
    prev = rng.choice([0, 1], n).astype(np.uint8)
    mv, _ = cases.ref_move_for(nav, world)
    try:
        assert nav.hip_init(), "no MI355X visible"
        assert mv.bench_hip(vdes, end=m) is not None                       # the velocity pass on the device
        vel, vd = mv.get_out()
        mv.set_state_aux(fstate, ticks, prev)
        ref_state, ref_flags = mv.state_update(vel, vd, end=m)             # the reference on THOSE velocities
        mv.set_state_aux(fstate, ticks, prev)
        host = mv.state_update_hip(vel, vd, end=m)                         # the host-buffer pass
        assert np.array_equal(host[0][:m], ref_state[:m]) and np.array_equal(host[1][:m], ref_flags[:m])
        before = mv.hip_resident_passes()
        mv.hip_resident_state_pass(True)
        try:
            for _ in range(2):
                assert mv.bench_hip(vdes, end=m) is not None               # velocity pass ...
                mv.set_state_aux(fstate, ticks, prev)
                st, fl, dv = mv.state_update_hip(vel, vd, end=m)           # ... state pass on what it left on the device
                assert np.array_equal(st[:m], ref_state[:m]) and np.array_equal(fl[:m], ref_flags[:m]) and np.array_equal(dv[:m], host[2][:m])
            assert mv.hip_resident_passes() == before + 2
            mv.set_state_aux(fstate, ticks, prev)
            st, fl, dv = mv.state_update_hip(vel, vd, end=m)               # no velocity pass in front: the host-buffer form
            assert np.array_equal(st[:m], ref_state[:m]) and np.array_equal(fl[:m], ref_flags[:m]) and mv.hip_resident_passes() == before + 2
        finally:
            mv.hip_resident_state_pass(False)
        assert ((dv[:m] & 0x80) == 0).mean() > 0.7
    finally:
        pfref.RefNav.hip_shutdown()
        pfref.RefMove.unload()


def test_resident_pass_of_a_world_with_every_arm_of_the_switch():
    """The whole state half as the benchmark runs it (tests/tools/bench_state_pass.py, a smaller world): the velocity pass,
    then move_hip_state_work on its resident snapshot -- the arrays of the TURNING / ENTER_ENTITY_RANGE / SURROUND_ENTITY
    arms as SPARSE rows (navhip_state_aux_in.sparse_units), the units of two active arrival zones counted and settled in
    ONE call on the same snapshot (navhip_arrival_settle_resident, nsettled NULL) -- and the host-buffer pass with a row
    per entity: every unit's next state and flags == entity_compute_update given the device's velocities, both ways;
    what the device's settle rule leaves in the units' arrival state and the surround arm's position == the reference's."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    emu = os.path.basename(os.environ.get("NAVHIP_LIB", "")) == "_navhip_emu.so"
    size = ["--agents", "3000", "--chunks", "4", "--flocks", "8", "--end", "400"] if emu else ["--agents", "20000", "--chunks", "8", "--flocks", "16"]
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "tools", "bench_state_pass.py"), "--reps", "1", "--threads", "4",
                        "--arms-scale", "5"] + size, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert "error" not in d, d
    assert d["host_buffers"]["identical"] and d["resident"]["identical"], d
    assert d["resident_passes"] == 2 and d["decided_on_device"] == 1.0
    assert all(d["slab_mix"][k] >= 5 for k in ("1", "5", "6", "7", "8")), d["slab_mix"]      # every arm among the work items
    decided, settled, differ, gate_host = d["settle_stats"]
    assert decided > 50 and settled > 5 and differ == 0 and d["surround_differ"] == 0
