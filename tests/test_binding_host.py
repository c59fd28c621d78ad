"""The host side of bindings/permafrost/move_hip.c WITHOUT a device (the reference's movement.c compiled into the
harness, oracle/_ref): the snapshot the WORK_TYPE_HIP arm hands to the library -- built from the engine's khash
tables through one arena, cached bucket positions, cached flock tables and a fork over worker threads -- must be
the tick's tables (struct move_gamestate, movement.c:296; the movestate columns; flock->ents in kh_foreach
order, the order cohesion_force sums in, :1660), tick after tick and after the entity set changes."""
import numpy as np
import pytest

from oracle import pfref

from . import cases

pytestmark = pytest.mark.skipif(not pfref.available(), reason="oracle/_ref (reference sources) not built")


def _check(mv, world, flock_order):
    snap = mv.hip_snapshot()
    moving = ~np.isin(world["state"], (2, 4))                       # (velocities of still units are not loaded)
    assert np.array_equal(snap["pos"].view(np.uint32), world["pos_xz"].view(np.uint32))
    assert np.array_equal(snap["radius"], world["radius"]) and np.array_equal(snap["flags"], world["flags"])
    assert np.array_equal(snap["state"], world["state"].astype(np.uint8))
    assert np.array_equal(snap["max_speed"], world["max_speed"])
    assert np.array_equal(snap["vel"][moving].view(np.uint32), world["vel_xz"][moving].view(np.uint32))
    assert np.array_equal(snap["flock"], world["flock"])
    k = len(world["flock_target_xz"])
    assert len(snap["flock_offsets"]) == k + 1
    for f in range(k):
        got = snap["flock_members"][snap["flock_offsets"][f]:snap["flock_offsets"][f + 1]]
        assert np.array_equal(got, flock_order[f]), f


@pytest.mark.parametrize("threads", [1, 4])
def test_snapshot_tables_of_the_binding(threads):
    grid, nav = cases.ref_nav_for(4, 4, seed=21)
    try:
        for n, seed in ((1500, 3), (1500, 4), (900, 5)):             # same size again, then another entity set
            world = cases.make_agents(grid, n, 4, seed=seed, clustered=False)
            mv, _ = cases.ref_move_for(nav, world)
            order = [mv.flock_order(f) for f in range(4)]
            mv.hip_threads(threads, min_items=64)
            for _ in range(3):                                       # (ticks 2 and 3 run on the cached positions)
                _check(mv, world, order)
            # the dry run of a whole velocity pass: fill, work items, scatter (outputs read as zero)
            mv.hip_dry_run(True)
            vdes = np.zeros((n, 2), np.float32)
            vdes[:, 0] = 1.0
            out = mv.velocity_hip(vdes)
            assert out is not None and not out.any()
            part = mv.velocity_hip(vdes, begin=100, end=700)
            assert part is not None and not part.any()
            mv.hip_dry_run(False)
            _check(mv, world, order)
            pfref.RefMove.unload()
    finally:
        lib = pfref.lib()
        lib.pfref_move_hip_dry_run(0)
        lib.pfref_move_hip_threads(1, 0)
        pfref.RefMove.unload()


def test_velocity_pass_host_side_with_formations_and_arrival():
    """The same dry run with formation members and units committed to arrival slots in the world: every optional
    column of the navhip_world is filled (the largest footprint of the per-tick arena), serial and forked."""
    grid, nav = cases.ref_nav_for(4, 4, seed=21)
    n, k = 1500, 4
    world = cases.make_agents(grid, n, k, seed=31 + n, clustered=False)
    world["state"], form = cases.formation_inputs(world, seed=6)
    sink, aflags = cases.arrival_inputs(world, seed=3)
    mv, _ = cases.ref_move_for(nav, world)
    try:
        mv.set_formation(form["form_ready"], form["cell_pos_xz"], form["form_cohesion_xz"],
                         form["form_align_xz"], form["form_drag_xz"])
        mv.set_arrival(sink, aflags)
        vdes = np.zeros((n, 2), np.float32)
        vdes[:, 1] = 1.0
        mv.hip_dry_run(True)
        for threads in (1, 4):
            mv.hip_threads(threads, min_items=64)
            for _ in range(2):
                out = mv.velocity_hip(vdes)
                assert out is not None and not out.any()
        snap = mv.hip_snapshot()
        assert np.array_equal(snap["state"], world["state"].astype(np.uint8))
    finally:
        lib = pfref.lib()
        lib.pfref_move_hip_dry_run(0)
        lib.pfref_move_hip_threads(1, 0)
        pfref.RefMove.unload()
