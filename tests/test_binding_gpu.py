"""The drop-in, proven: the reference's own planner (n_request_path, nav.c:1774), field cache and
sampler (N_DesiredPointSeekVelocity, nav.c:3468) and its movement tick (move_velocity_work,
movement.c:3395) drive libnavhip.so through the binding a maintainer would add (oracle/ref/nav_hip.c,
move_hip.c, compiled against the reference's headers inside the test harness).  The field cache
contents and the velocities must equal the all-CPU run of the same reference code."""
import numpy as np
import pytest

from oracle import pfref
from tests import cases

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not pfref.available(), reason="oracle/_ref (the reference build) is not present")]


def _agents(grid, n, k, seed):
    rng = np.random.RandomState(seed)
    h, w = grid.shape[0] // 64, grid.shape[1] // 64
    inner = np.zeros(grid.shape, bool)
    inner[3:-3, 3:-3] = True
    cells = np.argwhere((grid != 255) & inner)
    pick = cells[rng.randint(len(cells), size=n)]
    pos = cases.synth.cell_centre(w, h, pick[:, 0], pick[:, 1]) + rng.uniform(-1.5, 1.5, (n, 2)).astype(np.float32)
    dests = cases.synth.destinations(grid, k, seed=seed + 1)
    dxz = cases.synth.cell_centre(w, h, dests[:, 0], dests[:, 1])
    which = rng.randint(k, size=n)
    return pos.astype(np.float32), dxz[which].astype(np.float32), which


def _blockers_around(nav, grid, pos, rng, n_blocked, n_walled):
    """Blockers under some agents (their tile gets local island NONE -> nearest-pathable repair) and
    rings of blockers around others (orphaned islands -> island-to-nearest repair)."""
    for i in rng.choice(len(pos), n_blocked, replace=False):
        nav.blockers_circle(float(pos[i, 0]), float(pos[i, 1]), 3.0, incref=True)
    for i in rng.choice(len(pos), n_walled, replace=False):
        for a in np.linspace(0, 2 * np.pi, 14, endpoint=False):
            nav.blockers_circle(float(pos[i, 0] + 18 * np.cos(a)), float(pos[i, 1] + 18 * np.sin(a)), 4.0, incref=True)
    nav.flush_dirty()


@pytest.mark.parametrize("with_blockers", [False, True])
def test_planner_and_sampler_drive_the_device(with_blockers):
    grid, nav = cases.ref_nav_for(4, 4, seed=21)
    n, k = 400, 5
    pos, dst, which = _agents(grid, n, k, seed=3)
    if with_blockers:
        _blockers_around(nav, grid, pos, np.random.RandomState(9), 25, 12)
    ids = np.array([nav.dest_id(d) for d in dst], np.uint32)
    assert nav.hip_init(), "no MI355X visible"
    try:
        # (1) the reference alone: serial N_DesiredPointSeekVelocity, CPU field builds
        nav.cache_clear()
        pfref.RefNav.hip_mode(False)
        ref_out = nav.desired_velocities(ids, pos, dst)
        ref_cache = nav.cache_dump(ids)
        assert len(ref_cache) > 20 and np.abs(ref_out).max() > 0
        # (2) same serial code, every field build through N_HIP_FlowFieldUpdate & co. on the device
        nav.cache_clear()
        pfref.RefNav.hip_mode(True, 1)
        out = nav.desired_velocities(ids, pos, dst)
        cache = nav.cache_dump(ids)
        st = pfref.RefNav.hip_stats()
        assert st["device_builds"] > 20
        assert cache.keys() == ref_cache.keys()
        for key in ref_cache:
            assert np.array_equal(cache[key], ref_cache[key]), key
        assert np.array_equal(out.view(np.uint32), ref_out.view(np.uint32))
        # (3) the miss-collecting batched form: CPU builders behind the binding (control), then the device
        results = {}
        for backend in (0, 1):
            nav.cache_clear()
            pfref.RefNav.hip_mode(True, backend)
            before = pfref.RefNav.hip_stats()
            results[backend] = (nav.desired_velocities(ids, pos, dst, batched=True), nav.cache_dump(ids),
                                before, pfref.RefNav.hip_stats())
        b_out, b_cache, s0, s1 = results[1]
        c_out, c_cache, _, _ = results[0]
        assert b_cache.keys() == c_cache.keys()
        for key in c_cache:
            assert np.array_equal(b_cache[key], c_cache[key]), key
        assert np.array_equal(b_out.view(np.uint32), c_out.view(np.uint32))
        builds, batches = s1["device_builds"] - s0["device_builds"], s1["batches"] - s0["batches"]
        assert builds > 20 and batches <= 12 and batches < builds / 4, (builds, batches)
        # the batched form against the plain serial run: agents whose requests do not interact get the
        # same direction; a few differ because the serial run finishes agent i (including the planner
        # calls that re-map chunks of its destination) before agent i + 1 looks at the cache, the
        # batched one interleaves them -- the reference's own result depends on the agent order too
        same = (b_out.view(np.uint32) == ref_out.view(np.uint32)).all(1)
        assert same.mean() > 0.93, same.mean()
        if with_blockers:
            li = nav.plane(pfref.PLANE_LOCAL_ISLANDS)
            assert (li == 0xFFFF).sum() > (grid == 255).sum()      # blockers really took tiles away
    finally:
        pfref.RefNav.hip_mode(False)
        pfref.RefNav.hip_shutdown()


def test_movement_tick_drives_the_device():
    """move_velocity_work for every work item through the WORK_TYPE_HIP arm: snapshot tables and
    work items -> navhip_world -> navhip_agent_step_submit / _wait -> s_move_work.out[]."""
    grid, nav = cases.ref_nav_for(4, 4, seed=21, blockers=cases.random_blockers(cases.synth.cost_grid(4, 4, seed=21), 8, 0.02))
    n, k = 1500, 4
    world = cases.make_agents(grid, n, k, seed=31 + n, clustered=False)
    world["state"], form = cases.formation_inputs(world, seed=6)
    sink, aflags = cases.arrival_inputs(world, seed=3)
    mv, dest_ids = cases.ref_move_for(nav, world)
    mv.set_formation(form["form_ready"], form["cell_pos_xz"], form["form_cohesion_xz"],
                     form["form_align_xz"], form["form_drag_xz"])
    mv.set_arrival(sink, aflags)
    rng = np.random.RandomState(1)
    vdes = rng.normal(0, 1, (n, 2)).astype(np.float32)
    vdes /= np.linalg.norm(vdes, axis=1, keepdims=True)
    exp = mv.velocity(vdes)
    assert nav.hip_init()
    try:
        got = mv.velocity_hip(vdes)
        assert got is not None
        moving = ~np.isin(world["state"], (2, 4))
        assert np.array_equal(got[moving].view(np.uint32), exp[moving].view(np.uint32))
        # a slab of the work items (the split of move_submit_cpu_work)
        part = mv.velocity_hip(vdes, begin=300, end=900)
        sel = np.zeros(n, bool); sel[300:900] = True
        assert np.array_equal(part[sel & moving].view(np.uint32), exp[sel & moving].view(np.uint32))
    finally:
        pfref.RefNav.hip_shutdown()
        pfref.RefMove.unload()
