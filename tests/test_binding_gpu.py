"""The drop-in, proven: the reference's own planner (n_request_path, nav.c:1774), field cache and
sampler (N_DesiredPointSeekVelocity, nav.c:3468), its asynchronous field batch (N_PrepareAsyncWork /
N_RequestAsync*Field / N_AwaitAsyncFields, nav.c:3767-3969), its LOS chains, its blocker updates and its
movement tick (move_velocity_work, movement.c:3395) drive libnavhip.so through the binding a maintainer
would add (bindings/permafrost/nav_hip.c, field_hip.c, move_hip.c, compiled against the reference's
headers inside the test harness).  The field cache contents and the velocities must equal the all-CPU
run of the same reference code."""
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
        ref_los = nav.los_dump(ids)
        assert len(ref_cache) > 20 and np.abs(ref_out).max() > 0
        assert len(ref_los) > 10 and any(v.any() for v in ref_los.values())
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
        # ... and every LOS field of the chains (N_LOSFieldCreate at nav.c:1843,2035 -> N_HIP_LOSFieldCreate
        # -> navhip_build_los), in the reference's LOS cache (N_FC_PutLOSField)
        los = nav.los_dump(ids)
        ls = pfref.RefNav.hip_seam_stats()
        assert ls["los_device_fields"] >= len(ref_los) and los.keys() == ref_los.keys()
        for key in ref_los:
            assert np.array_equal(los[key], ref_los[key]), key
        # (3) the miss-collecting batched form: CPU builders behind the binding (control), then the device
        results = {}
        for backend in (0, 1):
            nav.cache_clear()
            pfref.RefNav.hip_mode(True, backend)
            before = pfref.RefNav.hip_stats()
            l0 = pfref.RefNav.hip_seam_stats()
            results[backend] = (nav.desired_velocities(ids, pos, dst, batched=True), nav.cache_dump(ids),
                                before, pfref.RefNav.hip_stats(), nav.los_dump(ids), l0, pfref.RefNav.hip_seam_stats())
        b_out, b_cache, s0, s1, b_los, l0, l1 = results[1]
        c_out, c_cache, _, _, c_los, _, _ = results[0]
        # the LOS fields of the batched rounds: recorded, built one chain LEVEL per device call
        assert b_los.keys() == c_los.keys()
        for key in c_los:
            assert np.array_equal(b_los[key], c_los[key]), key
        los_n, los_b = l1["los_device_fields"] - l0["los_device_fields"], l1["los_batches"] - l0["los_batches"]
        assert los_n >= len(c_los) and los_b < los_n / 2, (los_n, los_b)
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
        # the binding's host loops forked over worker threads (the engine's task fan-out): the same arrays
        mv.hip_threads(4, min_items=64)
        try:
            for _ in range(3):
                forked = mv.velocity_hip(vdes)
                assert np.array_equal(forked.view(np.uint32), got.view(np.uint32))
        finally:
            mv.hip_threads(1)
    finally:
        pfref.RefNav.hip_shutdown()
        pfref.RefMove.unload()


def _game(grid, n, seed, n_factions=3):
    rng = np.random.RandomState(seed)
    h, w = grid.shape[0] // 64, grid.shape[1] // 64
    inner = np.zeros(grid.shape, bool)
    inner[6:-6, 6:-6] = True
    cells = np.argwhere((grid != 255) & inner)
    pick = cells[rng.choice(len(cells), size=n, replace=False)]
    pos = cases.synth.cell_centre(w, h, pick[:, 0], pick[:, 1]) + rng.uniform(-1.5, 1.5, (n, 2)).astype(np.float32)
    radius = rng.choice([1.0, 1.0, 2.5, 4.0], size=n).astype(np.float32)
    faction = rng.randint(0, n_factions, n).astype(np.int32)
    flags = np.full(n, (1 << 3) | (1 << 4), np.uint32)             # MOVABLE | COMBATABLE
    flags[rng.rand(n) < 0.1] &= ~np.uint32(1 << 4)                  # some are not combatable: never a target
    return pos.astype(np.float32), radius, faction, flags


@pytest.mark.parametrize("W,H,layers", [(4, 4, 0x1), (3, 2, 0xf)])
def test_async_field_batch_drives_the_device(W, H, layers):
    """compute_async_fields of a tick (movement.c:4149-4164) with enemy-seek, surround and group-arrival
    jobs on several layers: the request functions of nav.c run unchanged, the jobs create no fiber, ONE
    navhip_build_region_fields call builds them -- the reference's own frontier extraction
    (field_enemies / entity / zone_initial_frontier) on the host -- and N_AwaitAsyncFields puts them into
    the cache.  Every field equals the all-CPU batch of the same calls."""
    grid, nav = cases.ref_nav_for(W, H, seed=33, layer_mask=layers)
    n = 300
    pos, radius, faction, flags = _game(grid, n, seed=4)
    rng = np.random.RandomState(8)
    # standing units block their tiles: the seeds of an enemy field are mostly NOT passable
    for i in rng.choice(n, 120, replace=False):
        nav.blockers_circle(float(pos[i, 0]), float(pos[i, 1]), float(radius[i]), int(faction[i]), incref=True)
    nav.flush_dirty()
    nav.game_load(pos, radius, faction, flags)
    for f, m in ((0, 0b110), (1, 0b001), (2, 0b001)):
        pfref.set_enemy_factions(f, m)
    nl = bin(layers).count("1")
    reqs = []
    for i in rng.choice(n, 60, replace=False):                       # STATE_SEEK_ENEMIES units
        reqs.append((pfref.ASYNC_ENEMY_SEEK, int(rng.randint(nl)), int(faction[i]), pos[i, 0], pos[i, 1], 0, 0))
    for i in rng.choice(n, 40, replace=False):                       # STATE_SURROUND_ENTITY units
        tgt = int(rng.randint(n))
        reqs.append((pfref.ASYNC_SURROUND, int(rng.randint(nl)), int(faction[i]), pos[i, 0], pos[i, 1], tgt, 0))
    for i in rng.choice(n, 12, replace=False):                       # flock arrival zones
        reqs.append((pfref.ASYNC_GROUP_ARRIVAL, int(rng.randint(nl)), 0, pos[i, 0], pos[i, 1], 0, int(rng.randint(2, 14))))
    reqs = np.array(reqs, dtype=pfref.ASYNC_REQ_DTYPE)
    assert nav.hip_init(), "no MI355X visible"
    try:
        nav.cache_clear()
        pfref.RefNav.hip_mode(False)
        ref = nav.async_batch(reqs)
        kinds = {(k >> 56) & 0xf for k in ref}
        assert kinds == {2, 4, 5} and len(ref) > 60, (kinds, len(ref))
        assert sum(v.any() for v in ref.values()) > 0.6 * len(ref)
        nav.cache_clear()
        pfref.RefNav.hip_mode(True, 1)
        s0 = pfref.RefNav.hip_seam_stats()
        got = nav.async_batch(reqs)
        s1 = pfref.RefNav.hip_seam_stats()
        assert got.keys() == ref.keys()
        bad = [hex(k) for k in ref if not np.array_equal(got[k], ref[k])]
        assert not bad, (len(bad), bad[:5])
        assert s1["async_device_jobs"] - s0["async_device_jobs"] == len(ref)
        assert s1["async_batches"] - s0["async_batches"] == 1 and s1["async_cpu_jobs"] == s0["async_cpu_jobs"]
        # a second tick: everything is cached, nothing is requested (nav.c:3798)
        assert nav.async_batch(reqs) == {}
    finally:
        pfref.RefNav.hip_mode(False)
        pfref.RefNav.hip_shutdown()
        pfref.RefNav.game_unload()
        for f in range(3):
            pfref.set_enemy_factions(f, 0)


def test_blocker_tick_through_the_binding(navlib):
    """N_BlockersIncref / N_BlockersDecref (nav.c:4663,4685) with the binding's one extra statement: the
    host planes are updated as ever, the recorded circles update the DEVICE planes in one
    navhip_blockers_circles call -- no plane is re-uploaded.  Afterwards the device's blockers, factions
    and local-island planes equal the host's on all eight ground / water layers, the chunks the device
    flags changed are a subset of the reference's dirty set (nav.c:1033-1046), and the planner's fields,
    built on the device from ITS planes, equal the all-CPU run."""
    import ctypes as C
    grid, nav = cases.ref_nav_for(4, 4, seed=21, layer_mask=0xff)
    n, k = 300, 4
    pos, dst, which = _agents(grid, n, k, seed=5)
    ids = np.array([nav.dest_id(d) for d in dst], np.uint32)
    assert nav.hip_init(), "no MI355X visible"
    L = navlib.lib()
    ctx = C.c_void_p(pfref.RefNav.hip_ctx())

    def dev_plane(layer, plane, dt, shape):
        out = np.zeros(shape, dt)
        assert L.navhip_download_plane(ctx, layer, plane, out.ctypes.data_as(C.c_void_p), out.nbytes) == 0
        return out

    try:
        pfref.RefNav.hip_mode(True, 1)
        rng = np.random.RandomState(12)
        cells = cases.synth.passable_cells(grid)
        circles = []
        for _ in range(400):
            c = cells[rng.randint(len(cells))]
            xz = cases.synth.cell_centre(4, 4, c[0], c[1])
            circles.append((float(xz[0]), float(xz[1]), float(np.float32(rng.uniform(1.0, 9.0))), int(rng.randint(0, 4))))
        for tick in range(3):
            if tick == 0:
                batch = [(c, True) for c in circles]
            else:                                           # some leave, some arrive
                gone = [circles[i] for i in rng.choice(len(circles), 60, replace=False)]
                circles = [c for c in circles if c not in gone]
                batch = [(c, False) for c in gone]
            for (x, z, r, f), inc in batch:
                nav.blockers_circle(x, z, r, f, incref=inc)
            dirty = [nav.dirty_chunks(l).astype(bool) for l in range(8)]
            nav.flush_dirty()                               # N_Update's relabel of the dirty local islands
            s0 = pfref.RefNav.hip_seam_stats()
            assert pfref.RefNav.hip_blockers_flush()
            s1 = pfref.RefNav.hip_seam_stats()
            assert s1["blocker_circles"] - s0["blocker_circles"] == len(batch) and s1["blocker_batches"] - s0["blocker_batches"] == 1
            for l in range(8):
                assert np.array_equal(dev_plane(l, navlib.PLANE_BLOCKERS, np.uint16, (4, 4, 64, 64)), nav.plane(pfref.PLANE_BLOCKERS, l)), (tick, l)
                assert np.array_equal(dev_plane(l, navlib.PLANE_FACTIONS, np.uint8, (4, 4, 15, 64, 64)), nav.plane(pfref.PLANE_FACTIONS, l)), (tick, l)
                assert np.array_equal(dev_plane(l, navlib.PLANE_LOCAL_ISLANDS, np.uint16, (4, 4, 64, 64)), nav.plane(pfref.PLANE_LOCAL_ISLANDS, l)), (tick, l)
                ch = np.zeros(16, np.uint8)
                assert L.navhip_changed_chunks(ctx, l, ch.ctypes.data_as(C.c_void_p), 1) == 0
                ch = ch.reshape(4, 4).astype(bool)
                assert not (ch & ~dirty[l]).any(), (tick, l)
                if l == 0:
                    assert ch.any()
        # the planner on the updated world: device fields from the device's own planes == all-CPU
        nav.cache_clear()
        pfref.RefNav.hip_mode(False)
        ref_out = nav.desired_velocities(ids, pos, dst)
        ref_cache = nav.cache_dump(ids)
        nav.cache_clear()
        pfref.RefNav.hip_mode(True, 1)
        out = nav.desired_velocities(ids, pos, dst, batched=False)
        cache = nav.cache_dump(ids)
        assert cache.keys() == ref_cache.keys() and len(cache) > 20
        for key in ref_cache:
            assert np.array_equal(cache[key], ref_cache[key]), key
        assert np.array_equal(out.view(np.uint32), ref_out.view(np.uint32))
    finally:
        pfref.RefNav.hip_mode(False)
        pfref.RefNav.hip_shutdown()


def test_field_cache_device_image_and_device_sampling():
    """The pool binding (N_FC_PutFlowField / N_FC_PutDestFFMapping / N_FC_ClearAll -> navhip_pool_*) and the
    WORK_TYPE_HIP arm sampling on the device: tick 1 starts with an empty cache -- every agent comes back
    flagged, the host's sampler runs the planner (its builds go through the binding and land in the
    pool), and steps those agents itself; tick 2 finds the fields resident and the mappings mirrored --
    the device samples them, no N_DesiredPointSeekVelocity call per agent.  Both ticks equal the all-CPU
    movement tick (host sampling + move_velocity_work) bit for bit."""
    grid, nav = cases.ref_nav_for(4, 4, seed=21)
    n, k = 1200, 5
    world = cases.make_agents(grid, n, k, seed=77, clustered=False)
    world["state"][:] = 0                                     # STATE_MOVING: the point-seek arm
    mv, dest_ids = cases.ref_move_for(nav, world)
    nav.cache_clear()
    pfref.RefNav.hip_mode(False)
    exp = mv.velocity(None)                                   # host sampling (plans on miss) + CPU step
    # the same tick again on the warm cache: a few agents change (the planner calls of later agents
    # re-mapped chunks of their destination during the first pass), then the cache is settled
    exp2 = mv.velocity(None)
    assert np.array_equal(mv.velocity(None).view(np.uint32), exp2.view(np.uint32)) and np.abs(exp).max() > 0
    assert nav.hip_init()
    try:
        assert pfref.RefNav.hip_pool(2048 + 256, 16)
        pfref.RefNav.hip_mode(True, 1)
        pfref.RefNav.hip_device_sampling(True)
        nav.cache_clear()                                     # N_HIP_FC_ClearAll: host cache and pool
        s0 = pfref.RefNav.hip_pool_stats()
        got1 = mv.velocity_hip(None)
        s1 = pfref.RefNav.hip_pool_stats()
        assert got1 is not None
        assert np.array_equal(got1.view(np.uint32), exp.view(np.uint32))
        # every agent needed the host the first time; what the host built is resident now
        assert s1["host_fallbacks"] - s0["host_fallbacks"] > 0.9 * n
        assert s1["puts"] + s1["built_resident"] > 20
        got2 = mv.velocity_hip(None)
        s2 = pfref.RefNav.hip_pool_stats()
        assert np.array_equal(got2.view(np.uint32), exp2.view(np.uint32))
        sampled, fell = s2["device_sampled"] - s1["device_sampled"], s2["host_fallbacks"] - s1["host_fallbacks"]
        assert sampled > 0.95 * n and fell < 0.05 * n, (sampled, fell)
        assert s2["maps"] > 20
    finally:
        pfref.RefNav.hip_device_sampling(False)
        pfref.RefNav.hip_mode(False)
        pfref.RefNav.hip_shutdown()
        pfref.RefMove.unload()
