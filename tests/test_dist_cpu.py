"""The N>1 path on CPU: two `gloo` processes run the tick's sharding plan and its two exchange
steps (permafrost_engine_amd.dist) exactly as tick.NavTick does on RCCL, with the per-rank compute
replaced by the oracle (test infrastructure) -- after the exchanges every rank must hold the same
tiles / agent results as a single process computing everything."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_slab_partition_covers_everything():
    from permafrost_engine_amd import dist as pdist
    for n in (0, 1, 7, 64, 100_000, 100_001):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                b, e = pdist.slab(n, r, world)
                assert 0 <= b <= e <= n
                seen.extend(range(b, e))
            assert seen == list(range(n))           # disjoint, ordered, complete
    # ceil split of move_submit_cpu_work (movement.c:3759): the first ranks take ceil(n/world)
    assert pdist.slab(10, 0, 4) == (0, 3) and pdist.slab(10, 3, 4) == (9, 10)


def test_request_slices_follow_destinations():
    from permafrost_engine_amd import dist as pdist
    dest_of_req = np.repeat(np.arange(5), [3, 4, 0, 2, 6])      # destination 2 has no requests
    sl = pdist.request_slices(dest_of_req, 5, 2)
    assert sl == [(0, 7), (7, 15)]                              # dests {0,1,2} | {3,4}
    sl = pdist.request_slices(dest_of_req, 5, 5)
    assert sl == [(0, 3), (3, 7), (0, 0), (7, 9), (9, 15)]


def test_auto_tile_exchange_plan():
    """tile_exchange="auto" (tick.NavTick): which destinations' tiles travel, and the request order that
    makes them one contiguous run per rank."""
    from permafrost_engine_amd import dist as pdist
    apr, fpr, world = 10, 3, 2
    flock = np.repeat(np.arange(6), 5)[:20].copy()          # rank 0: flocks 0,0,0,0,0,1,1,1,1,1 | rank 1: 2..3
    flock[:10] = [0, 0, 1, 1, 2, 2, 0, 1, 2, 5]             # rank 0 steps members of flocks 5 (rank 1's) too
    flock[10:] = [3, 3, 4, 4, 5, 5, 3, 4, 1, -1]            # rank 1 steps a member of flock 1 (rank 0's); one unflocked
    tr = pdist.travelling_destinations(flock, apr, fpr, 6)
    assert tr.tolist() == [False, True, False, False, False, True]
    assert not pdist.travelling_destinations(np.repeat([0, 3], 10), apr, fpr, 6).any()      # rank aligned: nothing travels
    # rank 0's requests (destinations 0..2, three chunks each): destination 1's first, the rest in order
    dest_of_req = np.array([0, 0, 0, 1, 1, 1, 2, 2, 2])
    order, n_first = pdist.travel_first(dest_of_req, tr)
    assert n_first == 3 and order.tolist() == [3, 4, 5, 0, 1, 2, 6, 7, 8]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_dests, n_agents, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank),
                          WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        from permafrost_engine_amd import dist as pdist, synth
        from tests import cases
        r, w, _ = pdist.init(backend="gloo")
        assert (r, w) == (rank, world)

        # identical synthetic job on every rank (tick.NavTick.__init__)
        grid = synth.cost_grid(2, 2, seed=5)
        liid = synth.local_islands(grid)
        dests = synth.destinations(grid, n_dests, seed=42)
        cols = synth.whole_map_requests(grid, dests, liid)
        n_req = len(cols["type"])
        oracle = cases.Oracle(grid)
        full_dirs = oracle.fields(cols)                           # single-process answer

        # step 1+2: each rank builds only its request slice, then the tile exchange
        bounds = pdist.request_slices(cols["dest"], n_dests, world)
        b, e = bounds[rank]
        pool = torch.zeros((n_req, 4096), dtype=torch.uint8)
        pool[b:e] = torch.from_numpy(full_dirs[b:e].reshape(e - b, 4096))
        pdist.exchange_rows(pool, bounds, rank, world)
        assert np.array_equal(pool.numpy().reshape(n_req, 64, 64), full_dirs), "tile exchange"

        # tile_exchange="auto": only a run at the head of every rank's slice travels (ragged sub-ranges
        # that do not tile the array): the others' rows stay as they were
        sub = [(bb, bb + (ee - bb) // 2) for bb, ee in bounds]
        part = torch.zeros((n_req, 4096), dtype=torch.uint8)
        part[b:e] = torch.from_numpy(full_dirs[b:e].reshape(e - b, 4096))
        pdist.exchange_rows(part, sub, rank, world)
        want = np.zeros((n_req, 4096), np.uint8)
        want[b:e] = full_dirs[b:e].reshape(e - b, 4096)
        for bb, ee in sub:
            want[bb:ee] = full_dirs[bb:ee].reshape(ee - bb, 4096)
        assert np.array_equal(part.numpy(), want), "partial tile exchange"

        # step 3+4: each rank steps only its agent slab, then the slab exchange
        world_arrays = cases.make_agents(grid, n_agents, n_dests, seed=3, clustered=False)
        vel_full, pos_full = oracle.step(world_arrays)
        ab = [pdist.slab(n_agents, k, world) for k in range(world)]
        a0, a1 = ab[rank]
        vel = torch.zeros((n_agents, 2), dtype=torch.float32)
        pos = torch.zeros((n_agents, 2), dtype=torch.float32)
        vel[a0:a1] = torch.from_numpy(vel_full[a0:a1])
        pos[a0:a1] = torch.from_numpy(pos_full[a0:a1])
        pdist.exchange_rows(vel, ab, rank, world)
        pdist.exchange_rows(pos, ab, rank, world)
        assert np.array_equal(vel.numpy(), vel_full) and np.array_equal(pos.numpy(), pos_full)

        # timing reduction used by bench.py
        t = pdist.max_over_ranks(float(rank + 1), torch.device("cpu"))
        assert t == float(world)
        pdist.barrier()
        dist.destroy_process_group()
        q.put((rank, "ok"))
    except Exception as exc:                                       # surface the failure in the parent
        import traceback
        q.put((rank, "FAIL: %r\n%s" % (exc, traceback.format_exc())))


@pytest.mark.parametrize("n_dests,n_agents", [(4, 600), (3, 501)])   # equal shares, ragged shares
def test_two_rank_gloo_tick_exchange(n_dests, n_agents):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_dests, n_agents, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_region_agents_stay_in_their_columns():
    """tick.NavTick's weak-scaling world: region q's agents are drawn from the passable cells of the
    global columns [q*rcols, (q+1)*rcols); the unrestricted call is unchanged (world == 1)."""
    from permafrost_engine_amd import synth
    g = synth.cost_grid(4, 2, seed=1234)
    mp = synth.map_pos(4, 2)
    for q in range(2):
        a = synth.agents(g, 400, 3, seed=7 + q, cols=(q * 128, (q + 1) * 128))
        col = (mp[0] - a["pos"][:, 0]) / 4.0
        assert col.min() > q * 128 - 0.5 and col.max() < (q + 1) * 128 + 0.5
    a0 = synth.agents(g, 400, 3, seed=7)
    a1 = synth.agents(g, 400, 3, seed=7, cols=(0, 256), rows=(0, 128))
    assert np.array_equal(a0["pos"], a1["pos"]) and np.array_equal(a0["vel"], a1["vel"])
    # a region of a 2-D tiling: rows and columns restricted
    a2 = synth.agents(g, 300, 3, seed=9, cols=(128, 256), rows=(64, 128))
    col = (mp[0] - a2["pos"][:, 0]) / 4.0
    row = (a2["pos"][:, 1] - mp[2]) / 4.0
    assert col.min() > 127.5 and col.max() < 256.5 and row.min() > 63.5 and row.max() < 128.5


def test_region_grid_fits_the_map_limit():
    from permafrost_engine_amd.tick import region_grid
    assert [region_grid(w) for w in (1, 2, 4, 8, 16)] == [(1, 1), (1, 2), (2, 2), (2, 4), (4, 4)]
    for w in range(1, 17):
        r, c = region_grid(w)
        assert r * c >= w and max(r, c) * 16 <= 64
