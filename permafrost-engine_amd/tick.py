"""The benchmark / demo driver of one navigation tick, everything resident in HBM.

One tick = (1) rebuild every chunk field of this rank's share of the flow fields into the field
pool, (2) [multi-GPU, only the tiles another rank samples] exchange baked tiles, (3) velocity step
+ position accept for this rank's slab of agents, sampling the pool on the device, (4) [multi-GPU]
all-gather the slab results, (5) advance the snapshot (pos <- new_pos, vel <- new velocity).  Nothing is cached
between ticks: the worst case of the reference's tick, where every cached field was invalidated
(N_ApplyDeferredInvalidations, nav.c:2208).

PyTorch supplies device buffers, the stream and torch.distributed; all compute is libnavhip.
"""
import ctypes as C
import os
import time

import numpy as np
import torch

from . import dist as pdist
from . import navhip, synth


class _HostCuda:
    """What NavTick uses of torch.cuda, for the ONE case in which "device" memory is host memory: the library under test
    is the host-emulator build of tests/hostsim (NAVHIP_LIB names _navhip_emu.so; test infrastructure, see
    tests/test_emulated_cpu.py).  Everything there runs synchronously and in order, so streams and events carry no state
    beyond a timestamp."""

    class Stream:
        def __init__(self, device=None, priority=0):
            self.cuda_stream = 0x1000           # (an opaque non-null handle: the emulated runtime never looks inside)

        def wait_event(self, event):
            pass

        def wait_stream(self, stream):
            pass

        def synchronize(self):
            pass

    class ExternalStream(Stream):
        def __init__(self, handle, device=None):
            self.cuda_stream = handle or 0x1000

    class Event:
        def __init__(self, enable_timing=False):
            self.t = 0.0

        def record(self, stream=None):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

        def synchronize(self):
            pass

    class _Props:
        multi_processor_count = 256

    @staticmethod
    def stream(s):
        import contextlib
        return contextlib.nullcontext()

    @staticmethod
    def set_device(dev):
        pass

    @staticmethod
    def synchronize(dev=None):
        pass

    @staticmethod
    def get_device_properties(dev):
        return _HostCuda._Props()


EMULATED = os.path.basename(os.environ.get("NAVHIP_LIB", "")) == "_navhip_emu.so"
tcuda = _HostCuda if EMULATED else torch.cuda


def region_grid(world):
    """(rows, cols) of the region tiling for `world` ranks: cols = the smallest power of two that is
    >= sqrt(world), rows = ceil(world / cols): 1x1, 1x2, 2x2, 2x4, 4x4."""
    cols = 1
    while cols * cols < world:
        cols *= 2
    return -(-world // cols), cols


class NavTick:
    """World layout (weak scaling): `world` REGIONS of chunk_w x chunk_w chunks
    tiling one map (region_grid(world): 1x2, 2x2, 2x4, 4x4 ... regions; a map side is at most 64
    chunks, the reference's 6-bit chunk ids).  Region r -- its
    `fields_per_rank` destinations, the chunk-field requests of those destinations (every chunk of
    the region: the corridor the planner would emit for units and destination inside the region)
    and its `agents_per_rank` agents, flock = destination -- belongs to rank r: the agents a rank
    steps sample the fields that rank built, so the baked tiles only travel when a flock has members
    on another rank (`tile_exchange`).  The map planes and the entity snapshot are replicated; agents
    of neighbouring regions see each other through the all-gathered snapshot."""

    def __init__(self, chunk_w=16, fields_per_rank=64, agents_per_rank=100_000, rank=0, world=1,
                 device=0, hz=20, seed_map=1234, verbose=False, obstacles=0, move_frac=0.01,
                 obstacle_ticks=128, tile_exchange="auto", solo=False, shared_map=False, crowd_cells=0,
                 debug_outputs=False, pipeline_fields=False, exchange="torch", planner_requests=True,
                 straddle=0.0, los=False, flow_velocities=False, share_fields=False, driver="c", serial=None,
                 time_fields=False):
        self.rank, self.world, self.device_index = rank, world, device
        self.dev = torch.device("cpu") if EMULATED else torch.device("cuda", device)
        tcuda.set_device(self.dev)
        self.W = chunk_w                            # region side in chunks
        # shared_map (BASELINE configs[3], strong scaling): ONE chunk_w x chunk_w map for every rank;
        # destinations and agents are split over the ranks, anywhere on the map
        self.shared_map = bool(shared_map)
        self.reg_rows, self.reg_cols = (1, 1) if shared_map else region_grid(world)
        self.Wt, self.H = chunk_w * self.reg_cols, chunk_w * self.reg_rows   # whole map, in chunks
        if max(self.Wt, self.H) > 64:
            raise ValueError("%d regions of %d chunks do not fit a 64x64-chunk map" % (world, chunk_w))
        self.nchunks = self.Wt * self.H
        self.K = fields_per_rank * world            # flow fields (destinations) in the whole job
        self.N = agents_per_rank * world            # agents in the whole job
        self.hz = hz
        t0 = time.time()
        Wt, H = self.Wt, self.H
        rcols = chunk_w * 64                        # cell rows / columns per region

        def region_cells(q):                        # (row0, row1, col0, col1) of region q, in cells
            if self.shared_map:
                return 0, rcols, 0, rcols
            qr, qc = divmod(q, self.reg_cols)
            return qr * rcols, (qr + 1) * rcols, qc * rcols, (qc + 1) * rcols

        # ---- synthetic map, identical on every rank --------------------
        grid = synth.cost_grid(Wt, H, seed=seed_map)
        self.ctx = navhip.NavContext(Wt, H, device=device)
        self.ctx.upload_plane(0, navhip.PLANE_COST_BASE, synth.to_chunks(grid))
        self.ctx.upload_plane(0, navhip.PLANE_BLOCKERS, np.zeros((H, Wt, 64, 64), np.uint16))
        self.n_obstacles = obstacles
        blockers = None
        if obstacles:
            # configs[4]: dynamic obstacles (circles, radius U(2,6) wu, seed 99) dropped through the
            # device N_BlockersIncref path; every tick `move_frac` of them move (decref + incref)
            rng = np.random.RandomState(99)
            cells = synth.passable_cells(grid)
            pos = synth.cell_centre(Wt, H, *cells[rng.randint(len(cells), size=obstacles)].T)
            circ = np.zeros(obstacles, navhip.CIRCLE_DTYPE)
            circ["x"], circ["z"] = pos[:, 0], pos[:, 1]
            circ["radius"] = rng.uniform(2.0, 6.0, obstacles)
            circ["delta"] = 1
            self._circ_host = circ.copy()       # (parity tests replay them through the reference)
            self.ctx.N_BlockersUpdate(circ)
            self.ctx.changed_chunks(0, clear=True)
            blockers = synth.from_chunks(self.ctx.download_plane(0, navhip.PLANE_BLOCKERS))
            nmove = max(1, int(round(obstacles * move_frac)))
            moves = np.zeros((obstacle_ticks, 2 * nmove), navhip.CIRCLE_DTYPE)
            cur = circ.copy()
            for t in range(obstacle_ticks):
                who = rng.choice(obstacles, nmove, replace=False)
                moves[t, :nmove] = cur[who]
                moves[t, :nmove]["delta"] = -1
                npos = synth.cell_centre(Wt, H, *cells[rng.randint(len(cells), size=nmove)].T)
                cur["x"][who], cur["z"][who] = npos[:, 0], npos[:, 1]
                moves[t, nmove:] = cur[who]
                moves[t, nmove:]["delta"] = 1
            self.n_moves = 2 * nmove
            self._moves_host = moves
        self.ctx.relabel_local_islands(0)           # n_update_local_island_field on the device
        liid = synth.from_chunks(self.ctx.download_plane(0, navhip.PLANE_LOCAL_ISLANDS))

        # ---- destinations (cheap, all regions) and agents (replicated snapshot) ----------------
        dests, ag_parts = [], []
        for q in range(world):
            r0, r1, c0, c1 = region_cells(q)
            sub = grid[r0:r1, c0:c1]
            d = synth.destinations(sub, fields_per_rank, seed=42 + q)
            dests.append(d + np.array([r0, c0]))
            a = synth.agents(grid, agents_per_rank, fields_per_rank, seed=7 + q, hz=hz, blockers=blockers,
                             cols=(c0, c1), rows=(r0, r1), crowd_cells=crowd_cells)
            a["flock"] = a["flock"] + q * fields_per_rank
            ag_parts.append(a)
        dests = np.concatenate(dests)
        if straddle > 0 and world > 1:
            # flocks that straddle ranks: in the last `straddle` of every rank's uid slab sit agents of the
            # NEXT region (position and flock; the lower half of its flocks only) -- stepped here, sampling
            # fields another rank builds
            m = int(round(agents_per_rank * (1.0 - straddle)))
            swapped = []
            for q, a in enumerate(ag_parts):
                nxt = ag_parts[(q + 1) % world]
                take = np.zeros(agents_per_rank, bool)
                take[m:] = (nxt["flock"][m:] % fields_per_rank) < max(1, fields_per_rank // 2)
                swapped.append({k: (v if k == "hz" else
                                    np.where(take.reshape((-1,) + (1,) * (np.ndim(v) - 1)), nxt[k], v))
                                for k, v in a.items()})
            ag_parts = swapped
        ag = {k: (np.concatenate([a[k] for a in ag_parts]) if k != "hz" else hz) for k in ag_parts[0]}
        # ---- request stream: region-major, destination-major inside a region -------------------
        # tile_exchange: "auto" = only the fields some other rank samples travel (none when flocks
        # are rank aligned, the default world; `straddle` makes some); "all" = every rank holds every
        # tile, all-gathered every tick
        # (SURVEY section 8(e) worst case: any agent may sample any field)
        # solo (tests): this one process builds every region's fields and steps every agent
        self.solo = bool(solo)
        self.tile_exchange = "all" if ((tile_exchange == "all" or solo) and world > 1) else "none"
        # "auto": destination d (built by rank d // fields_per_rank) travels when some agent of its flock
        # sits in another rank's uid slab
        travels = np.zeros(self.K, bool)
        if world > 1:
            travels = pdist.travelling_destinations(ag["flock"], agents_per_rank, fields_per_rank, self.K)
            if self.tile_exchange == "none" and travels.any():
                self.tile_exchange = "auto"
        regions = range(world) if self.tile_exchange != "none" else [rank]
        req_parts, dest_of_req, self.req_bounds, nreq = [], [], [(0, 0)] * world, 0
        self.xchg_bounds = [(0, 0)] * world        # the rows of a rank the others need
        for q in regions:
            r0, r1, c0, c1 = region_cells(q)
            d_q = dests[q * fields_per_rank:(q + 1) * fields_per_rank] - np.array([r0, c0])
            # the reference planner's own request stream where a fixture holds it (the single-GPU configs:
            # tests/tools/make_requests.py), else the numpy stand-in
            cols = synth.planner_requests(grid[r0:r1, c0:c1], d_q) if planner_requests else None
            self.request_source = "reference planner (n_request_path) fixture" if cols is not None else \
                "numpy stand-in (synth.whole_map_requests)"
            if cols is None:
                cols = synth.whole_map_requests(grid[r0:r1, c0:c1], d_q, liid[r0:r1, c0:c1])
            n_q = len(cols["type"])
            if self.tile_exchange == "auto":
                # the travelling destinations' requests first: one contiguous run per rank to exchange
                order, n_first = pdist.travel_first(np.asarray(cols["dest"]) + q * fields_per_rank, travels)
                cols = {k: np.asarray(v)[order] for k, v in cols.items() if k in synth.REQ_FIELDS or k == "dest"}
                self.xchg_bounds[q] = (nreq, nreq + n_first)
            else:
                self.xchg_bounds[q] = (nreq, nreq + n_q)
            reqs_q = navhip.make_reqs(n_q)
            for k in synth.REQ_FIELDS:
                reqs_q[k] = cols[k]
            reqs_q["chunk_r"] += r0 // 64
            reqs_q["chunk_c"] += c0 // 64
            portal = reqs_q["type"] == navhip.TARGET_PORTAL
            reqs_q["next_chunk_r"][portal] += r0 // 64
            reqs_q["next_chunk_c"][portal] += c0 // 64
            req_parts.append(reqs_q)
            dest_of_req.append(np.asarray(cols["dest"]) + q * fields_per_rank)
            self.req_bounds[q] = (nreq, nreq + n_q)
            nreq += n_q
        reqs = np.concatenate(req_parts)
        dest_of_req = np.concatenate(dest_of_req)
        n_req = len(reqs)
        if obstacles:
            reqs["flags"] = navhip.REQ_LIVE_IIDS | navhip.REQ_IF_CHANGED
        # share_fields: the reference keys its field cache by N_FlowFieldID (field.c:1952) -- chunk + target, NOT
        # the destination -- so destinations whose paths leave a chunk through the same portal share ONE field
        # (N_FC_PutDestFFMapping maps both to it, nav.c:2008-2021), and a tick after a wholesale invalidation
        # rebuilds every DISTINCT field once.  Identical request records are built once and every (dest, chunk)
        # entry of the slot table points at the shared slot.  (Default off: every request is rebuilt.)
        self.n_requests_served = n_req
        mapped_slot = np.arange(n_req)
        if share_fields:
            if world != 1 or obstacles:
                raise ValueError("share_fields: single-process worlds without moving obstacles only")
            uniq, first, inv = np.unique(reqs, return_index=True, return_inverse=True)
            order = np.sort(first)                       # (keep the stream's order: first occurrences)
            rank_of = np.empty(len(first), np.int64)
            rank_of[np.argsort(first)] = np.arange(len(first))
            mapped_slot = rank_of[inv.reshape(-1)]
            reqs_all, dest_all = reqs, dest_of_req
            reqs = reqs[order]
            dest_of_req = dest_of_req[order]
            n_req = len(reqs)
            self.req_bounds = [(0, n_req)]
            self.xchg_bounds = [(0, n_req)]
        # field slot = position in the (local) request stream
        self.req_begin, self.req_end = (0, n_req) if solo else self.req_bounds[rank]
        self.n_req_local = self.req_end - self.req_begin
        self.n_req_total = self.n_req_local * world if self.tile_exchange == "none" else n_req
        self._dest_of_req = dest_of_req
        slot_tbl = -np.ones((self.K, self.nchunks), np.int32)
        if share_fields:
            slot_tbl[dest_all, reqs_all["chunk_r"].astype(np.int64) * Wt + reqs_all["chunk_c"]] = mapped_slot
        else:
            slot_tbl[dest_of_req, reqs["chunk_r"].astype(np.int64) * Wt + reqs["chunk_c"]] = np.arange(n_req)
        self.agent_bounds = [pdist.slab(self.N, r, world) for r in range(world)]

        offs, members = navhip.flock_csr(ag["flock"], self.K)
        targets = synth.cell_centre(Wt, H, dests[:, 0], dests[:, 1])
        self.a0, self.a1 = (0, self.N) if solo else pdist.slab(self.N, rank, world)
        self.grid = grid
        self.map_cells = grid.size

        # ---- device state ----------------------------------------------------------------------
        def dev(a):
            return torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)

        self.tick_no = 0
        self.overlap = True
        if obstacles:
            self.d_moves = dev(self._moves_host.view(np.uint8).reshape(obstacle_ticks, self.n_moves, 24))

        self.d_reqs = dev(reqs.view(np.uint8).reshape(n_req, 32))
        self.pool = torch.zeros((n_req, 4096), dtype=torch.uint8, device=self.dev)
        n = self.N
        self.t = {
            "pos_xz": dev(ag["pos"]), "vel_xz": dev(ag["vel"]), "radius": dev(ag["radius"]),
            "max_speed": dev(ag["max_speed"]), "speed": dev(ag["speed"]),
            "flags": dev(np.full(n, navhip.ENTITY_FLAG_MOVABLE, np.uint32)),
            "state": dev(np.zeros(n, np.uint8)), "has_dest_los": dev(np.zeros(n, np.uint8)),
            "flock": dev(ag["flock"]), "flock_target_xz": dev(targets.astype(np.float32)),
            "flock_offsets": dev(offs), "flock_members": dev(members),
            "flock_field_slot": dev(slot_tbl), "field_pool": self.pool,
        }
        self.new_pos = torch.zeros((n, 2), dtype=torch.float32, device=self.dev)
        self.new_vel = torch.zeros((n, 2), dtype=torch.float32, device=self.dev)
        self.status = torch.zeros(n, dtype=torch.uint8, device=self.dev)
        self.res4 = torch.zeros((n, 4), dtype=torch.float32, device=self.dev) if world > 1 else None
        # (parity tests: the desired direction / preferred velocity every agent was stepped with)
        self.vdes_out = torch.zeros((n, 2), dtype=torch.float32, device=self.dev) if debug_outputs else None
        self.vpref_out = torch.zeros((n, 2), dtype=torch.float32, device=self.dev) if debug_outputs else None
        # host copies of what the job was built from (parity tests replay it through the reference)
        self.host = {"reqs": reqs, "dest_of_req": dest_of_req, "slot_tbl": slot_tbl, "dests": dests,
                     "targets": targets.astype(np.float32), "flock": ag["flock"], "flock_offsets": offs,
                     "flock_members": members, "radius": ag["radius"], "max_speed": ag["max_speed"],
                     "speed": ag["speed"], "liid": liid}
        # ---- "has_dest_los computed by the reference LOS code" -----------------------
        # los=True: the LOS fields of every (destination, chunk) the reference planner would hold -- its own
        # N_LOSFieldCreate chain, from the fixture -- are built on the device (navhip_build_los_dev, level by level
        # along the chain) and every agent's has_dest_los is answered per tick from them (NAVHIP_LOS_LOOKUP:
        # N_HasDestLOS, nav.c:4026).  Built once at start-up like the reference's LOS cache (a static map).
        self.los_source = "has_dest_los = 0 for every agent (no LOS fields)"
        self.n_los = 0
        if los:
            lc = synth.planner_los(grid, dests) if (world == 1 or shared_map) else None
            if lc is None:
                self.los_source = "has_dest_los = 0: no planner LOS fixture for this world"
            else:
                self._build_los(lc, dests, dev)
        # the agent chain: the library's own stream (a hardware queue to itself, the same one for every context of the
        # process -- navhip_stream_main; NAVTICK_TORCH_STREAM=1: a stream of torch's pool, as rounds 1-5 had it, for the A/B)
        if os.environ.get("NAVTICK_TORCH_STREAM") == "1":
            self.stream = tcuda.Stream(device=self.dev, priority=-1)
        else:
            self.stream = tcuda.ExternalStream(self.ctx.stream_main(), device=self.dev)
        # multi-GPU: the slab all-gather of tick t runs on its own stream and is only awaited by the
        # snapshot consumers of tick t+1 (spatial hash + cohesion, then the agent step); the field
        # builds of tick t+1 do not read positions and overlap with it
        self.pipelined = world > 1 and not solo
        # (the library's own streams for everything beside the agent chain: each has a hardware queue to itself, on a pipe of
        # the command processor that self.stream's queue does not sit on -- navhip_stream_beside)
        self.comm = tcuda.ExternalStream(self.ctx.stream_beside(self.stream.cuda_stream), device=self.dev) if self.pipelined else None
        # exchange = "navhip": the slab all-gather goes through the library's own C entry point
        # (navhip_comm_allgather_step_dev: librccl called directly -- what a C host uses) instead of
        # torch.distributed; rank 0's communicator id travels over the process group that launched us
        self.exchange_mode = exchange if self.pipelined else "none"
        if self.exchange_mode == "navhip":
            import torch.distributed as tdist
            box = [navhip.comm_unique_id() if rank == 0 else None]
            tdist.broadcast_object_list(box, src=0)
            self.ctx.comm_init(rank, world, box[0])
            self._bounds = np.array([b for b, _ in self.agent_bounds] + [self.N], np.int32)
        self.ev_step = tcuda.Event()
        self.ev_comm = tcuda.Event()
        self._comm_pending = False
        self._stepped = False
        self._make_structs()
        if obstacles:
            # the pool starts fully built (untimed), afterwards only changed chunks are repaired
            full = reqs.copy()
            full["flags"] = navhip.REQ_LIVE_IIDS
            d_full = dev(full.view(np.uint8).reshape(n_req, 32))
            self.ctx.build_fields_dev(d_full, n_req, self.pool, stream=self.stream.cuda_stream)
            self.stream.synchronize()
        # pipeline_fields: the fields tick t+1 samples are built DURING tick t, on their own stream, into
        # the other half of a double-buffered pool, starting once the narrow, serial front of tick t's
        # agent step (spatial hash + neighbour walk) is through.  Same work per tick -- one rebuild of
        # every chunk field, one step of every agent --, same results (the static map does not depend on
        # the agents); the field builds just stop gating the agent step of their own tick.  (With moving
        # obstacles the blocker updates of tick t+1 would race with the probes of tick t: not pipelined.)
        self.pipeline_fields = bool(pipeline_fields) and not obstacles
        if self.pipeline_fields:
            # Where the builds of tick t+1 run inside tick t is a scheduling choice, measured (round 3, after the
            # front of the step had become a third shorter; 20 ticks after 5 / 100 ticks, ms per tick):
            #   configs[2] (16 384 chunk fields, 0.09 ms alone): behind the NEIGHBOUR WALK on 5 of the 8 XCDs
            #     0.316-0.320 / 0.455-0.460 (128 ... 176 CUs all the same); on 192: 0.327 / 0.472; on all 256:
            #     0.336 / 0.477; started with the tick on 192 (round 2's choice): 0.340 / 0.483.  The front --
            #     spatial hash, neighbour walk: the critical path -- then only shares the chip with the cohesion
            #     kernel, and the ClearPath phase keeps three XCDs to itself.
            #   configs[3] (131 072 chunk fields, 0.58 ms alone -- as long as the rest of the tick): started
            #     WITH THE TICK on ALL compute units 0.97; with the tick on 224 / 192: 1.01 / 1.08; behind the
            #     neighbour walk on 224 / 160: 0.99 / 1.20.
            #   configs[1] (4 096 chunk fields, 45 us): no difference (0.259-0.265).
            ncu_all = tcuda.get_device_properties(self.dev).multi_processor_count
            long_build = self.n_req_local >= 65536
            ncu = int(os.environ.get("NAVTICK_FIELD_CUS", str(ncu_all if long_build else ncu_all * 5 // 8)))
            if 0 < ncu < ncu_all:
                self.fstream = tcuda.ExternalStream(self.ctx.stream_beside(self.stream.cuda_stream, ncu_all - ncu, ncu), device=self.dev)
            else:
                self.fstream = tcuda.ExternalStream(self.ctx.stream_beside(self.stream.cuda_stream), device=self.dev)
            self.fields_after = os.environ.get("NAVTICK_FIELDS_AFTER", "start" if long_build else "neighbours")
            # nothing wide is enqueued on self.stream between prefetch and step: the front stays on it
            # ... and the snapshot buffers ping-pong: the one a step read is next written by the ClearPath
            # kernels of the following step
            self.prefetch_flags = navhip.PREFETCH_FRONT_INLINE | navhip.PREFETCH_SNAPSHOT_HELD
            self.pool_next = torch.zeros_like(self.pool)
            self.ev_fields, self.ev_fields_next = tcuda.Event(), tcuda.Event()
            self.fev = []
            if self.n_req_local:                       # the fields of tick 0 (start-up, untimed)
                self.ctx.build_fields_dev(self.d_reqs[self.req_begin:self.req_end], self.n_req_local,
                                          self.pool[self.req_begin:self.req_end], stream=self.stream.cuda_stream)
            if self.tile_exchange != "none" and not self.solo:
                with tcuda.stream(self.stream):
                    pdist.exchange_rows(self.pool, self.xchg_bounds, self.rank, self.world)
            self.ev_fields.record(self.stream)
            self.stream.synchronize()
        if flow_velocities:
            self._flow_aligned_velocities()
        if not hasattr(self, "velocity_source"):
            self.velocity_source = "N(0, 0.35) per component (synth.agents)"
        # driver: who enqueues a tick.  "c" = the library's own loop (navhip_tick_*, csrc/tick_api.hip: ONE call per tick,
        # the schedule below in C) -- for every world whose baked tiles do not travel; "python" = this file's compute() /
        # exchange() / advance(), the reference implementation of that schedule, which the C loop is tested against.
        # serial (C driver only; NAVTICK_SERIAL=1): the whole tick on ONE stream, no side streams and no events -- the
        # host side shrinks to 48 us per tick, the tick does not (0.455 against 0.342 ms at configs[2]: the overlap of
        # cohesion / ClearPath / field builds is gone; profiles/r05_host_overhead_c.txt).  An option for a host that
        # wants one stream, not a default.
        self.driver = driver if (self.tile_exchange == "none" or self.solo) else "python"
        if serial is None:
            serial = os.environ.get("NAVTICK_SERIAL") == "1"
        self.serial = bool(serial)
        # time_fields (C driver): HIP events on the FIELD stream around the builds of every fourth tick -- how long the
        # builds take inside the tick, beside the agent step (navhip_tick_info.fields_ms; field_build_times())
        self.time_fields = bool(time_fields)
        self._ctick = None
        self.tick_driver = "python (tick.py)"
        self.ev = []                   # (phase, start_event, end_event) of the timed steps
        self.tick_ev = []              # one event at the start of every tick_every-th recorded tick
        self.tick_every, self._tick_rec = 5, 0
        self.record = False
        self.mark_every = 20
        if verbose:
            print("[rank %d] setup %.1fs: %d chunk-field requests (%d local), %d agents (%d local)"
                  % (rank, time.time() - t0, n_req, self.n_req_local, n, self.a1 - self.a0), flush=True)

    def _build_los(self, lc, dests, dev):
        """The planner's LOS chain on the device: pool slot = position in level order (level = hops from the
        destination chunk along the chain), one navhip_build_los_dev per level, each field from its predecessor."""
        Wt = self.Wt
        n = len(lc["dest"])
        key = lc["dest"] * self.nchunks + lc["chunk_r"] * Wt + lc["chunk_c"]
        pkey = lc["dest"] * self.nchunks + (lc["chunk_r"] + lc["prev_dr"]) * Wt + (lc["chunk_c"] + lc["prev_dc"])
        has_prev = (lc["prev_dr"] != 0) | (lc["prev_dc"] != 0)
        index_of = {int(k): i for i, k in enumerate(key)}
        level = np.zeros(n, np.int64)
        prev_i = np.full(n, -1, np.int64)
        for i in range(n):                       # (creation order: a predecessor always comes first)
            if has_prev[i]:
                prev_i[i] = index_of[int(pkey[i])]
                level[i] = level[prev_i[i]] + 1
        order = np.argsort(level, kind="stable")
        slot_of = np.empty(n, np.int64)
        slot_of[order] = np.arange(n)
        reqs = np.zeros(n, navhip.LOS_REQ_DTYPE)
        reqs["faction_id"] = navhip.FACTION_ID_NONE
        reqs["chunk_r"], reqs["chunk_c"] = lc["chunk_r"][order], lc["chunk_c"][order]
        d = lc["dest"][order]
        reqs["target_chunk_r"], reqs["target_chunk_c"] = dests[d, 0] // 64, dests[d, 1] // 64
        reqs["target_tile_r"], reqs["target_tile_c"] = dests[d, 0] % 64, dests[d, 1] % 64
        reqs["prev_dr"], reqs["prev_dc"] = lc["prev_dr"][order], lc["prev_dc"][order]
        prev_slot = np.where(prev_i[order] >= 0, slot_of[np.maximum(prev_i[order], 0)], 0)
        d_reqs = dev(reqs.view(np.uint8).reshape(n, 16))
        d_prev_slot = dev(prev_slot)
        self.los_pool = torch.zeros((n, 4096), dtype=torch.uint8, device=self.dev)
        lv = level[order]
        bounds = np.searchsorted(lv, np.arange(lv.max() + 2))
        tcuda.synchronize(self.dev)
        for L in range(len(bounds) - 1):
            b, e = int(bounds[L]), int(bounds[L + 1])
            if e == b:
                continue
            d_prev = self.los_pool.index_select(0, d_prev_slot[b:e]) if L > 0 else None
            tcuda.synchronize(self.dev)     # (torch's stream -> the library's: start-up, untimed)
            self.ctx.build_los_dev(d_reqs[b:e], e - b, d_prev, self.los_pool[b:e])
            self.ctx.sync()
        tbl = -np.ones((self.K, self.nchunks), np.int32)
        tbl[lc["dest"], lc["chunk_r"] * Wt + lc["chunk_c"]] = slot_of
        self.t["los_pool"] = self.los_pool
        self.t["flock_los_slot"] = dev(tbl)
        self.t["has_dest_los"] = torch.full((self.N,), navhip.LOS_LOOKUP, dtype=torch.uint8, device=self.dev)
        self.host["los"] = {"reqs": reqs, "slot_tbl": tbl, "levels": len(bounds) - 1, "prev_slot": prev_slot,
                            "level": lv}
        self.n_los = n
        self.los_source = ("device lookup (NAVHIP_LOS_LOOKUP) in %d LOS fields built by navhip_build_los from the "
                           "reference planner's N_LOSFieldCreate chain (fixture), %d levels" % (n, len(bounds) - 1))

    def _flow_aligned_velocities(self):
        """initial velocity = 0.5 max along the sampled flow direction + N(0, 0.1).  The
        directions are the ones the device samples (one untimed step of the start-up pool); seed 11."""
        n = self.N
        keep = self.vdes_out, self.vpref_out
        self.vdes_out = torch.zeros((n, 2), dtype=torch.float32, device=self.dev)
        self.vpref_out = torch.zeros((n, 2), dtype=torch.float32, device=self.dev)
        if not self.pipeline_fields and self.n_req_local and not self.n_obstacles:
            self.ctx.build_fields_dev(self.d_reqs[self.req_begin:self.req_end], self.n_req_local,
                                      self.pool[self.req_begin:self.req_end], stream=self.stream.cuda_stream)
        self._make_structs()
        wb, we = self.world_s.work_begin, self.world_s.work_end
        self.world_s.work_begin, self.world_s.work_end = 0, n          # (every rank samples every agent: replicated)
        self.ctx.agent_step_dev(self.world_s, self.out_s, stream=self.stream.cuda_stream)
        self.stream.synchronize()
        self.ctx.sync()
        vdes = self.vdes_out.cpu().numpy()
        rng = np.random.RandomState(11)
        speed = self.host["max_speed"][:, None] / float(self.hz)
        vel = (0.5 * speed * vdes + rng.normal(0.0, 0.1, size=(n, 2))).astype(np.float32)
        self.t["vel_xz"] = torch.from_numpy(vel).to(self.dev)
        if self.world > 1 and not self.solo:
            # (a rank only holds the fields its own agents sample: every rank keeps its slab's rows)
            pdist.exchange_rows(self.t["vel_xz"], self.agent_bounds, self.rank, self.world)
        self.vdes_out, self.vpref_out = keep
        self.world_s.work_begin, self.world_s.work_end = wb, we
        self._make_structs()
        self.velocity_source = "0.5 max_speed/hz along the sampled flow direction + N(0, 0.1) (SURVEY 8(d))"

    def _make_structs(self):
        arrays = dict(self.t)
        # (a rank that steps a slab: this driver never changes its flock tables -- the promise that lets the
        # library carry the cohesion term's lane grouping from tick to tick, navhip.h: static_epoch)
        if (self.a0, self.a1) != (0, self.N):
            arrays["static_epoch"] = 1
        self.world_s, self._keep = navhip.make_world(self.Wt, self.H, arrays, hz=self.hz)
        self.world_s.work_begin, self.world_s.work_end = self.a0, self.a1
        self.out_s = navhip.StepOut()
        self.out_s.vel_xz = self.new_vel.data_ptr()
        self.out_s.new_pos_xz = self.new_pos.data_ptr()
        self.out_s.status = self.status.data_ptr()
        if self.vdes_out is not None:
            self.out_s.vdes_xz = self.vdes_out.data_ptr()
            self.out_s.vpref_xz = self.vpref_out.data_ptr()

    def _mark(self, name):
        # phase timing by HIP events on the launch stream, on every `mark_every`-th recorded tick:
        # timing events are not free (a marker packet between back-to-back kernels; six per tick
        # cost ~4 % of the tick when recorded every tick)
        if not self.record or self.tick_no % self.mark_every:
            return None
        e = tcuda.Event(enable_timing=True)
        e.record(self.stream)
        return (name, e)

    # ---- the C driver ---------------------------------------------------------------------------------------------
    def _c_tick(self):
        """The navhip_tick of the current buffers (made on first use; dropped whenever the Python path steps)."""
        if self._ctick is not None:
            return self._ctick
        d = navhip.TickDesc()
        C.memmove(C.byref(d.world), C.byref(self.world_s), C.sizeof(navhip.World))
        d.pos_xz_1, d.vel_xz_1 = self.new_pos.data_ptr(), self.new_vel.data_ptr()
        d.status = self.status.data_ptr()
        if self.vdes_out is not None:
            d.vdes_xz, d.vpref_xz = self.vdes_out.data_ptr(), self.vpref_out.data_ptr()
        d.n_reqs, d.req_slot0 = self.n_req_local, self.req_begin
        d.dev_reqs = self.d_reqs[self.req_begin:self.req_end].data_ptr() if self.n_req_local else None
        keep = [self.world_s, self._keep, self.new_pos, self.new_vel, self.status, self.d_reqs, self.pool]
        d.stream = self.stream.cuda_stream
        if self.pipeline_fields:
            d.field_pool_1 = self.pool_next.data_ptr()
            d.field_stream = self.fstream.cuda_stream
            d.fields_stage = navhip.STAGE_START if self.fields_after == "start" else navhip.STAGE_NEIGHBOURS
            keep.append(self.pool_next)
        if self.n_obstacles:
            d.dev_moves, d.n_moves, d.n_move_ticks = self.d_moves.data_ptr(), self.n_moves, int(self.d_moves.shape[0])
            d.move_tick0 = self.tick_no % int(self.d_moves.shape[0])
        if self.pipelined and self.exchange_mode == "navhip":
            self._bounds_c = np.ascontiguousarray(self._bounds, np.int32)
            d.bounds = self._bounds_c.ctypes.data
            d.comm_stream = self.comm.cuda_stream
        # (the snapshot arrays are this object's own: nothing else writes them or enqueues on its stream between ticks)
        d.flags = ((navhip.TICK_SERIAL if self.serial else 0) | (navhip.TICK_TIME_FIELDS if self.time_fields else 0)
                   | navhip.TICK_OWNS_SNAPSHOT)
        self._ctick = navhip.Tick(self.ctx, d, keep)
        self._ctick_tick0 = self.tick_no
        self.tick_driver = "c (navhip_tick_run%s)" % (", one stream" if self.serial else "")
        return self._ctick

    def field_build_times(self):
        """(sum of milliseconds, samples) of the field builds the C tick has timed so far (time_fields=True)."""
        if self._ctick is None:
            return 0.0, 0
        info = self._ctick.info()
        return info.fields_ms * info.fields_samples, int(info.fields_samples)

    def _c_tick_drop(self):
        if self._ctick is not None:
            self._ctick.sync()
            self.c_tick_info = self._ctick.info()
            self._ctick.close()
            self._ctick = None
            if self.pipeline_fields:
                # (the Python path's event of "this tick's fields are built": they are -- everything was waited for)
                self.ev_fields.record(self.stream)
            self._make_structs()

    def _step_c(self):
        T = self._c_tick()
        if self.pipelined and self.exchange_mode != "navhip":
            # torch.distributed between the two halves of the C tick: the tick starts behind the previous exchange
            if self._comm_pending:
                self.stream.wait_event(self.ev_comm)
            T.compute()
            self.ev_step.record(self.stream)
            self.exchange()
            T.advance()
        else:
            T.run(1)
        # (the Python view of the ping-pong, without rebuilding the structs: the C tick has its own two)
        self.t["pos_xz"], self.new_pos = self.new_pos, self.t["pos_xz"]
        self.t["vel_xz"], self.new_vel = self.new_vel, self.t["vel_xz"]
        if self.pipeline_fields and not self.serial:
            self.pool, self.pool_next = self.pool_next, self.pool
            self.t["field_pool"] = self.pool
        self.tick_no += 1

    def step(self):
        """One tick, asynchronous on self.stream."""
        use_c = self.driver == "c" and not (self.record and self.mark_every <= 1)
        if self.record:
            # one timing event every `tick_every` ticks: an event on the agent stream is a packet on the
            # tick's critical path (recording every tick cost 1.5-2 % of the tick)
            if self._tick_rec % self.tick_every == 0:
                e = tcuda.Event(enable_timing=True)
                e.record(self.stream)
                self.tick_ev.append(e)
            self._tick_rec += 1
        if use_c:
            return self._step_c()
        self._c_tick_drop()
        self.compute()
        self.exchange()
        self.advance()

    def compute(self):
        """Field builds + velocity step of this rank's share (everything up to the exchange)."""
        s = self.stream
        self._marks = marks = []
        if self.pipelined and self.overlap:
            # behind the previous tick's all-gather, concurrently with the field builds below
            with tcuda.stream(self.comm):
                self.ctx.agent_prefetch_dev(self.world_s, stream=self.comm.cuda_stream)
        if self.pipeline_fields:
            return self._compute_pipelined(marks)
        with tcuda.stream(s):
            # snapshot-only parts of the agent step (spatial hash, cohesion: they read no nav plane)
            # start now on the library's side streams and overlap with the blocker updates and the
            # field builds below
            if self.overlap and not self.pipelined:
                self.ctx.agent_prefetch_dev(self.world_s, stream=s.cuda_stream)
            if self.n_obstacles:
                marks.append(self._mark("blockers"))
                t = self.tick_no % self.d_moves.shape[0]
                self.ctx.blockers_circles_dev(self.d_moves[t], self.n_moves, stream=s.cuda_stream)
            marks.append(self._mark("fields"))
            if self.n_req_local:
                self.ctx.build_fields_dev(self.d_reqs[self.req_begin:self.req_end], self.n_req_local,
                                          self.pool[self.req_begin:self.req_end], stream=s.cuda_stream)
            if self.n_obstacles:
                self.ctx.clear_changed(stream=s.cuda_stream)
            marks.append(self._mark("gather_tiles"))
            if self.tile_exchange != "none" and not self.solo:
                pdist.exchange_rows(self.pool, self.xchg_bounds, self.rank, self.world)
            marks.append(self._mark("agents"))
            if self._comm_pending:
                s.wait_event(self.ev_comm)            # the other ranks' rows of the snapshot
            self.ctx.agent_step_dev(self.world_s, self.out_s, stream=s.cuda_stream)
            marks.append(self._mark("gather_agents"))
            self.ev_step.record(s)

    def _compute_pipelined(self, marks):
        s, f = self.stream, self.fstream
        # In a jam the workgroup ClearPath searches hold every register of the chip for milliseconds: a build
        # that starts behind the neighbour walk has not got its workgroups resident by then, finishes after
        # them, and the next step waits for it (crowded world: 5.3 -> 6.8 ms per tick).  The list lengths of
        # the last steps arrive in pinned memory without a wait: from a jam's worth of such searches on, the
        # builds start with the tick.
        stage = self.fields_after
        if stage == "neighbours" and self.ctx.step_lists_peek()[4] >= 8192:
            stage = "start"
        if stage == "start" and self.pipelined:
            f.wait_stream(s)                      # (the end of the previous tick)
        if not self.pipelined:
            with tcuda.stream(s):
                # (this object's arrays and stream: the snapshot of a tick is the output of the last one)
                follows = navhip.PREFETCH_FOLLOWS_STEP if self._stepped else 0
                self.ctx.agent_prefetch_dev(self.world_s, stream=s.cuda_stream, flags=self.prefetch_flags | follows)
        with tcuda.stream(s):
            marks.append(self._mark("agents"))
            s.wait_event(self.ev_fields)                  # this tick's fields (built during the last one)
            if self._comm_pending:
                s.wait_event(self.ev_comm)
            self.ctx.agent_step_dev(self.world_s, self.out_s, stream=s.cuda_stream)
            self._stepped = True
            marks.append(self._mark("gather_agents"))
            if self.pipelined:
                self.ev_step.record(s)
        # the fields of the NEXT tick: enqueued behind the step (whose own wait for the cohesion term is the launch that
        # says "the neighbour walk is done"), started by the device as soon as that is so
        timed = self.record and self.tick_no % self.mark_every == 0
        with tcuda.stream(f):
            if stage == "neighbours":
                self.ctx.stream_wait_stage(f.cuda_stream, navhip.STAGE_NEIGHBOURS)
            elif not self.pipelined:
                # (the start of the prefetch just enqueued: the end of the previous tick, without
                # another event on the agent stream)
                self.ctx.stream_wait_stage(f.cuda_stream, navhip.STAGE_START)
            if timed:
                e0 = tcuda.Event(enable_timing=True)
                e0.record(f)
            if self.n_req_local:
                self.ctx.build_fields_dev(self.d_reqs[self.req_begin:self.req_end], self.n_req_local,
                                          self.pool_next[self.req_begin:self.req_end], stream=f.cuda_stream)
            if timed:
                e1 = tcuda.Event(enable_timing=True)
                e1.record(f)
            if self.tile_exchange != "none" and not self.solo:
                pdist.exchange_rows(self.pool_next, self.xchg_bounds, self.rank, self.world)
            if timed:
                e2 = tcuda.Event(enable_timing=True)
                e2.record(f)
                self.fev.append((e0, e1, e2))
            self.ev_fields_next.record(f)

    def _comm_behind_step(self):
        # the exchange stream behind the step's outputs: the word the step stored in device memory when it ended (2-3 us),
        # or -- a step that ran on one stream has none -- the event recorded behind it (12 us between two queues)
        if not self.ctx.stream_wait_stage(self.comm.cuda_stream, navhip.STAGE_END, check=False):
            self.comm.wait_event(self.ev_step)

    def exchange(self):
        """The slab results (new position + velocity) of every rank -> every rank."""
        if not self.pipelined:
            return
        if self.exchange_mode == "navhip":
            self._comm_behind_step()
            self.ctx.comm_allgather_step_dev(self.new_pos, self.new_vel, self._bounds, stream=self.comm.cuda_stream)
            self.ev_comm.record(self.comm)
            self._comm_pending = True
            return
        with tcuda.stream(self.comm):
            self._comm_behind_step()
            # ONE collective per tick: this rank's rows of [new position | new velocity] (16 B per
            # agent) packed into one buffer, all-gathered, unpacked
            b, e = self.agent_bounds[self.rank]
            self.res4[b:e, 0:2] = self.new_pos[b:e]
            self.res4[b:e, 2:4] = self.new_vel[b:e]
            pdist.exchange_rows(self.res4, self.agent_bounds, self.rank, self.world)
            self.new_pos.copy_(self.res4[:, 0:2])
            self.new_vel.copy_(self.res4[:, 2:4])
            self.ev_comm.record(self.comm)
        self._comm_pending = True

    def advance(self):
        """Advance the snapshot: ping-pong the position / velocity buffers."""
        with tcuda.stream(self.stream):
            self._marks.append(self._mark("end"))
            self.t["pos_xz"], self.new_pos = self.new_pos, self.t["pos_xz"]
            self.t["vel_xz"], self.new_vel = self.new_vel, self.t["vel_xz"]
            if self.pipeline_fields:
                self.pool, self.pool_next = self.pool_next, self.pool
                self.ev_fields, self.ev_fields_next = self.ev_fields_next, self.ev_fields
                self.t["field_pool"] = self.pool
            self._make_structs()
        if self.record and self._marks and self._marks[0] is not None:
            self.ev.append(self._marks)
        self.tick_no += 1

    def phase_ms(self):
        """Average HIP-event duration of every phase over the recorded steps."""
        out = {}
        for marks in self.ev:
            for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
                out.setdefault(n0, []).append(e0.elapsed_time(e1))
        for e0, e1, e2 in getattr(self, "fev", []):
            out.setdefault("fields", []).append(e0.elapsed_time(e1))
            out.setdefault("gather_tiles", []).append(e1.elapsed_time(e2))
        return {k: float(np.mean(v)) for k, v in out.items()}

    def tick_ms(self):
        """GPU-timeline milliseconds per tick, one value per window of tick_every recorded ticks (start
        event to start event)."""
        return [a.elapsed_time(b) / self.tick_every for a, b in zip(self.tick_ev[:-1], self.tick_ev[1:])]

    def sync(self):
        if self._ctick is not None:
            self._ctick.sync()
        if self.comm is not None:
            self.comm.synchronize()
        if self.pipeline_fields:
            self.fstream.synchronize()
        self.stream.synchronize()
        tcuda.synchronize(self.dev)

    def close(self):
        self.sync()
        if self._ctick is not None:
            self.c_tick_info = self._ctick.info()
            self._ctick.close()
            self._ctick = None
        self.ctx.close()
