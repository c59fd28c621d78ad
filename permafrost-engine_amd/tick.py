"""The benchmark / demo driver of one navigation tick, everything resident in HBM.

One tick = (1) rebuild every chunk field of this rank's share of the flow fields into the field
pool, (2) [multi-GPU] all-gather the baked tiles, (3) velocity step + position accept for this
rank's slab of agents, sampling the pool on the device, (4) [multi-GPU] all-gather the slab
results, (5) advance the snapshot (pos <- new_pos, vel <- new velocity).  Nothing is cached
between ticks: the worst case of the reference's tick, where every cached field was invalidated
(N_ApplyDeferredInvalidations, nav.c:2208).

PyTorch supplies device buffers, the stream and torch.distributed; all compute is libnavhip.
"""
import ctypes as C
import time

import numpy as np
import torch

from . import dist as pdist
from . import navhip, synth


class NavTick:
    def __init__(self, chunk_w=16, fields_per_rank=64, agents_per_rank=100_000, rank=0, world=1,
                 device=0, hz=20, seed_map=1234, verbose=False, obstacles=0, move_frac=0.01,
                 obstacle_ticks=128):
        self.rank, self.world, self.device_index = rank, world, device
        self.dev = torch.device("cuda", device)
        torch.cuda.set_device(self.dev)
        self.W = chunk_w
        self.nchunks = chunk_w * chunk_w
        self.K = fields_per_rank * world            # flow fields (destinations) in the whole job
        self.N = agents_per_rank * world            # agents in the whole job
        self.hz = hz
        t0 = time.time()

        # ---- synthetic map + request stream, identical on every rank --------
        grid = synth.cost_grid(chunk_w, chunk_w, seed=seed_map)
        self.ctx = navhip.NavContext(chunk_w, chunk_w, device=device)
        self.ctx.upload_plane(0, navhip.PLANE_COST_BASE, synth.to_chunks(grid))
        self.ctx.upload_plane(0, navhip.PLANE_BLOCKERS, np.zeros((chunk_w, chunk_w, 64, 64), np.uint16))
        self.n_obstacles = obstacles
        blockers = None
        if obstacles:
            # configs[4]: dynamic obstacles (circles, radius U(2,6) wu, seed 99) dropped through the
            # device N_BlockersIncref path; every tick `move_frac` of them move (decref + incref)
            rng = np.random.RandomState(99)
            cells = synth.passable_cells(grid)
            pos = synth.cell_centre(chunk_w, chunk_w, *cells[rng.randint(len(cells), size=obstacles)].T)
            circ = np.zeros(obstacles, navhip.CIRCLE_DTYPE)
            circ["x"], circ["z"] = pos[:, 0], pos[:, 1]
            circ["radius"] = rng.uniform(2.0, 6.0, obstacles)
            circ["delta"] = 1
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
                npos = synth.cell_centre(chunk_w, chunk_w, *cells[rng.randint(len(cells), size=nmove)].T)
                cur["x"][who], cur["z"][who] = npos[:, 0], npos[:, 1]
                moves[t, nmove:] = cur[who]
                moves[t, nmove:]["delta"] = 1
            self.n_moves = 2 * nmove
            self._moves_host = moves
        self.ctx.relabel_local_islands(0)           # n_update_local_island_field on the device
        liid = synth.from_chunks(self.ctx.download_plane(0, navhip.PLANE_LOCAL_ISLANDS))
        dests = synth.destinations(grid, self.K, seed=42)
        cols = synth.whole_map_requests(grid, dests, liid)
        n_req = len(cols["type"])
        reqs = navhip.make_reqs(n_req)
        for k in synth.REQ_FIELDS:
            reqs[k] = cols[k]
        if obstacles:
            reqs["flags"] = navhip.REQ_LIVE_IIDS | navhip.REQ_IF_CHANGED
        # requests are emitted destination-major: field slot = position in the stream
        dest_of_req = cols["dest"]
        self.n_req_total = n_req
        self._dest_of_req = dest_of_req
        slot_tbl = -np.ones((self.K, self.nchunks), np.int32)
        slot_tbl[dest_of_req, cols["chunk_r"] * chunk_w + cols["chunk_c"]] = np.arange(n_req)
        # this rank's slice of the request stream: whole destinations, contiguous
        self.req_bounds = pdist.request_slices(dest_of_req, self.K, world)
        self.req_begin, self.req_end = self.req_bounds[rank]
        self.n_req_local = self.req_end - self.req_begin
        self.agent_bounds = [pdist.slab(self.N, r, world) for r in range(world)]

        ag = synth.agents(grid, self.N, self.K, seed=7, hz=hz, blockers=blockers)
        offs, members = navhip.flock_csr(ag["flock"], self.K)
        targets = synth.cell_centre(chunk_w, chunk_w, dests[:, 0], dests[:, 1])
        self.a0, self.a1 = pdist.slab(self.N, rank, world)
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
        self.stream = torch.cuda.Stream(device=self.dev)
        self._make_structs()
        if obstacles:
            # the pool starts fully built (untimed), afterwards only changed chunks are repaired
            full = reqs.copy()
            full["flags"] = navhip.REQ_LIVE_IIDS
            d_full = dev(full.view(np.uint8).reshape(n_req, 32))
            self.ctx.build_fields_dev(d_full, n_req, self.pool, stream=self.stream.cuda_stream)
            self.stream.synchronize()
        self.ev = []                   # (phase, start_event, end_event) of the timed steps
        self.record = False
        if verbose:
            print("[rank %d] setup %.1fs: %d chunk-field requests (%d local), %d agents (%d local)"
                  % (rank, time.time() - t0, n_req, self.n_req_local, n, self.a1 - self.a0), flush=True)

    def _make_structs(self):
        arrays = dict(self.t)
        self.world_s, self._keep = navhip.make_world(self.W, self.W, arrays, hz=self.hz)
        self.world_s.work_begin, self.world_s.work_end = self.a0, self.a1
        self.out_s = navhip.StepOut()
        self.out_s.vel_xz = self.new_vel.data_ptr()
        self.out_s.new_pos_xz = self.new_pos.data_ptr()
        self.out_s.status = self.status.data_ptr()

    def _mark(self, name):
        if not self.record:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(self.stream)
        return (name, e)

    def step(self):
        """One tick, asynchronous on self.stream."""
        s = self.stream
        marks = []
        with torch.cuda.stream(s):
            if self.n_obstacles:
                marks.append(self._mark("blockers"))
                t = self.tick_no % self.d_moves.shape[0]
                self.ctx.blockers_circles_dev(self.d_moves[t], self.n_moves, stream=s.cuda_stream)
            # snapshot-only parts of the agent step (spatial hash, cohesion) start now on the
            # library's side streams and overlap with the field builds below
            if self.overlap:
                self.ctx.agent_prefetch_dev(self.world_s, stream=s.cuda_stream)
            marks.append(self._mark("fields"))
            if self.n_req_local:
                self.ctx.build_fields_dev(self.d_reqs[self.req_begin:self.req_end], self.n_req_local,
                                          self.pool[self.req_begin:self.req_end], stream=s.cuda_stream)
            if self.n_obstacles:
                self.ctx.clear_changed(stream=s.cuda_stream)
            marks.append(self._mark("gather_tiles"))
            pdist.exchange_rows(self.pool, self.req_bounds, self.rank, self.world)
            marks.append(self._mark("agents"))
            self.ctx.agent_step_dev(self.world_s, self.out_s, stream=s.cuda_stream)
            marks.append(self._mark("gather_agents"))
            pdist.exchange_rows(self.new_pos, self.agent_bounds, self.rank, self.world)
            pdist.exchange_rows(self.new_vel, self.agent_bounds, self.rank, self.world)
            marks.append(self._mark("end"))
            # advance the snapshot: ping-pong the position / velocity buffers
            self.t["pos_xz"], self.new_pos = self.new_pos, self.t["pos_xz"]
            self.t["vel_xz"], self.new_vel = self.new_vel, self.t["vel_xz"]
            self._make_structs()
        self.tick_no += 1
        if self.record:
            self.ev.append(marks)

    def phase_ms(self):
        """Average HIP-event duration of every phase over the recorded steps."""
        out = {}
        for marks in self.ev:
            for (n0, e0), (n1, e1) in zip(marks[:-1], marks[1:]):
                out.setdefault(n0, []).append(e0.elapsed_time(e1))
        return {k: float(np.mean(v)) for k, v in out.items()}

    def sync(self):
        self.stream.synchronize()
        torch.cuda.synchronize(self.dev)

    def close(self):
        self.sync()
        self.ctx.close()
