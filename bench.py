#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's config.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one navigation tick over one batch of synthetic input with everything resident in
HBM: rebuild all chunk fields of the flow fields (64 whole-map flow fields = 16 384 chunk fields
per GPU), then one velocity step (flow sampling -> steering forces -> neighbour gather ->
ClearPath -> truncate -> position accept) for 100 000 agents per GPU, then advance the snapshot.
At N=1 this is configs[2] of BASELINE.json ("1024x1024 map, 64 concurrent flow fields, 100 000
agents, 1xMI355X"), the configuration the target (>=1e7 agent-steps/s) is quoted on.  At N>1 the
job is weak-scaled: every GPU brings its own 64 flow fields and 100 000 agents on the same map;
field requests and agent slabs are sharded, baked tiles and slab results are all-gathered (RCCL).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
BYTES_PER_CELL = 4             # 1 B cost + 2 B blockers read, 1 B direction written
BYTES_PER_AGENT_STEP = 112     # 96 B record in, 8 B velocity + 8 B position out
PLANE_BYTES_PER_MAP_CELL = 3   # per tick, once: cost (1) + blockers (2)


def usable_cores():
    """Host cores this process may really use: min(affinity, cgroup CPU quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // per))
        except Exception:
            pass
    return min(n, 256)


def cpu_baseline(chunk_w, k_fields, n_agents, hz, budget_field_s=10.0, budget_agent_s=10.0):
    """The reference's own code (oracle/_ref) timed on this box's host cores on a bounded sample of
    the same workload.  Reported, never the target."""
    try:
        from oracle import pfref
        if not pfref.available():
            return None
        import numpy as np
        from permafrost_engine_amd import synth
        cores = usable_cores()
        grid = synth.cost_grid(chunk_w, chunk_w, seed=1234)
        nav = pfref.RefNav(synth.to_chunks(grid))      # the reference's portal / island build
        dests = synth.destinations(grid, k_fields, seed=42)
        # (i) chunk fields: the planner's own request stream for a sample of destinations
        cells = synth.passable_cells(grid)
        rng = np.random.RandomState(3)
        reqs = []
        for d in dests[:16]:
            for _ in range(6):
                a = cells[rng.randint(len(cells))]
                nav.request_path(synth.cell_centre(chunk_w, chunk_w, a[0], a[1]),
                                 synth.cell_centre(chunk_w, chunk_w, d[0], d[1]), clear_cache=True)
                r, _, _ = nav.trace()
                reqs.append(r)
        reqs = np.concatenate(reqs)
        reqs["inout"] = 0
        t_f1 = nav.field_bench(reqs[:256], reps=1, nthreads=1)
        n1 = min(len(reqs), 256)
        cells_per_s_1 = n1 * 4096 / t_f1
        # ~budget_field_s seconds of single-core work, spread over all cores
        reps = max(1, int(budget_field_s / (t_f1 / n1 * len(reqs))))
        t_f = nav.field_bench(reqs, reps=reps, nthreads=cores)
        cells_per_s = len(reqs) * reps * 4096 / t_f
        # (ii) velocity step: the full 100k-agent snapshot loaded, a slab of it stepped
        ag = synth.agents(grid, n_agents, k_fields, seed=7, hz=hz)
        targets = synth.cell_centre(chunk_w, chunk_w, dests[:, 0], dests[:, 1])
        dest_ids = []
        for f in range(k_fields):
            ok, did = nav.request_path(ag["pos"][f], targets[f], clear_cache=(f == 0))
            dest_ids.append(did)
        nav.trace()
        mv = pfref.RefMove(nav, ag["pos"], ag["vel"], ag["radius"], ag["max_speed"], ag["speed"],
                           np.full(n_agents, pfref.ENTITY_FLAG_MOVABLE, np.uint32),
                           np.zeros(n_agents, np.int32), ag["flock"], np.zeros(n_agents, np.uint8),
                           targets, np.array(dest_ids, np.uint32), hz=hz)
        vdes = np.zeros((n_agents, 2), np.float32)
        vdes[:, 0] = 1.0
        t_1, _ = mv.bench(vdes, reps=1, nthreads=1, begin=0, end=min(400, n_agents))
        per_agent = t_1 / min(400, n_agents)
        # ~budget_agent_s seconds of single-core work, spread over all cores
        m = int(min(n_agents, max(cores * 8, budget_agent_s / per_agent)))
        t_a, _ = mv.bench(vdes, reps=1, nthreads=cores, begin=0, end=m)
        pfref.RefMove.unload()
        return {
            "value": m / t_a, "unit": "agent-steps/s", "cores": cores, "kind": "reference",
            "sample": "reference movement.c move_velocity_work on %d of the %d agents (full snapshot "
                      "loaded, desired directions given), %d pthreads; reference N_FlowFieldUpdate on "
                      "%d planner-emitted chunk-field requests x%d" % (m, n_agents, cores, len(reqs), reps),
            "flow_field_cells_per_s": cells_per_s, "flow_field_cells_per_s_1core": cells_per_s_1,
            "agent_steps_per_s_1core": 1.0 / per_agent,
            "cores_note": "threads = usable cores (min of affinity and the cgroup cpu.max quota); "
                          "os.cpu_count() = %d" % (os.cpu_count() or 0),
            "cpu_work_s": {"fields": t_f1 / n1 * len(reqs) * reps, "agents": per_agent * m},
        }
    except Exception as exc:                      # the baseline is informational
        return {"value": None, "unit": "agent-steps/s", "cores": os.cpu_count(), "kind": "reference",
                "sample": "unavailable: %r" % (exc,)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--map", type=int, default=16, help="map side in chunks (16 = 1024x1024 cells)")
    ap.add_argument("--fields", type=int, default=64, help="whole-map flow fields per GPU")
    ap.add_argument("--agents", type=int, default=100_000, help="agents per GPU")
    ap.add_argument("--obstacles", type=int, default=0,
                    help="configs[4]: dynamic obstacles, 1%% moved per tick, incremental field repair")
    ap.add_argument("--tile-exchange", choices=("auto", "all"), default="auto",
                    help="multi-GPU: auto = only baked tiles that another rank's agents sample travel "
                         "(none in this workload: flocks are rank aligned); all = all-gather every tile "
                         "every tick (any agent may sample any field)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    from permafrost_engine_amd import dist as pdist
    import __graft_entry__ as ge
    ge.build_navhip()
    from permafrost_engine_amd import tick

    rank, world, local = pdist.init()
    if world != args.gpus and world > 1:
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libnavhip has no CPU fallback")
    torch.cuda.set_device(local)

    T = tick.NavTick(chunk_w=args.map, fields_per_rank=args.fields, agents_per_rank=args.agents,
                     rank=rank, world=world, device=local, verbose=(rank == 0 and False),
                     obstacles=args.obstacles, obstacle_ticks=args.warmup + args.steps + 8,
                     tile_exchange=args.tile_exchange)
    for _ in range(args.warmup):
        T.step()
    T.sync()
    pdist.barrier()
    torch.cuda.synchronize()
    T.record = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        T.step()
    T.sync()
    torch.cuda.synchronize()
    pdist.barrier()
    dt = time.perf_counter() - t0
    dt = pdist.max_over_ranks(dt, T.dev)

    phases = T.phase_ms()
    # per-kernel split of the agent phase: a few extra (untimed) ticks with the library's own HIP
    # events between its kernels, on the launch stream
    T.record = False
    T.ctx.set_profiling(True)
    ksplit = []
    for _ in range(5):
        T.step()
        T.sync()
        ksplit.append(T.ctx.last_step_ms())
    T.ctx.set_profiling(False)
    k_sp, k_coh, k_step = (float(sum(x[i] for x in ksplit) / len(ksplit)) for i in range(3))
    agents_total = T.N
    cells_total = T.n_req_total * 4096
    ms_per_step = dt / args.steps * 1e3
    value = agents_total * args.steps / dt

    # ---- roofline of the dominant kernel phase (HIP events on the launch stream) ---------------
    f_ms, a_ms = phases.get("fields", 0.0), phases.get("agents", 0.0)
    f_bytes = T.n_req_local * 4096 * BYTES_PER_CELL
    a_bytes = (T.a1 - T.a0) * BYTES_PER_AGENT_STEP + T.map_cells * PLANE_BYTES_PER_MAP_CELL
    f_gbs = f_bytes / (f_ms * 1e-3) / 1e9 if f_ms > 0 else 0.0
    a_gbs = a_bytes / (a_ms * 1e-3) / 1e9 if a_ms > 0 else 0.0
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    measured = {}
    if os.path.exists(tpath):
        try:
            measured = json.load(open(tpath))
        except Exception:
            measured = {}
    # What actually bounds these kernels is VALU issue, not HBM: next to the
    # HBM figures, report the instruction-issue floor from the committed SQ counters.
    sq = {}
    try:
        sq = json.load(open(os.path.join(ROOT, "profiles", "sq_counters.json")))["kernels"]
    except Exception:
        sq = {}

    def valu(names, measured_ms):
        ks = [k for k in sq if k.startswith(names)]
        if not ks or measured_ms <= 0:
            return None
        insts = sum(sq[k]["SQ_INSTS_VALU"] for k in ks)
        floor_ms = insts / 1024 * 4 / 2.4e9 * 1e3
        return {"valu_insts_per_launch": insts, "issue_floor_ms": floor_ms,
                "frac_of_issue_peak": floor_ms / measured_ms,
                "note": "wave64 VALU instructions (rocprofv3 SQ_INSTS_VALU, profiles/sq_counters.json) at "
                        "one per 4 cycles per SIMD, 1024 SIMDs, 2.4 GHz, over the measured phase time"}

    dom = "agents" if a_ms >= f_ms else "fields"
    roof = {
        "bound": "hbm", "kernel": "k_agent_step (+k_cohesion, spatial hash)" if dom == "agents" else "k_field_bfs",
        "achieved": a_gbs if dom == "agents" else f_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": (a_gbs if dom == "agents" else f_gbs) / HBM_PEAK_GBS,
        "traffic": measured.get(dom + "_bytes_per_launch", traffic),
        "avg_launch_ms": a_ms if dom == "agents" else f_ms,
        "algorithmic_bytes_per_launch": a_bytes if dom == "agents" else f_bytes,
        "launch": "one navhip_agent_step_dev call" if dom == "agents" else "one navhip_build_fields_dev call",
        "kernels_ms": {"k_sp_*": k_sp, "k_cohesion": k_coh, "k_agent_step": k_step} if dom == "agents" else None,
        "valu_issue": valu(("k_agent_",), a_ms) if dom == "agents" else valu(("k_field_", "k_coh", "k_sp_"), f_ms),
    }
    roof_other = {
        "bound": "hbm", "kernel": "k_field_bfs" if dom == "agents" else "k_agent_step (+k_cohesion, spatial hash)",
        "achieved": f_gbs if dom == "agents" else a_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": (f_gbs if dom == "agents" else a_gbs) / HBM_PEAK_GBS,
        "traffic": measured.get(("fields" if dom == "agents" else "agents") + "_bytes_per_launch"),
        "avg_launch_ms": f_ms if dom == "agents" else a_ms,
        "algorithmic_bytes_per_launch": f_bytes if dom == "agents" else a_bytes,
        "valu_issue": valu(("k_field_", "k_coh", "k_sp_"), f_ms) if dom == "agents" else valu(("k_agent_",), a_ms),
    }

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(args.map, args.fields, args.agents, 20)

    if rank == 0:
        line = {
            "metric": "agent-steps/sec (+ flow-field cells/sec), 1024^2 map, 100k agents per GPU, "
                      "64 whole-map flow fields per GPU rebuilt every tick",
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "u64-bitmask/u8 fields, f32 agents",
            "data": "synthetic",
            "config": {"workload": ("configs[4] (dynamic obstacles, incremental repair): " if args.obstacles else "")
                                   + "configs[2] per GPU: %dx%d-cell map region (%dx%d chunks), %d flow "
                                   "fields (%d chunk fields) + %d agents per GPU, fields rebuilt + agents "
                                   "stepped every tick%s" % (args.map * 64, args.map * 64, args.map, args.map,
                                                             args.fields, T.n_req_local, args.agents,
                                                             "" if world == 1 else
                                                             "; the %d regions tile one %dx%d-cell map"
                                                             % (world, T.H * 64, T.Wt * 64)),
                       "map_chunks": args.map, "flow_fields_per_gpu": args.fields,
                       "agents_per_gpu": args.agents, "hz": 20, "dynamic_obstacles": args.obstacles,
                       "parallelism": "regions (requests + agent slabs) sharded x%d; all-gather of slab "
                                      "results (16 B/agent); baked tiles: %s" % (world, T.tile_exchange)},
            ("flow_field_cells_kept_valid_per_s" if args.obstacles else "flow_field_cells_per_s"):
                cells_total * args.steps / dt,
            "flow_field_cells_per_s_kernel": (T.n_req_local * 4096 * world) / (f_ms * 1e-3) if f_ms > 0 else None,
            "agent_steps_per_s_kernel": agents_total / (a_ms * 1e-3) if a_ms > 0 else None,
            "phase_ms": phases,
            "roofline": roof,
            "roofline_secondary": roof_other,
            "cpu_baseline": cpu,
        }
        print(json.dumps(line), flush=True)
    T.close()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
