#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE.json's configs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {0,1,2,3,4}]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one navigation tick over one batch of synthetic input with everything resident in
HBM: rebuild every chunk field of the flow fields, then one velocity step (flow sampling ->
steering forces -> neighbour gather -> ClearPath -> truncate -> position accept) for every agent,
then advance the snapshot.  --config selects the BASELINE.json configuration:

  0  256x256 map, 1 flow field, 1 000 agents: the reference's own CPU-runnable case.  The GPU path is
     timed on it and the WHOLE workload is also timed through the reference build on the host
     (unsampled cpu_baseline).
  1  1024x1024 map, 16 flow fields, 50 000 agents
  2  1024x1024 map, 64 flow fields (16 384 chunk fields), 100 000 agents -- the configuration the
     target (>= 1e7 agent-steps/s) is quoted on; DEFAULT.  With --gpus N > 1 it is weak-scaled:
     every GPU brings its own region of the map, 64 flow fields and 100 000 agents.
  3  2048x2048 map, 128 flow fields, 200 000 agents in total, split over the N GPUs by
     destination / agent slab (strong scaling); runs on one GPU too.
  4  config 2 + 10 000 dynamic obstacles, 1 % moved per tick, incremental field repair.

`--gpus N` with N > 1 started WITHOUT torchrun re-launches itself under `torch.distributed.run --nproc-per-node N`
(one rank per GPU; it refuses, rc 3, when the node shows fewer than N devices).  N ranks split ONE world of the
configuration's size (strong scaling: BASELINE.json's metric is "1024^2 map, 100k agents, 1/2/4/8 GPUs") unless
`--scaling weak`; the weak-scaled job is run behind it and reported in `weak_scaling`.

Prints ONE JSON line on rank 0, numbers only (what the keys mean: profiles/README.md, "the bench line"), shorter than
6 KB with the summary of every regime LAST (`summary`), because the driver's record keeps the tail of the line.
`value` = agents x steps / wall time of the K timed steps (barrier + synchronize on both sides, max over ranks).
"""
import argparse
import hashlib
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

CONFIGS = {      # map side in chunks, flow fields, agents, obstacles, shared map
    0: dict(map=4, fields=1, agents=1_000, obstacles=0, shared=False),
    1: dict(map=16, fields=16, agents=50_000, obstacles=0, shared=False),
    2: dict(map=16, fields=64, agents=100_000, obstacles=0, shared=False),
    3: dict(map=32, fields=128, agents=200_000, obstacles=0, shared=True),
    4: dict(map=16, fields=64, agents=100_000, obstacles=10_000, shared=False),
}


def _strip_comments(text):
    """C/C++ source without comments and with runs of whitespace collapsed (string literals kept)."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == '"' or c == "'":
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c)
            i += 1
    return " ".join("".join(out).split())


def csrc_files(d=None):
    d = d or os.path.join(ROOT, "permafrost-engine_amd", "csrc")
    return sorted(f for f in os.listdir(d) if f.endswith((".hip", ".h")))


def csrc_sha(d=None, files=None):
    """Identity of the kernel CODE (comments and whitespace do not count): profiles/*.json measured on
    another tree are stale.  `files`: the sources a stamp covers (it lists them: every source of the tree it
    was measured on) -- a translation unit ADDED since cannot change the kernels of the others and leaves
    the stamp valid; a change to any covered file, headers included, or its removal, does not."""
    d = d or os.path.join(ROOT, "permafrost-engine_amd", "csrc")
    h = hashlib.sha1()
    for f in sorted(files) if files is not None else csrc_files(d):
        if f.endswith((".hip", ".h")):
            h.update(f.encode())
            try:
                text = open(os.path.join(d, f), encoding="utf-8", errors="replace").read()
            except OSError:
                return "missing:" + f
            h.update(_strip_comments(text).encode())
    return h.hexdigest()[:12]


def stamp_kernels(stamp):
    """The kernel names a profiles/*.json stamp holds numbers for (template arguments dropped)."""
    ks = stamp.get("kernels") or stamp.get("per_kernel_fetch_bytes_raw") or {}
    return sorted({k.split("<")[0] for k in ks})


def stamp_units(files, kernels, d=None):
    """The sources out of `files` that can change the measured `kernels`: every header, every translation
    unit that defines or names one of them, and every unit that defines no kernel at all (host code that
    schedules the launches).  A unit whose own kernels were not measured and that names no measured kernel
    (state_kernels.hip for the tick's counters) cannot: each .hip is compiled by itself."""
    d = d or os.path.join(ROOT, "permafrost-engine_amd", "csrc")
    if not kernels:
        return sorted(files)
    import re
    names = re.compile(r"\b(" + "|".join(re.escape(k) for k in kernels) + r")\b")
    out = []
    for f in sorted(files):
        if f.endswith(".hip"):
            try:
                text = _strip_comments(open(os.path.join(d, f), encoding="utf-8", errors="replace").read())
            except OSError:
                out.append(f)                  # (a covered file that is gone: csrc_sha says so)
                continue
            if "__global__" in text and not names.search(text):
                continue
        out.append(f)
    return out


def stamp_is_current(stamp):
    """A profiles/*.json stamp {csrc_sha, files[, covers]} against this tree.  `covers` (newer stamps): the
    subset of `files` stamp_units() found able to change the measured kernels -- the sha is over those."""
    files = stamp.get("covers") or stamp.get("files")
    return bool(stamp.get("csrc_sha")) and stamp.get("csrc_sha") == csrc_sha(files=files)


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


def state_pass_probe(chunk_w, k_fields, n_agents, timeout=240):
    """tests/tools/bench_state_pass.py in a subprocess: its JSON line, or what went wrong."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "tools", "bench_state_pass.py"), "--chunks", str(chunk_w),
                            "--flocks", str(k_fields), "--agents", str(n_agents)],
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, cwd=ROOT)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return {"error": "exit %d: %s" % (r.returncode, (r.stderr or r.stdout)[-300:])}
        d = json.loads(lines[-1])
        if "error" in d or "resident" not in d:
            return d
        # (the line keeps the numbers: the pass on the velocity pass's resident snapshot, the host-buffer pass beside it)
        return {"identical": d["identical"], "decided_on_device": d["decided_on_device"], "unit_mix": d["unit_mix"],
                "hip_ms_per_tick": d["resident"]["hip_ms_per_tick"], "hip_ms_parts": d["resident"]["hip_ms_parts"],
                "host_buffers_ms_per_tick": d["host_buffers"]["hip_ms_per_tick"], "host_threads": d["host_threads"],
                "cpu_ms_per_tick_1core": d["cpu_ms_per_tick_1core"], "to_arrived": d["to_arrived"], "to_waiting": d["to_waiting"]}
    except Exception as exc:
        return {"error": repr(exc)}


def cpu_baseline(chunk_w, k_fields, n_agents, hz, whole=False, budget_field_s=10.0, budget_agent_s=10.0, dropin=True):
    """The reference's own code (oracle/_ref) timed on this box's host cores, on a bounded sample of
    the SAME workload (whole=True: all of it, config 0).  Reported, never the target."""
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
        # (i) chunk fields: a random sample of the very request list the GPU builds every tick
        liid = synth.from_chunks(nav.plane(pfref.PLANE_LOCAL_ISLANDS))
        cols = synth.planner_requests(grid, dests) or synth.whole_map_requests(grid, dests, liid)
        n_all = len(cols["type"])
        reqs_all = np.zeros(n_all, pfref.FIELD_REQ_DTYPE)
        for k in synth.REQ_FIELDS:
            if k in reqs_all.dtype.names:
                reqs_all[k] = cols[k]
        rng = np.random.RandomState(3)
        probe = reqs_all[rng.choice(n_all, size=min(n_all, 256), replace=False)]
        t_f1 = nav.field_bench(probe, reps=1, nthreads=1)
        per_field = t_f1 / len(probe)
        cells_per_s_1 = 4096 / per_field
        if whole:
            reqs, reps = reqs_all, 1
        else:
            m = int(min(n_all, max(cores * 16, budget_field_s / per_field)))
            reqs, reps = reqs_all[rng.choice(n_all, size=m, replace=False)], 1
        t_f = nav.field_bench(reqs, reps=reps, nthreads=cores)
        cells_per_s = len(reqs) * reps * 4096 / t_f
        # (ii) velocity step: the full snapshot loaded, a slab of it stepped
        ag = synth.agents(grid, n_agents, k_fields, seed=7, hz=hz)
        targets = synth.cell_centre(chunk_w, chunk_w, dests[:, 0], dests[:, 1])
        dest_ids = []
        for f in range(k_fields):
            ok, did = nav.request_path(ag["pos"][f % n_agents], targets[f], clear_cache=(f == 0))
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
        m = n_agents if whole else int(min(n_agents, max(cores * 8, budget_agent_s / per_agent)))
        t_a, _ = mv.bench(vdes, reps=1, nthreads=cores, begin=0, end=m)
        # ---- the drop-in as the engine sees it: the SAME velocity half of the reference's movement tick, all
        # work items, through the binding's WORK_TYPE_HIP arm (bindings/permafrost/move_hip.c: snapshot tables
        # and work items -> navhip_world -> navhip_agent_step_submit / _wait, host buffers and PCIe included ->
        # s_move_work.out[]) and through its WORK_TYPE_CPU arm (move_velocity_work on `cores` pthreads)
        drop = None
        if dropin:
            try:
                if nav.hip_init():
                    mv.bench_hip(vdes, reps=1, end=n_agents)            # (allocations, first launches)
                    reps = 5
                    r1 = mv.bench_hip(vdes, reps=reps, end=n_agents)    # the binding's host loops on the calling task
                    # ... and forked over worker threads, as the engine forks move_velocity_work over its tasks
                    host_threads = max(1, min(8, cores))
                    mv.hip_threads(host_threads)
                    mv.bench_hip(vdes, reps=1, end=n_agents)
                    r = min((mv.bench_hip(vdes, reps=reps, end=n_agents) for _ in range(2)), key=lambda x: x[0] if x else 1e9)
                    mv.hip_threads(1)
                    t_cpu_all = t_a if m == n_agents else mv.bench(vdes, reps=1, nthreads=cores, begin=0, end=n_agents)[0]
                    if r is not None:
                        dt, parts = r
                        # (what the keys mean: profiles/README.md, "the bench line")
                        drop = {"work_items": n_agents, "host_threads": host_threads, "cores": cores,
                                "hip_ms_per_tick": dt / reps * 1e3, "cpu_ms_per_tick": t_cpu_all * 1e3,
                                "speedup": t_cpu_all / (dt / reps),
                                "hip_ms_fill": parts["fill"] / reps * 1e3,
                                "hip_ms_submit_to_wait": parts["device"] / reps * 1e3,
                                "hip_ms_scatter": parts["scatter"] / reps * 1e3,
                                "host_share": (parts["fill"] + parts["scatter"]) / dt,
                                "agent_steps_per_s_hip": n_agents * reps / dt}
                        if r1 is not None:
                            drop["one_host_thread"] = {
                                "hip_ms_per_tick": r1[0] / reps * 1e3, "hip_ms_fill": r1[1]["fill"] / reps * 1e3,
                                "hip_ms_submit_to_wait": r1[1]["device"] / reps * 1e3,
                                "hip_ms_scatter": r1[1]["scatter"] / reps * 1e3}
                pfref.RefNav.hip_shutdown()
            except Exception as exc:
                drop = {"error": repr(exc)}
            # ---- the STATE half of the same tick through the binding (heading gate, state update, settle pass, flag arms:
            # csrc/state_kernels.hip, written after the round's last GPU session).  In a process of its own with a time
            # limit: whatever happens in there, this line is printed.
            if isinstance(drop, dict):
                drop["state_pass"] = state_pass_probe(chunk_w, k_fields, n_agents)
        pfref.RefMove.unload()
        # (iii) the flow sampling of the velocity step (a13: N_DesiredPointSeekVelocity per agent, serial on the nav
        # task -- compute_desired_velocity, movement.c:4166), which the slab timing above is given.  The reference's
        # field cache holds 2 048 chunk fields: eight destinations' worth of ITS OWN fields are put into it under its
        # ids and mappings (what n_request_path leaves behind) and the agents of those eight flocks sampled.
        sampling = None
        try:
            if "dest" in cols and k_fields >= 1:
                nd = min(8, k_fields)
                sel = np.flatnonzero(cols["dest"] < nd)
                if 0 < len(sel) <= 2048:
                    dirs = nav.field_update_many(reqs_all[sel], nthreads=cores)
                    ids8 = np.array([nav.dest_id(t) for t in targets[:nd]], np.uint32)
                    nav.cache_clear()
                    nav.cache_put_fields(reqs_all[sel], ids8[cols["dest"][sel]], dirs)
                    ag8 = np.flatnonzero(ag["flock"] < nd)[:20000]
                    t0 = time.perf_counter()
                    nav.desired_velocities(ids8[ag["flock"][ag8]], ag["pos"][ag8], targets[ag["flock"][ag8]])
                    t_s = (time.perf_counter() - t0) / max(1, len(ag8))
                    sampling = {"us_per_agent_1core": t_s * 1e6, "agents_sampled": int(len(ag8)), "destinations": int(nd),
                                "agent_steps_per_s_with_serial_sampling": 1.0 / (t_a / m + t_s)}
        except Exception as exc:
            sampling = {"error": repr(exc)}
        return {
            "dropin": drop,
            "flow_sampling": sampling,
            "value": m / t_a, "unit": "agent-steps/s", "cores": cores, "kind": "reference",
            "sample": "move_velocity_work on %d of %d agents, %d threads; N_FlowFieldUpdate on %d of %d chunk fields"
                      % (m, n_agents, cores, len(reqs), n_all),
            "skips": "a13 sampling timed apart (flow_sampling); a23 position accept",
            "flow_field_cells_per_s": cells_per_s, "flow_field_cells_per_s_1core": cells_per_s_1,
            "agent_steps_per_s_1core": 1.0 / per_agent,
            "os_cpu_count": os.cpu_count() or 0,
            "cpu_work_s": {"fields": per_field * len(reqs) * reps, "agents": per_agent * m},
        }
    except Exception as exc:                      # the baseline is informational
        return {"value": None, "unit": "agent-steps/s", "cores": os.cpu_count(), "kind": "reference",
                "sample": "unavailable: %r" % (exc,)}


def r4(x, nd=4):
    """Numbers of the line rounded to `nd` significant digits (the driver's record keeps only the tail of the line)."""
    if isinstance(x, dict):
        return {k: r4(v, nd) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [r4(v, nd) for v in x]
    if isinstance(x, float):
        return float("%.*g" % (nd, x))
    return x


def status_histogram(T):
    import numpy as np
    from permafrost_engine_amd import navhip
    st = T.status[T.a0:T.a1].cpu().numpy()
    lists = T.ctx.last_step_lists()
    n = max(1, len(st))
    import ctypes as C
    att = (C.c_ulonglong * 9)()
    cp = None
    if navhip.lib().navhip_debug_cp_attempts(att, 1) == 0:
        cp = {"returned_in_attempt_2_to_8plus": [int(att[i]) for i in range(1, 8)], "gave_up": int(att[0]),
              "attempts_of_retried": int(att[8])}
    return {
        "moved": float((st & navhip.ST_MOVED).astype(bool).mean()),
        "field_miss": float((st & navhip.ST_FIELD_MISS).astype(bool).mean()),
        "field_none": float((st & navhip.ST_FIELD_NONE).astype(bool).mean()),
        "unsupported": float((st & navhip.ST_UNSUPPORTED).astype(bool).mean()),
        "cp_rows_1-2_3-4_5-8_9-16_nbrs": [x / n for x in lists[:4]],
        "cp_wave_17-64_nbrs": lists[4] / n, "whole_step_on_wave": lists[5] / n,
        "cp_retries": cp,
    }


def profiled_ticks(T, n):
    """n ticks with every kernel group of the agent step back to back on ONE stream (no prefetch, no
    overlap with the field builds), the library's own HIP events between the groups.  Returns the
    mean milliseconds per group (first tick dropped when n > 1)."""
    import numpy as np
    from permafrost_engine_amd import navhip
    T.ctx.set_profiling(True)
    keep = (T.overlap, T.record, T.mark_every, T.ev, T.tick_ev, T.pipeline_fields, getattr(T, "fev", []))
    T.overlap, T.record, T.mark_every, T.ev, T.tick_ev, T.pipeline_fields, T.fev = False, True, 1, [], [], False, []
    rows = []
    for _ in range(n):
        T.step()
        T.sync()
        rows.append(T.ctx.last_step_ms())
    serial = T.phase_ms()
    T.overlap, T.record, T.mark_every, T.ev, T.tick_ev, T.pipeline_fields, T.fev = keep
    T.ctx.set_profiling(False)
    rows = rows[1:] if len(rows) > 1 else rows
    g = {name: float(np.mean([x[i] for x in rows])) for i, name in enumerate(navhip.STEP_PHASES)}
    g["fields"] = serial.get("fields", 0.0)
    return g


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def emulated():
    """The library under test is the host-emulator build (tests/hostsim; NAVHIP_LIB names it): test infrastructure,
    the only case in which this script runs without a GPU."""
    return os.path.basename(os.environ.get("NAVHIP_LIB", "")) == "_navhip_emu.so"


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks ourselves, one per GPU, exactly as the
    driver's multi-GPU command does, and hand back their exit code.  Refuses when the node shows fewer than N devices."""
    import subprocess
    if not emulated():
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:
            sys.stderr.write("bench.py --gpus %d: only %d GPU(s) visible on this node; not faking ranks\n" % (n, have))
            return 3
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, cwd=ROOT)


def run_ticks(T, pdist, torch, warmup, steps, early=None):
    import numpy as np
    from permafrost_engine_amd.tick import tcuda
    for _ in range(max(0, warmup - 3) if early is not None else warmup):
        T.step()
    T.sync()
    if early is not None:
        early.update(profiled_ticks(T, min(3, warmup)))
    pdist.barrier()
    tcuda.synchronize()
    T.record = True
    T.ev, T.tick_ev, T.fev, T._tick_rec = [], [], [], 0
    f_sum0, f_n0 = T.field_build_times()
    t0 = time.perf_counter()
    for _ in range(steps):
        T.step()
    T.host_enqueue_ms = (time.perf_counter() - t0) / max(1, steps) * 1e3      # (host time to enqueue one tick)
    if T._tick_rec % T.tick_every == 0:           # close the last window
        e = tcuda.Event(enable_timing=True)
        e.record(T.stream)
        T.tick_ev.append(e)
    T.sync()
    tcuda.synchronize()
    pdist.barrier()
    dt = time.perf_counter() - t0
    T.record = False
    dt = pdist.max_over_ranks(dt, T.dev)
    # the field builds as they ran INSIDE the timed ticks (events on the field stream, every fourth tick)
    f_sum1, f_n1 = T.field_build_times()
    T.fields_in_tick_ms = (f_sum1 - f_sum0) / (f_n1 - f_n0) if f_n1 > f_n0 else None
    T.fields_in_tick_samples = f_n1 - f_n0
    ticks = np.array(T.tick_ms()) if steps > 1 and len(T.tick_ev) > 1 else np.array([dt * 1e3 / max(1, steps)])
    return dt, ticks


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--map", type=int, default=None, help="override: map side in chunks (16 = 1024x1024 cells)")
    ap.add_argument("--fields", type=int, default=None, help="override: whole-map flow fields (per GPU unless config 3)")
    ap.add_argument("--agents", type=int, default=None, help="override: agents (per GPU unless config 3)")
    ap.add_argument("--obstacles", type=int, default=None,
                    help="override: dynamic obstacles, 1%% moved per tick, incremental field repair")
    ap.add_argument("--crowded", action="store_true", help="the main run uses the crowded world")
    ap.add_argument("--no-crowded", action="store_true", help="skip the crowded-world secondary measurement")
    ap.add_argument("--tile-exchange", choices=("auto", "all"), default="auto",
                    help="multi-GPU: auto = only baked tiles that another rank's agents sample travel "
                         "(none in this workload: flocks are rank aligned); all = all-gather every tile "
                         "every tick (any agent may sample any field)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=("weak", "strong"), default=None,
                    help="--gpus N > 1: weak = every GPU brings its own region, fields and agents (default for configs "
                         "0-2, 4); strong = ONE world of the configuration's size, its destinations and its agents "
                         "split over the N ranks (requests by destination, uid slabs: movement.c:3759-3762; default "
                         "for config 3)")
    ap.add_argument("--no-weak", action="store_true",
                    help="--gpus N > 1 with the default (strong) scaling: skip the weak-scaled job run behind it")
    ap.add_argument("--no-los", action="store_true",
                    help="has_dest_los = 0 for every agent instead of the per-tick device lookup in the planner's LOS fields")
    ap.add_argument("--no-sustained", action="store_true",
                    help="skip the secondary 100-tick measurement that a run with --steps < 100 adds")
    ap.add_argument("--no-dropin", action="store_true", help="skip timing the reference's own movement tick with the binding")
    ap.add_argument("--no-pipeline-fields", action="store_true",
                    help="build a tick's fields inside that tick, in front of its agent step (default: the fields "
                         "of tick t+1 are built during tick t, beside the agent step; same work, same results)")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(relaunch_under_torchrun(args.gpus))
    cfg = dict(CONFIGS[args.config])
    for k in ("map", "fields", "agents", "obstacles"):
        if getattr(args, k) is not None:
            cfg[k] = getattr(args, k)

    import numpy as np
    import torch
    from permafrost_engine_amd import dist as pdist
    import __graft_entry__ as ge
    ge.build_navhip()
    from permafrost_engine_amd import tick

    rank, world, local = pdist.init()
    if world != args.gpus and world > 1:
        args.gpus = world
    if emulated():
        local = 0
    elif not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: libnavhip has no CPU fallback")
    else:
        torch.cuda.set_device(local)

    # BASELINE.json's metric is ONE world ("1024^2 map, 100k agents, 1/2/4/8 GPUs"; the reference forks one snapshot
    # over its tasks, movement.c:3746-3774): N > 1 ranks split it (strong) unless --scaling weak; the weak-scaled job
    # (every GPU brings its own region, fields and agents) is then run behind it and reported in `weak_scaling`
    shared = (cfg["shared"] or world > 1) if args.scaling is None else args.scaling == "strong"
    f_rank = cfg["fields"] // world if shared else cfg["fields"]
    a_rank = cfg["agents"] // world if shared else cfg["agents"]
    if shared and (cfg["fields"] % world or cfg["agents"] % world):
        raise SystemExit("bench.py --scaling strong: %d ranks do not divide %d fields / %d agents"
                         % (world, cfg["fields"], cfg["agents"]))
    CROWD = 17      # cells: a flock of ~1 600 packed into ~35 x 35 cells -> ~30 neighbours within r = 10

    def make(crowd, share=False, weak=False):
        return tick.NavTick(share_fields=share, chunk_w=cfg["map"], fields_per_rank=cfg["fields"] if weak else f_rank,
                            agents_per_rank=cfg["agents"] if weak else a_rank,
                            rank=rank, world=world, device=local, obstacles=cfg["obstacles"],
                            obstacle_ticks=args.warmup + args.steps + 16, tile_exchange=args.tile_exchange,
                            shared_map=shared and not weak, crowd_cells=CROWD if crowd else 0,
                            pipeline_fields=not args.no_pipeline_fields, los=not args.no_los, flow_velocities=True,
                            time_fields=True)

    T = make(args.crowded)
    fields_ahead, tick_every = T.pipeline_fields, T.tick_every
    request_source, los_source, velocity_source = T.request_source, T.los_source, T.velocity_source
    early = {}
    dt, ticks = run_ticks(T, pdist, torch, args.warmup, args.steps, early)
    phases = T.phase_ms()
    hist = status_histogram(T)
    host_enqueue_ms, tick_driver = T.host_enqueue_ms, getattr(T, "tick_driver", "python (tick.py)")
    fields_in_tick_ms, fields_in_tick_samples = T.fields_in_tick_ms, T.fields_in_tick_samples

    # per-kernel-group durations at the END of the run (the world has crowded by then); the same
    # split for the last warm-up ticks is in `early`
    prof_first = T.tick_no + 1
    groups = profiled_ticks(T, 6)
    T_ticks_done = T.tick_no
    prof_ticks = "ticks %d-%d" % (prof_first + 1, T_ticks_done)

    agents_total = T.N
    cells_total = T.n_req_total * 4096
    ms_per_step = dt / args.steps * 1e3
    value = agents_total * args.steps / dt

    # ---- roofline (HBM bound; algorithmic bytes per launch) ---------------
    f_bytes = T.n_req_local * 4096 * BYTES_PER_CELL
    a_bytes = (T.a1 - T.a0) * BYTES_PER_AGENT_STEP + T.map_cells * PLANE_BYTES_PER_MAP_CELL
    a_ms = sum(groups[k] for k in ("sp_build", "agent_nbr", "cohesion", "agent_finish"))
    f_ms = groups["fields"]
    sha = csrc_sha()
    measured, calib = {}, {}
    try:
        measured = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        pass
    calib_file = None
    for name in ("archive/r03_valu_calib.json", "archive/r02_valu_calib.json"):      # (a hardware constant: the newest run kept)
        try:
            calib = json.load(open(os.path.join(ROOT, "profiles", name)))
            calib_file = "profiles/" + name
            break
        except Exception:
            pass
    stale = not stamp_is_current(measured)
    added_since = sorted(set(csrc_files()) - set(measured.get("files") or csrc_files()))   # (units a stamp does not cover)
    sq = {}
    try:
        sqj = json.load(open(os.path.join(ROOT, "profiles", "sq_counters.json")))
        if stamp_is_current(sqj):
            sq = sqj["kernels"]
    except Exception:
        pass
    cyc = None
    try:
        cyc = float(calib["results"]["waves_per_simd_8"]["v_fma_f32"])
    except Exception:
        pass

    late_window = T_ticks_done >= 60

    def valu(prefixes, measured_ms):
        ks = [k for k in sq if k.startswith(prefixes)]
        if not ks or measured_ms <= 0 or cyc is None:
            return None
        # (the counters were taken on a 100-tick run: per kernel the last dispatches -- the crowded world
        # -- and dispatches 6-11; a short run is priced with the early ones)
        key = "SQ_INSTS_VALU" if late_window else "SQ_INSTS_VALU_early_ticks"
        insts = sum((sq[k].get(key) or sq[k]["SQ_INSTS_VALU"]) for k in ks)
        floor_ms = insts / 1024 * cyc / 2.4e9 * 1e3
        return {"valu_insts_per_launch": insts, "issue_floor_ms": floor_ms, "cycles_per_inst": cyc,
                "frac_of_issue_peak": floor_ms / measured_ms,
                "counters_of": "ticks 100-110" if key == "SQ_INSTS_VALU" else "ticks 27-32"}

    def roof(which, g=None, when=None, in_tick_ms=None):
        """SURVEY.md section 8(d)'s convention -- algorithmic bytes per launch / the launch's duration / 8 TB/s -- for the
        field kernel (ONE kernel) or the agent step (a group of kernels: its dominant kernel is named with its own time).
        The duration is the one measured INSIDE the timed ticks where there is one (the field builds: events on their
        stream), else the kernel alone on one stream (profiled ticks); hbm_frac_measured prices the bytes the PMC counters
        saw instead of the algorithmic ones."""
        g = g or groups
        when = when or prof_ticks
        a_ms_ = sum(g[k] for k in ("sp_build", "agent_nbr", "cohesion", "agent_finish"))
        alone, by = (a_ms_, a_bytes) if which == "agents" else (g["fields"], f_bytes)
        ms = in_tick_ms if in_tick_ms else alone
        gbs = by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        # (counters and durations of one line from the same window of the world: the profiled ticks behind a short run
        # are ticks 27-32, behind a 100-tick run ticks 106-111 -- traffic.json holds both)
        traffic = None if stale else measured.get(which + ("_bytes_per_launch" if late_window else "_bytes_per_launch_early"))
        out = {
            "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
            "traffic": traffic,
            "hbm_frac_measured": (traffic / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic and ms > 0 else None,
            "traffic_stale": bool(stale and measured), "traffic_csrc_sha": measured.get("csrc_sha"),
            "avg_launch_ms": ms, "algorithmic_bytes_per_launch": by,
        }
        if which == "fields":
            out["kernel"] = "k_field_bfs"
            out["launch_timing"] = ("HIP events on the field stream around the builds of every fourth TIMED tick "
                                    "(%d samples): the kernel as it runs beside the agent step" % fields_in_tick_samples) \
                if in_tick_ms else "HIP events, the kernel alone on one stream, " + when
            out["avg_launch_ms_alone"] = alone
            out["frac_alone"] = by / (alone * 1e-3) / 1e9 / HBM_PEAK_GBS if alone > 0 else None
            out["valu_issue"] = valu(("k_field_",), alone)
        else:
            out["kernel"] = "agent step (a GROUP: k_sp_* + k_agent_nbr + k_cohesion + k_agent_mid + k_cp_*)"
            out["dominant_kernel"] = {"name": "k_cohesion", "ms_alone": g["cohesion"]}
            # the groups run back to back on ONE stream with events between them: a SERIAL time that may exceed
            # ms_per_step, where k_cohesion and the ClearPath kernels overlap on side streams
            out["launch_timing"] = "HIP events, groups serial on one stream (may exceed ms_per_step), " + when
            out["kernels_ms"] = {k: g[k] for k in ("sp_build", "agent_nbr", "cohesion", "coh_regroup", "agent_finish")}
            out["valu_issue"] = valu(("k_agent_", "k_cp_", "k_coh", "k_sp_"), alone)
        return out

    # `roofline` is ONE kernel: k_field_bfs -- the kernel with the most algorithmic bytes per launch (268 MB against the
    # agent step's 14 MB) and, inside the tick, the longest single launch; the agent step, a group of kernels, follows
    # as `roofline_secondary`.  (A world without field requests: the agent step alone.)
    dom = "fields" if f_bytes > 0 and f_ms > 0 else "agents"
    other = "fields" if dom == "agents" else "agents"
    in_tick = {"fields": fields_in_tick_ms, "agents": None}

    # ---- the sustained regime: a run shorter than 100 ticks (the driver's --steps 20) times the friendliest
    # window of the world -- the flocks have not converged yet.  A second, fresh world is then run for 100 ticks
    # behind it, so that the line always carries ticks 5 / 50 / 100, the 100-tick mean and the roofline of the
    # late ticks next to the headline.
    sustained = None
    if rank == 0 and world == 1 and args.steps < 100 and not args.no_sustained and not args.crowded:
        T.close()
        T = None
        Ts = make(False)
        sdt, sticks = run_ticks(Ts, pdist, torch, 5, 100)
        s_first = Ts.tick_no + 1
        sgroups = profiled_ticks(Ts, 6)

        def s_at(i):
            w = (i - 1) // Ts.tick_every
            return float(sticks[w]) if len(sticks) > w else None
        sustained = {"ticks": 100, "warmup": 5,
                     "ms_per_step": sdt / 100 * 1e3, "ms_per_step_median": float(np.median(sticks)),
                     "agent_steps_per_s": Ts.N * 100 / sdt, "ms_tick_5_50_100": [s_at(5), s_at(50), s_at(100)],
                     "cp_wave_17-64_nbrs": status_histogram(Ts)["cp_wave_17-64_nbrs"], "roofline": None}
        sr = roof("agents", sgroups, "ticks %d-%d" % (s_first + 1, Ts.tick_no))
        sustained["roofline"] = {k: sr[k] for k in ("achieved", "frac", "avg_launch_ms", "kernels_ms")}
        sustained["fields_in_tick_ms"] = Ts.fields_in_tick_ms
        Ts.close()

    # ---- the same tick with the reference's field SHARING: its cache is keyed by N_FlowFieldID (chunk + target,
    # not destination), so a tick after a wholesale invalidation rebuilds every DISTINCT chunk field once and maps
    # many (destination, chunk) pairs to it.  The headline rebuilds every request; this block says what the
    # library's slot table buys a host that shares like the reference does.
    shared_fields = None
    if rank == 0 and world == 1 and not args.no_crowded and not args.crowded and args.config in (1, 2, 3):
        if T is not None:
            T.close()
        T = None
        Tq = make(False, share=True)
        qdt, qticks = run_ticks(Tq, pdist, torch, args.warmup, args.steps)
        shared_fields = {"chunk_field_requests_served_per_tick": Tq.n_requests_served,
                         "distinct_chunk_fields_built_per_tick": Tq.n_req_local,
                         "ms_per_step": qdt / args.steps * 1e3, "ms_per_step_median": float(np.median(qticks)),
                         "agent_steps_per_s": Tq.N * args.steps / qdt,
                         "flow_field_cells_built_per_s": Tq.n_req_local * 4096 * args.steps / qdt}
        Tq.close()

    crowded = None
    if rank == 0 and world == 1 and not args.no_crowded and not args.crowded and args.config in (1, 2):
        if T is not None:
            T.close()
        T = None
        Tc = make(True)
        cdt, cticks = run_ticks(Tc, pdist, torch, 3, 40)
        chist = status_histogram(Tc)
        crowded = {"ticks": 40, "warmup": 3, "agent_steps_per_s": Tc.N * 40 / cdt, "ms_per_step": cdt / 40 * 1e3,
                   "ms_per_step_median": float(np.median(cticks)), "cp_wave_17-64_nbrs": chist["cp_wave_17-64_nbrs"],
                   "cp_retries": chist["cp_retries"]}
        Tc.close()

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg["map"], cfg["fields"], cfg["agents"], 20, whole=(args.config == 0),
                           dropin=not args.no_dropin)

    # ---- N > 1, strong by default: the weak-scaled job of the same configuration behind it --------------------
    weak = None
    if world > 1 and shared and args.scaling is None and not args.no_weak and cfg["map"] * tick.region_grid(world)[1] <= 64:
        T.close()
        T = None
        Tw = make(args.crowded, weak=True)
        wdt, wticks = run_ticks(Tw, pdist, torch, args.warmup, args.steps)
        weak = {"value": Tw.N * args.steps / wdt, "ms_per_step": wdt / args.steps * 1e3, "agents": Tw.N,
                "chunk_fields": Tw.n_req_total, "map_chunks": [Tw.Wt, Tw.H]}
        Tw.close()

    if rank == 0:
        def at(i):          # (ticks: one value per window of T.tick_every ticks)
            w = (i - 1) // tick_every
            return float(ticks[w]) if len(ticks) > w else None
        dims = T_dims(cfg, world, shared)
        backend = torch.distributed.get_backend() if torch.distributed.is_initialized() else "none"
        drop = (cpu or {}).pop("dropin", None) if isinstance(cpu, dict) else None
        state_pass = drop.pop("state_pass", None) if isinstance(drop, dict) else None
        sampling = (cpu or {}).pop("flow_sampling", None) if isinstance(cpu, dict) else None
        t5 = [at(5), at(50), at(100)] if args.steps >= 100 or sustained is None else sustained["ms_tick_5_50_100"]
        # What every key means, and the prose that used to ride in the line: profiles/README.md ("the bench line").
        # The driver's record keeps the TAIL of the line: the summary of every regime comes last.
        line = {
            "metric": "agent-steps/sec (+ flow-field cells/sec), every chunk field rebuilt + every agent stepped per tick",
            "value": value, "unit": "agent-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True,
            "scaling": None if world == 1 else ("strong" if shared else "weak"), "vs_baseline": None,
            "dtype": "u64-bitmask/u8 fields, f32 agents (f64 exp)", "data": "synthetic",
            "config": {"workload": "BASELINE configs[%d]%s: %dx%d cells, %d flow fields%s = %d chunk fields, %d agents%s"
                                   % (args.config, " crowded" if args.crowded else "", dims[0], dims[1], cfg["fields"],
                                      "" if shared or world == 1 else "/GPU", cells_total // 4096, agents_total,
                                      ", %d obstacles" % cfg["obstacles"] if cfg["obstacles"] else ""),
                       "baseline_config": args.config, "map_chunks": cfg["map"], "flow_fields": cfg["fields"],
                       "agents": cfg["agents"], "hz": 20, "dynamic_obstacles": cfg["obstacles"],
                       "parallelism": "requests by destination + uid slabs x%d, 1 all-gather (16 B/agent) per tick"
                                      % world,
                       "ranks": world, "rccl_ranks": pdist.comm_ranks(), "backend": backend,
                       "tile_exchange": args.tile_exchange,
                       "requests": "reference planner fixture" if "fixture" in request_source else "numpy stand-in",
                       "has_dest_los": "device lookup" if "device lookup" in los_source else "0",
                       "initial_velocities": "flow aligned" if "flow" in velocity_source else "N(0,0.35)",
                       "fields_ahead": bool(fields_ahead), "tick_driver": tick_driver,
                       "library": ("NAVHIP_LIB override: " + os.path.basename(os.environ["NAVHIP_LIB"]))
                                  if os.environ.get("NAVHIP_LIB") else "libnavhip.so (in-tree)"},
            "roofline": roof(dom, in_tick_ms=in_tick[dom]),
            "cpu_baseline": cpu,
            "roofline_secondary": {k: v for k, v in roof(other, in_tick_ms=in_tick[other]).items()
                                   if k in ("achieved", "frac", "traffic", "hbm_frac_measured", "kernel", "dominant_kernel",
                                            "avg_launch_ms", "algorithmic_bytes_per_launch", "kernels_ms", "launch_timing",
                                            "valu_issue")},
            "csrc_sha": sha,
            "kernel_groups_ms_serial": {"after_warmup": early, "after_timed_region": groups} if args.steps >= 100 else
                                       {"after_timed_region": groups},
            "phase_ms_overlapped": phases,
            "status": hist,
            "flow_sampling_cpu": sampling,
            "shared_fields": shared_fields,
            "weak_scaling": weak,
            "sustained_100": sustained,
            "crowded_world": crowded,
            "dropin": drop,
            "state_pass": state_pass,
            ("flow_field_cells_kept_valid_per_s" if cfg["obstacles"] else "flow_field_cells_per_s"):
                cells_total * args.steps / dt,
            "summary": {"ms_per_step": ms_per_step, "ms_per_step_median": float(np.median(ticks)),
                        "ms_tick_5_50_100": t5,
                        "ms_tick_5_50_100_of": "this run" if args.steps >= 100 or sustained is None else "sustained_100",
                        "sustained_100_ms": sustained["ms_per_step"] if sustained else None,
                        # "run 100 ticks; report median tick"
                        "survey_8d_median_ms": (sustained["ms_per_step_median"] if sustained else
                                                float(np.median(ticks)) if args.steps >= 100 else None),
                        "crowded_ms": crowded["ms_per_step"] if crowded else None,
                        "dropin_ms": drop.get("hip_ms_per_tick") if isinstance(drop, dict) else None,
                        "state_pass_ms": state_pass.get("hip_ms_per_tick") if isinstance(state_pass, dict) else None,
                        "host_enqueue_ms": host_enqueue_ms,
                        "roofline_frac": roof(dom, in_tick_ms=in_tick[dom])["frac"], "ranks": world},
        }
        print(json.dumps(r4(line), separators=(",", ":")), flush=True)
    if T is not None:
        T.close()
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


def T_dims(cfg, world, shared):
    from permafrost_engine_amd import tick
    rows, cols = (1, 1) if shared else tick.region_grid(world)
    return cfg["map"] * cols * 64, cfg["map"] * rows * 64, cfg["map"] * cols, cfg["map"] * rows


if __name__ == "__main__":
    main()
