#!/usr/bin/env python3
"""The STATE half of the reference's movement tick (fork_join_state_updates, movement.c:4196) at a benchmark
configuration through the binding, as the tick runs it: the velocity pass first (move_hip_velocity_work ->
navhip_agent_step_submit / _wait), then bindings/permafrost/move_hip.c's move_hip_state_work on what that pass left on
the device (navhip_state_pass_resident: heading gate -> state update -> flag / counter / target arms,
csrc/state_kernels.hip; the settle pass for units of arriving flocks) -- against the reference's own
entity_compute_update per unit on one core, GIVEN THE SAME VELOCITIES (the device's), every unit's next state and flags
compared.  The world holds every arm of the switch: MOVING / ARRIVED / WAITING, formation members on the move and
ARRIVING_TO_CELL, TURNING, ENTER_ENTITY_RANGE, SURROUND_ENTITY, and two flocks with an active arrival zone.
Prints one JSON line (numbers only; what the keys mean: profiles/README.md).  bench.py's cpu_baseline leg runs it in a
process of its own (`state_pass`; it uses the reference build under oracle/, which is why it lives with the test tools).

    python tests/tools/bench_state_pass.py [--chunks 16] [--flocks 64] [--agents 100000] [--reps 3] [--threads 8]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chunks", type=int, default=16)
    ap.add_argument("--flocks", type=int, default=64)
    ap.add_argument("--agents", type=int, default=100000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--threads", type=int, default=8, help="host threads of the binding's fill / scatter loops")
    ap.add_argument("--plain", action="store_true", help="MOVING / ARRIVED / WAITING units only (round 4's world)")
    ap.add_argument("--end", type=int, default=0, help="work items [0, end) only (a slab; the emulator steps ~30 agents a second)")
    ap.add_argument("--arms-scale", type=float, default=1.0, help="multiplies the share of ENTER_ENTITY_RANGE / SURROUND_ENTITY units")
    args = ap.parse_args()
    import numpy as np
    from oracle import pfref
    from permafrost_engine_amd import synth
    from tests import cases
    if not pfref.available():
        print(json.dumps({"error": "oracle/_ref is not present"}))
        return
    W, K, N = args.chunks, args.flocks, args.agents
    grid = synth.cost_grid(W, W, seed=1234)
    nav = pfref.RefNav(synth.to_chunks(grid))
    dests = synth.destinations(grid, K, seed=42)
    ag = synth.agents(grid, N, K, seed=7, hz=20)
    targets = synth.cell_centre(W, W, dests[:, 0], dests[:, 1])
    rng = np.random.RandomState(5)
    pos = ag["pos"].copy()
    state = np.zeros(N, np.int32)
    u = rng.rand(N)
    state[u < 0.15] = 2                       # STATE_ARRIVED: their flock mates arrive next to them
    state[(u >= 0.15) & (u < 0.20)] = 4       # STATE_WAITING
    fstate = np.zeros(N, np.uint8)
    ticks = np.full(N, 40, np.int32)
    prev = np.zeros(N, np.uint8)
    mix = {}
    zones = {}
    if not args.plain:
        state[(u >= 0.20) & (u < 0.24)] = 8   # STATE_ARRIVING_TO_CELL
        state[(u >= 0.24) & (u < 0.28)] = 1   # STATE_MOVING_IN_FORMATION
        state[(u >= 0.28) & (u < 0.31)] = 7   # STATE_TURNING
        # (the two arms with per-unit nav queries on the host -- the reference's own, 30-90 us each -- in the numbers an
        # army has them: a garrison order, a harvest / attack group)
        a6, a5 = 0.31 + 0.01 * args.arms_scale, 0.31 + 0.013 * args.arms_scale
        state[(u >= 0.31) & (u < a6)] = 6     # STATE_ENTER_ENTITY_RANGE
        state[(u >= a6) & (u < a5)] = 5       # STATE_SURROUND_ENTITY
        fstate = ((rng.rand(N) < 0.3) * 1 | (rng.rand(N) < 0.7) * 2 | (rng.rand(N) < 0.7) * 4 | (rng.rand(N) < 0.5) * 8
                  | (rng.rand(N) < 0.5) * 16).astype(np.uint8)
        ticks = rng.choice([1, 2, 3, 40], N).astype(np.int32)
        prev = rng.choice([0, 1], N).astype(np.uint8)
        # two flocks with an active arrival zone round their targets (struct arrival_state), their units round it
        for f, (fill, active_row, num_rows) in ((1, (0.8, 1, 3)), (2, (0.95, 2, 3))):
            if f >= K:
                break
            t = targets[f]
            cell = (int((t[1] + W * 128.0) // 4), int((W * 128.0 - t[0]) // 4))
            zones[f] = cases.arrival_zone_at(grid, cell, 8, rng, fill, active_row, num_rows)
            m = np.flatnonzero(ag["flock"] == f)
            pos[m] = (zones[f]["centre_xz"] + rng.normal(0, 22.0, (len(m), 2))).astype(np.float32)
        pos = np.clip(pos, -W * 128.0 + 14, W * 128.0 - 14).astype(np.float32)
    dest_ids = []
    for f in range(K):
        ok, did = nav.request_path(pos[f % N], targets[f], clear_cache=(f == 0))
        dest_ids.append(did)
    nav.trace()
    mv = pfref.RefMove(nav, pos, ag["vel"], ag["radius"], ag["max_speed"], ag["speed"],
                       np.full(N, pfref.ENTITY_FLAG_MOVABLE, np.uint32), state, ag["flock"], np.zeros(N, np.uint8),
                       targets, np.array(dest_ids, np.uint32), hz=20)
    vdes = rng.normal(0, 1, (N, 2)).astype(np.float32)
    vdes /= np.maximum(np.linalg.norm(vdes, axis=1, keepdims=True), 1e-6)
    vdes[rng.rand(N) < 0.05] = 0

    units = None

    def set_inputs():
        """movestate as the tick finds it (the reference's switch writes into it: reset before every pass)."""
        mv.set_state_aux(fstate, ticks, prev)
        if args.plain:
            return
        mv.set_turning(ent_rot, target_dir)
        mv.set_range_targets(tgt, t_range, t_prev)          # (also sets surround_target_uid: the same field)
        mv.set_surround(tgt, s_tprev, s_nprev)
        if units is not None:
            mv.set_arrival_units(units)

    if not args.plain:
        ang = rng.uniform(-np.pi, np.pi, N)
        off = np.where(rng.rand(N) < 0.5, rng.uniform(-4.5, 4.5, N), rng.uniform(6, 180, N) * rng.choice([-1, 1], N))
        target_dir = pfref.RefMove.dir_quat(np.stack([np.cos(ang), np.sin(ang)], 1))
        ent_rot = pfref.RefMove.dir_quat(np.stack([np.cos(ang + np.deg2rad(off)), np.sin(ang + np.deg2rad(off))], 1))
        # targets of the ENTER_ENTITY_RANGE / SURROUND_ENTITY units: a unit of the same flock (near), or none
        tgt = np.full(N, -1, np.int32)
        order = np.argsort(ag["flock"], kind="stable")
        start = np.searchsorted(ag["flock"][order], np.arange(K + 1))
        for i in np.flatnonzero((state == 6) | (state == 5)):
            if rng.rand() < 0.08:
                continue
            f = ag["flock"][i]
            cand = order[start[f]:start[f + 1]]
            j = cand[rng.randint(len(cand))]
            tgt[i] = j if j != i else -1
        t_range = rng.choice([0.0, 5.0, 20.0, 60.0], N).astype(np.float32)
        t_prev = (pos[np.maximum(tgt, 0)] + rng.normal(0, 4.0, (N, 2))).astype(np.float32)
        s_tprev = pos[np.maximum(tgt, 0)].copy()
        moved = rng.rand(N) < 0.1             # (most surround targets stand still: buildings, resources)
        s_tprev[moved] += rng.normal(0, 3.0, (moved.sum(), 2)).astype(np.float32)
        s_nprev = (pos + rng.normal(0, 6.0, (N, 2))).astype(np.float32)
        if zones:
            sink = pos + rng.normal(0, 12.0, (N, 2)).astype(np.float32)
            for f, z in zones.items():
                m = np.flatnonzero(ag["flock"] == f)
                sink[m] = z["slots_xz"][rng.randint(len(z["slots_xz"]), size=len(m))]
                mv.set_arrival_zone(f, z)
            units = {"substate": rng.randint(0, 4, N).astype(np.uint8), "sink_valid": (rng.rand(N) < 0.7).astype(np.uint8),
                     "sink_xz": sink.astype(np.float32), "order_pos_xz": (pos + rng.normal(0, 3.5, (N, 2))).astype(np.float32),
                     "progress_anchor_xz": (pos + rng.normal(0, 1.4, (N, 2))).astype(np.float32),
                     "progress_anchored": (rng.rand(N) < 0.7).astype(np.uint8), "stuck": rng.randint(0, 14, N).astype(np.int32)}
        mix = {"arriving_to_cell": int((state == 8).sum()), "formation": int(((state <= 1) & (fstate & 1 > 0)).sum()),
               "turning": int((state == 7).sum()), "enter_range": int((state == 6).sum()), "surround": int((state == 5).sum()),
               "arrival_zone_units": int(np.isin(ag["flock"], list(zones)).sum())}
    M = args.end if 0 < args.end < N else N
    out = {"work_items": M, "flocks": K, "chunks": W, "unit_mix": mix}
    try:
        if not nav.hip_init():
            out["error"] = "no device"
        else:
            mv.hip_threads(args.threads)
            # the velocity pass of the tick on the device; the reference's state pass GIVEN those velocities, one core
            set_inputs()                 # (the arrival state of the units feeds the velocity pass too)
            assert mv.bench_hip(vdes, end=M) is not None
            vel, vd = mv.get_out()
            set_inputs()
            t0 = time.perf_counter()
            ref_state, ref_flags = mv.state_update(vel, vd, end=M)
            out["cpu_ms_per_tick_1core"] = (time.perf_counter() - t0) * 1e3
            # (a) the host-buffer pass (every array travels), (b) the pass on the resident snapshot of the velocity pass
            for name, resident in (("host_buffers", False), ("resident", True)):
                mv.hip_resident_state_pass(resident)
                times, best_parts, same, dv = [], None, True, None
                for _ in range(args.reps + 1):
                    if resident:
                        set_inputs()
                        assert mv.bench_hip(vdes, end=M) is not None
                    set_inputs()
                    st, fl, dv = mv.state_update_hip(vel, vd, end=M)
                    same = same and bool(np.array_equal(st[:M], ref_state[:M]) and np.array_equal(fl[:M], ref_flags[:M]))
                    if os.environ.get("BSP_DEBUG") and not same:
                        bad = np.flatnonzero((st != ref_state) | (fl != ref_flags))
                        sys.stderr.write("%s: %s\n" % (name, [(int(i), int(state[i]), int(fstate[i]), int(st[i]), int(ref_state[i]), int(fl[i]), int(ref_flags[i]), int(dv[i]), int(ag["flock"][i])) for i in bad[:12]]))
                    times.append(mv.hip_state_work_seconds())
                    if times[-1] == min(times):
                        best_parts = mv.hip_state_times()
                out[name] = {"identical": same, "hip_ms_per_tick": min(times[1:]) * 1e3, "hip_ms_per_tick_all": [t * 1e3 for t in times[1:]],
                             "hip_ms_parts": best_parts, "decided_on_device": float(((dv[:M] & 0x80) == 0).mean())}
            mv.hip_resident_state_pass(False)
            out["resident_passes"] = mv.hip_resident_passes()
            out["identical"] = out["host_buffers"]["identical"] and out["resident"]["identical"]
            out["decided_on_device"] = out["resident"]["decided_on_device"]
            out["hip_ms_per_tick"] = out["resident"]["hip_ms_per_tick"]
            out["host_threads"] = args.threads
            out["to_arrived"], out["to_waiting"] = int(((st[:M] == 2) & (state[:M] != 2)).sum()), int(((st[:M] == 4) & (state[:M] != 4)).sum())
            out["slab_mix"] = {str(k): int((state[:M] == k).sum()) for k in (1, 5, 6, 7, 8)}
            out["speedup_vs_1core"] = out["cpu_ms_per_tick_1core"] / out["hip_ms_per_tick"]
            if not args.plain:
                out["settle_stats"] = list(mv.hip_settle_stats())
                out["surround_differ"] = mv.hip_surround_differ()
    finally:
        pfref.RefNav.hip_shutdown()
        pfref.RefMove.unload()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
