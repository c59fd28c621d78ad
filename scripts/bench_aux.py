#!/usr/bin/env python3
"""Secondary measurements recorded in DESIGN.md: the host-buffer (PCIe-inclusive) rates of the two
hot entry points, LOS field throughput, repair-build throughput.  Prints one JSON object."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge            # noqa: E402
ge.build_navhip()
from permafrost_engine_amd import navhip, synth   # noqa: E402


def timed(fn, reps=5):
    fn()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t0) / reps


def main():
    import torch
    torch.cuda.init()          # (before libnavhip touches HIP: torch brings its own runtime)
    W, K, N = 16, 64, 100_000
    grid = synth.cost_grid(W, W, seed=1234)
    ctx = navhip.NavContext(W, W)
    ctx.upload_plane(0, navhip.PLANE_COST_BASE, synth.to_chunks(grid))
    ctx.upload_plane(0, navhip.PLANE_BLOCKERS, np.zeros((W, W, 64, 64), np.uint16))
    ctx.relabel_local_islands(0)
    liid = synth.from_chunks(ctx.download_plane(0, navhip.PLANE_LOCAL_ISLANDS))
    dests = synth.destinations(grid, K, seed=42)
    cols = synth.whole_map_requests(grid, dests, liid)
    reqs = navhip.make_reqs(len(cols["type"]))
    for k in synth.REQ_FIELDS:
        reqs[k] = cols[k]
    out = {}
    # host-buffer chunk fields: H2D requests + kernel + D2H 4 KB per field
    t = timed(lambda: ctx.N_FlowFieldUpdate(reqs), reps=3)
    out["fields_host_api_cells_per_s"] = len(reqs) * 4096 / t
    out["fields_host_api_ms"] = t * 1e3
    dirs, _ = ctx.N_FlowFieldUpdate(reqs)
    # host-buffer agent step: H2D snapshot (incl. the 67 MB field pool) + kernels + D2H results
    slot = -np.ones((K, W * W), np.int32)
    slot[cols["dest"], cols["chunk_r"] * W + cols["chunk_c"]] = np.arange(len(reqs))
    ag = synth.agents(grid, N, K, seed=7)
    offs, members = navhip.flock_csr(ag["flock"], K)
    arrays = {"pos_xz": ag["pos"], "vel_xz": ag["vel"], "radius": ag["radius"], "max_speed": ag["max_speed"],
              "speed": ag["speed"], "flags": np.full(N, navhip.ENTITY_FLAG_MOVABLE, np.uint32),
              "state": np.zeros(N, np.uint8), "has_dest_los": np.zeros(N, np.uint8), "flock": ag["flock"],
              "flock_target_xz": synth.cell_centre(W, W, dests[:, 0], dests[:, 1]),
              "flock_offsets": offs, "flock_members": members, "flock_field_slot": slot,
              "field_pool": dirs.reshape(len(dirs), 4096), "vdes_xz": None}
    t = timed(lambda: ctx.agent_step(arrays, want=("vel_xz", "new_pos_xz", "status")), reps=3)
    out["agents_host_api_steps_per_s"] = N / t
    out["agents_host_api_ms"] = t * 1e3
    arrays2 = dict(arrays)
    arrays2["field_pool"] = None
    arrays2["flock_field_slot"] = None
    arrays2["vdes_xz"] = np.tile(np.array([[1.0, 0.0]], np.float32), (N, 1))
    t = timed(lambda: ctx.agent_step(arrays2, want=("vel_xz", "new_pos_xz", "status")), reps=3)
    out["agents_host_api_given_vdes_steps_per_s"] = N / t
    # the host-buffer step the binding uses (bindings/permafrost/move_hip.c): fields resident in the device pool
    # (built there once; the step uploads the 3 MB entity snapshot and downloads 1.7 MB of results),
    # navhip_agent_step_submit / _poll through the pinned staging area -- against the SAME step with
    # everything already on the device (navhip_agent_step_dev)
    ctx.pool_create(len(reqs) + 8, K)
    ctx.pool_build(reqs, readback=False)
    all_ids = np.array([navhip.N_FlowFieldID(reqs[i]) for i in range(len(reqs))], np.uint64)
    ctx.pool_map(cols["dest"], cols["chunk_r"], cols["chunk_c"], all_ids)
    arrays3 = dict(arrays)
    arrays3["field_pool"] = None
    arrays3["flock_field_slot"] = None
    arrays3["use_resident_pool"] = True
    import ctypes as C
    L = navhip.lib()

    def c_level(arr, outs, epoch=0):
        """submit + wait on prebuilt structs: the C boundary itself, no Python per-call work"""
        w, keep = navhip.make_world(W, W, arr, 20)
        w.n_field_slots = navhip.POOL_RESIDENT
        w.static_epoch = epoch
        so = navhip.StepOut()
        so.vel_xz, so.new_pos_xz, so.status = (o.ctypes.data for o in outs)

        def call():
            assert L.navhip_agent_step_submit(ctx._h, C.byref(w), C.byref(so)) == 0
            assert L.navhip_agent_step_wait(ctx._h) == 0
        return timed(call, reps=10), keep

    outs = (np.zeros((N, 2), np.float32), np.zeros((N, 2), np.float32), np.zeros(N, np.uint8))
    t_res, _k1 = c_level(arrays3, outs)
    out["agents_host_resident_pool_ms"] = t_res * 1e3
    out["agents_host_resident_pool_steps_per_s"] = N / t_res
    outs_e = tuple(np.zeros_like(o) for o in outs)
    t_ep, _k3 = c_level(arrays3, outs_e, epoch=7)
    out["agents_host_resident_pool_static_epoch_ms"] = t_ep * 1e3
    assert np.array_equal(outs_e[0].view(np.uint32), outs[0].view(np.uint32))
    # the same with the caller's arrays in pinned memory (navhip_host_alloc): no staging memcpy
    def pin(v):
        v = np.ascontiguousarray(v)
        buf = navhip.host_alloc(max(1, v.nbytes))
        pv = np.frombuffer(buf, dtype=v.dtype, count=v.size).reshape(v.shape)
        pv[...] = v
        return pv
    pinned = {k: (pin(v) if isinstance(v, np.ndarray) else v) for k, v in arrays3.items()}
    pouts = tuple(pin(o) for o in outs)
    t_pin, _k2 = c_level(pinned, pouts)
    out["agents_host_resident_pool_pinned_arrays_ms"] = t_pin * 1e3
    assert np.array_equal(pouts[0].view(np.uint32), outs[0].view(np.uint32))
    dev = torch.device("cuda", 0)
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in arrays.items()
         if isinstance(v, np.ndarray)}
    d["flock_target_xz"] = d["flock_target_xz"].float()
    wdev, keep = navhip.make_world(W, W, d, hz=20)
    o_vel = torch.zeros((N, 2), dtype=torch.float32, device=dev)
    o_pos = torch.zeros((N, 2), dtype=torch.float32, device=dev)
    o_st = torch.zeros(N, dtype=torch.uint8, device=dev)
    so = navhip.StepOut()
    so.vel_xz, so.new_pos_xz, so.status = o_vel.data_ptr(), o_pos.data_ptr(), o_st.data_ptr()
    st = torch.cuda.Stream(device=dev)

    def step_dev():
        ctx.agent_step_dev(wdev, so, stream=st.cuda_stream)
        st.synchronize()
    t_dev = timed(step_dev, reps=10)
    out["agents_dev_same_world_ms"] = t_dev * 1e3
    out["agents_host_resident_over_dev"] = t_res / t_dev
    out["agents_host_resident_pinned_over_dev"] = t_pin / t_dev
    out["agents_host_resident_static_epoch_over_dev"] = t_ep / t_dev
    # LOS: the destination-chunk field of 4096 random destinations
    rng = np.random.RandomState(1)
    cells = synth.passable_cells(grid)
    pick = cells[rng.randint(len(cells), size=4096)]
    lr = np.zeros(4096, navhip.LOS_REQ_DTYPE)
    lr["faction_id"] = 0xF
    lr["chunk_r"] = lr["target_chunk_r"] = pick[:, 0] // 64
    lr["chunk_c"] = lr["target_chunk_c"] = pick[:, 1] // 64
    lr["target_tile_r"], lr["target_tile_c"] = pick[:, 0] % 64, pick[:, 1] % 64
    t = timed(lambda: ctx.N_LOSFieldCreate(lr), reps=2)
    out["los_fields_per_s_host_api"] = 4096 / t
    out["los_ms_per_4096"] = t * 1e3
    ctx.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
