#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE ITSELF (oracle/_ref/libpfref.so = the
reference's own nav/field/clearpath/movement sources compiled in place from /root/reference).

Run in the build container (needs /root/reference or a prebuilt oracle/_ref):
    python tests/tools/make_golden.py
The fixtures pin the oracle restatement (tests/test_oracle_cpu.py) and the HIP path
(tests/test_golden_gpu.py) where the reference cannot travel.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import navoracle, pfref          # noqa: E402
from tests import cases                      # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def o_reqs(ref_reqs):
    out = np.zeros(len(ref_reqs), navoracle.FIELD_REQ_DTYPE)
    for name in ("layer", "type", "faction_id", "chunk_r", "chunk_c", "tile_r", "tile_c",
                 "port_r0", "port_c0", "port_r1", "port_c1", "next_r0", "next_c0", "next_r1",
                 "next_c1", "next_chunk_r", "next_chunk_c", "port_iid", "next_iid"):
        out[name] = ref_reqs[name]
    out["flags"] = np.where(ref_reqs["inout"] != 0, 1, 0)
    return out


def fields(w=3, h=3, seed=77, name="fields_3x3"):
    grid = cases.synth.cost_grid(w, h, seed=seed)
    blk = cases.random_blockers(grid, seed=3)
    grid, nav = cases.ref_nav_for(w, h, seed=seed, blockers=blk)
    reqs_t = cases.tile_requests(grid, 16, seed=6)
    reqs_p, before, _ = cases.planner_requests(nav, grid, pairs=10, seed=10)
    reqs = np.concatenate([reqs_t, reqs_p])
    before = np.concatenate([np.zeros((len(reqs_t), 64, 64), np.uint8), before])
    reqs, before = cases.with_inplace(reqs, before, seed=4, count=8)
    dirs, integ = cases.ref_fields(nav, reqs, before)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"),
                        cost=nav.plane(pfref.PLANE_COST), blockers=nav.plane(pfref.PLANE_BLOCKERS),
                        local_islands=nav.plane(pfref.PLANE_LOCAL_ISLANDS),
                        reqs=o_reqs(reqs).view(np.uint8).reshape(len(reqs), 32), before=before,
                        dirs=dirs, integ=integ)
    print("%s: %d requests (%d portal, %d in-place)" % (
        name, len(reqs), int((reqs["type"] == 0).sum()), int((reqs["inout"] != 0).sum())))


def agents(w=4, h=4, seed=21, name="agents_4x4", n=700):
    grid = cases.synth.cost_grid(w, h, seed=seed)
    blk = cases.random_blockers(grid, seed=8, frac=0.02)
    grid, nav = cases.ref_nav_for(w, h, seed=seed, blockers=blk)
    k = 3
    world = cases.make_agents(grid, n, k, seed=99, clustered=True)
    mv, dest_ids = cases.ref_move_for(nav, world)
    mv.velocity(None)                        # first pass fills / merges the reference's field cache
    ref_vel = mv.velocity(None)
    vdes = mv.vdes()
    slots, pool = cases.cached_field_table(nav, dest_ids, w, h)
    arrays = cases.step_arrays(world, vdes, [mv.flock_order(f) for f in range(k)])
    out = {k2: np.asarray(v) for k2, v in arrays.items()}
    out.update(ref_vel=ref_vel, ref_vdes_sampled=vdes, flock_field_slot=slots, field_pool=pool,
               cost=nav.plane(pfref.PLANE_COST), blockers=nav.plane(pfref.PLANE_BLOCKERS),
               local_islands=nav.plane(pfref.PLANE_LOCAL_ISLANDS))
    # spatial queries straight from bitmap_grid.h
    q = np.concatenate([world["pos_xz"][::11], [[0, 0], [-w * 128.0, h * 128.0]]]).astype(np.float32)
    out["sq_query"] = q
    bounds = (-w * 128.0, w * 128.0, -h * 128.0, h * 128.0)
    for key, r, cap in (("r30", 30.0, 128), ("r10", 10.0, 512), ("wide", 1400.0, 200)):
        c, ids = pfref.spatial_query(bounds, world["pos_xz"], q, r, cap)
        out["sq_%s_range" % key], out["sq_%s_cap" % key] = np.float32(r), np.int32(cap)
        out["sq_%s_counts" % key], out["sq_%s_ids" % key] = c, ids
    # ClearPath problems
    ent, des, dyn, nd, stat, ns = cases.cp_problems(7, 160, 10, 6, 7.0)
    exp = np.zeros((len(ent), 2), np.float32)
    for i in range(len(ent)):
        exp[i] = pfref.clearpath_new_velocity(ent[i], des[i], dyn[i, :nd[i]], stat[i, :ns[i]])
    out.update(cp_ent=ent, cp_des=des, cp_dyn=dyn, cp_nd=nd, cp_stat=stat, cp_ns=ns, cp_out=exp)
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    pfref.RefMove.unload()
    print("%s: %d agents, %d flocks, %d cached fields" % (name, n, k, len(pool)))


def state_pass(name="state_4x4"):
    """The state pass around navhip_state_update: the heading gate, the settled-neighbour count and the
    arrival overlay's settle rule -- inputs of tests/test_state_gpu.py, answers of the reference's own
    entity_compute_update / adjacent_settled_count / G_Arrival_ShouldSettle."""
    from tests import test_state_gpu as T
    out = {}
    nav, world, new_vel, vdes, facing, off, tight = T.gate_inputs()
    next_rot = pfref.RefMove.dir_quat(facing)
    mv, _ = cases.ref_move_for(nav, world)
    turn, vel = mv.heading_gate(new_vel, vdes, next_rot)
    pfref.RefMove.unload()
    out.update(gate_pos=world["pos_xz"], gate_vel=world["vel_xz"], gate_state=world["state"], gate_next_rot=next_rot,
               gate_new_vel=new_vel, gate_vdes=vdes, gate_tight=tight, gate_turn=turn)
    nav, world, uids = T.count_inputs()
    mv, _ = cases.ref_move_for(nav, world)
    out.update(count_pos=world["pos_xz"], count_radius=world["radius"], count_flags=world["flags"],
               count_state=world["state"], count_uids=uids, count_ref=mv.settled_count(uids))
    pfref.RefMove.unload()
    grid, nav, zones, units = T._zone_world(seed=9)
    out.update(zone_cost=nav.plane(pfref.PLANE_COST, 0), zone_blockers=nav.plane(pfref.PLANE_BLOCKERS, 0), n_zones=len(zones))
    for i, (z, u) in enumerate(zip(zones, units)):
        settle, keys, after = pfref.arrival_should_settle(nav, z, u)
        out["zone%d_scalars" % i] = np.array([z["layer"], z["radius"], z["active_row"], z["num_rows"]], np.int32)
        out["zone%d_floats" % i] = np.array([z["centre_xz"][0], z["centre_xz"][1], z["unit_radius"], z["fill_frac"]], np.float32)
        out["zone%d_slots" % i], out["zone%d_ring" % i], out["zone%d_keys" % i] = z["slots_xz"], z["slot_ring"], keys
        for f, a in u.items():
            out["unit%d_%s" % (i, f)] = a
        out["ref%d_settle" % i] = settle
        for f, a in after.items():
            out["ref%d_%s" % (i, f)] = a
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("%s: gate %d units (%d turn), count %d units, settle %d zones" % (
        name, len(turn), int(turn.sum()), len(uids), len(zones)))


if __name__ == "__main__":
    os.makedirs(GOLD, exist_ok=True)
    if "--state" in sys.argv:
        state_pass()
        sys.exit(0)
    fields()
    agents()
    # non-square maps (the multi-GPU world is 32 x 64 chunks at 8 ranks)
    fields(5, 2, seed=305, name="fields_5x2")
    agents(5, 2, seed=305, name="agents_5x2", n=500)
    state_pass()
