#!/usr/bin/env python3
"""The cone-test-order experiment of tests/tools/cp_order_model.py on an EVOLVED jam, CPU only: a few flocks of the crowded
world (synth.agents(crowd_cells=17): ~1 560 agents packed into ~35 x 35 cells each) are stepped for T ticks with the
restatement oracle (oracle/navoracle.c: the velocity step + the position accept test, desired direction = straight at
the flock target), then every agent's ClearPath problem of the last tick is rebuilt the way find_neighbours does
(movement.c:2768: r = 10, static = still or slower than 0.3, 32 of each at most) and the candidates of its first
attempt are tested against the cones in the orders compared.  Developer tool (test infrastructure: uses oracle/).

    python tests/tools/jam_evolved_model.py [--flocks 3] [--ticks 40] [--problems 120] [--threads 8]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
sys.path.insert(0, ROOT)
import cp_model as M            # noqa: E402
import cp_order_model as O      # noqa: E402
from oracle import navoracle    # noqa: E402
from permafrost_engine_amd import synth    # noqa: E402

f32 = np.float32


def evolved_problems(flocks, ticks, max_problems, threads=8):
    """[(ent, des, nbs, isdyn)] of the tick after `ticks` ticks of a `flocks`-flock jam (see the module docstring)."""
    W = 16
    grid = synth.cost_grid(W, W, seed=1234)
    k = flocks
    n = 1562 * k
    ag = synth.agents(grid, n, k, seed=7, crowd_cells=17)
    dests = synth.destinations(grid, k, seed=42)
    targets = synth.cell_centre(W, W, dests[:, 0], dests[:, 1]).astype(f32)
    nav = navoracle.OracleNav(synth.to_chunks(grid))
    pos, vel = ag["pos"].astype(f32), ag["vel"].astype(f32)
    flock = ag["flock"]
    lists = [np.flatnonzero(flock == f) for f in range(k)]
    offs = np.zeros(k + 1, np.int32)
    offs[1:] = np.cumsum([len(l) for l in lists])
    base = {"radius": ag["radius"], "max_speed": ag["max_speed"], "speed": ag["speed"],
            "flags": np.full(n, 1 << 3, np.uint32), "state": np.zeros(n, np.uint8),
            "has_dest_los": np.zeros(n, np.uint8), "flock": flock, "flock_target_xz": targets,
            "flock_offsets": offs, "flock_members": np.concatenate(lists).astype(np.int32)}

    def toward():
        d = targets[flock] - pos
        return (d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-3)).astype(f32)
    for t in range(ticks):
        out = nav.agent_step(dict(base, pos_xz=pos, vel_xz=vel, vdes_xz=toward()), hz=20, nthreads=threads)
        pos, vel = out["new_pos_xz"].copy(), out["vel_xz"].copy()
    from scipy.spatial import cKDTree
    tree = cKDTree(pos)
    out = nav.agent_step(dict(base, pos_xz=pos, vel_xz=vel, vdes_xz=toward()), hz=20, nthreads=threads)
    vpref = out["vpref_xz"]
    rng = np.random.RandomState(3)
    speed = np.linalg.norm(vel, axis=1)
    probs = []
    for uid in rng.permutation(n):
        if len(probs) >= max_problems:
            break
        nb = [j for j in tree.query_ball_point(pos[uid], 10.0) if j != uid]
        stat = [j for j in nb if speed[j] < 0.3][:32]
        dyn = [j for j in nb if speed[j] >= 0.3][:32]
        if len(stat) + len(dyn) < 17:
            continue
        order = dyn + stat
        nbs = np.zeros((len(order), 5), f32)
        nbs[:, 0:2] = pos[order]
        nbs[:len(dyn), 2:4] = vel[dyn]
        nbs[:, 4] = 1.0
        isdyn = np.arange(len(order)) < len(dyn)
        ent = np.array([pos[uid, 0], pos[uid, 1], vel[uid, 0], vel[uid, 1], 1.0], f32)
        probs.append((ent, vpref[uid].astype(f32), nbs, isdyn))
    return probs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flocks", type=int, default=3)
    ap.add_argument("--ticks", type=int, default=40)
    ap.add_argument("--problems", type=int, default=120)
    ap.add_argument("--threads", type=int, default=8)
    args = ap.parse_args()
    W = 16
    grid = synth.cost_grid(W, W, seed=1234)
    k = args.flocks
    n = 1562 * k
    ag = synth.agents(grid, n, k, seed=7, crowd_cells=17)
    dests = synth.destinations(grid, k, seed=42)
    targets = synth.cell_centre(W, W, dests[:, 0], dests[:, 1]).astype(f32)
    nav = navoracle.OracleNav(synth.to_chunks(grid))
    pos, vel = ag["pos"].astype(f32), ag["vel"].astype(f32)
    flock = ag["flock"]
    lists = [np.flatnonzero(flock == f) for f in range(k)]
    offs = np.zeros(k + 1, np.int32)
    offs[1:] = np.cumsum([len(l) for l in lists])
    base = {"radius": ag["radius"], "max_speed": ag["max_speed"], "speed": ag["speed"],
            "flags": np.full(n, 1 << 3, np.uint32), "state": np.zeros(n, np.uint8),
            "has_dest_los": np.zeros(n, np.uint8), "flock": flock, "flock_target_xz": targets,
            "flock_offsets": offs, "flock_members": np.concatenate(lists).astype(np.int32)}
    for t in range(args.ticks):
        d = targets[flock] - pos
        d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-3)
        arrays = dict(base, pos_xz=pos, vel_xz=vel, vdes_xz=d.astype(f32))
        out = nav.agent_step(arrays, hz=20, nthreads=args.threads)
        pos, vel = out["new_pos_xz"].copy(), out["vel_xz"].copy()
        if t % 10 == 9:
            print("tick %d: mean speed %.3f wu per tick" % (t + 1, float(np.linalg.norm(vel, axis=1).mean())), flush=True)
    # the problems of the next tick
    from scipy.spatial import cKDTree
    tree = cKDTree(pos)
    d = targets[flock] - pos
    d /= np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-3)
    out = nav.agent_step(dict(base, pos_xz=pos, vel_xz=vel, vdes_xz=d.astype(f32)), hz=20, nthreads=args.threads)
    vpref = out["vpref_xz"]
    rng = np.random.RandomState(3)
    speed = np.linalg.norm(vel, axis=1)
    tot = {m: 0 for m in ("depth", "row_hint", "col_hint", "both")}
    ncand = nvalid_pairs = used = 0
    nbc = []
    for uid in rng.permutation(n):
        if used >= args.problems:
            break
        nb = [j for j in tree.query_ball_point(pos[uid], 10.0) if j != uid]
        stat = [j for j in nb if speed[j] < 0.3][:32]
        dyn = [j for j in nb if speed[j] >= 0.3][:32]
        if len(stat) + len(dyn) < 17:
            continue
        order = dyn + stat
        nbs = np.zeros((len(order), 5), f32)
        nbs[:, 0:2] = pos[order]
        nbs[:len(dyn), 2:4] = vel[dyn]                         # (static neighbours: velocity forced to zero, :2820)
        nbs[:, 4] = 1.0
        isdyn = np.arange(len(order)) < len(dyn)
        ent = np.array([pos[uid, 0], pos[uid, 1], vel[uid, 0], vel[uid, 1], 1.0], f32)
        des = vpref[uid].astype(f32)
        C = M.make_cones(ent, nbs, isdyn)
        nc = len(C["ax"])
        ex, ez = f32(ent[0]), f32(ent[1])
        dwx, dwz = ex + des[0], ez + des[1]
        if nc < 17 or not M.inside_cone(C, np.arange(nc), dwx, dwz).any():
            continue
        px, pz, dx, dz, s = M.rays_of(C)
        nr = 2 * nc
        I, J = np.meshgrid(np.arange(nr), np.arange(nr), indexing="ij")
        ok, cx, cz = M.ray_isect(px[I], pz[I], dx[I], dz[I], s[I], px[J], pz[J], dx[J], dz[J], s[J])
        ok &= I != J
        order_c = O.depth_order(C, dwx, dwz)
        for m in tot:
            t_, nc_ = O.count(C, order_c, cx, cz, ok, m)
            tot[m] += t_
        ncand += nc_
        nvalid_pairs += nr * nr
        used += 1
        nbc.append((len(dyn), len(stat)))
    nbc = np.array(nbc)
    print("%d problems of the evolved jam (mean %.1f dynamic + %.1f static neighbours), %d valid candidates of %d ray pairs"
          % (used, nbc[:, 0].mean(), nbc[:, 1].mean(), ncand, nvalid_pairs))
    for m, t_ in tot.items():
        print("%-9s %.2f cone tests per candidate" % (m, t_ / max(1, ncand)))


if __name__ == "__main__":
    main()
