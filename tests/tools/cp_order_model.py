#!/usr/bin/env python3
"""Which cone should a candidate of the ClearPath search be tested against FIRST?  CPU-only experiment on the jam
problems of tests/tools/cp_model.py (numpy float32 model of clearpath.c:552-660): every valid ray-pair candidate of a
problem's first attempt is tested against the cones in a given order until one contains it, and the tests are
counted.  Orders compared (the RESULT of the search does not depend on the order -- a candidate is dropped by any
cone that contains it and accepted only when none does):

  depth     the kernel's order: how deep des_v lies inside each cone, deepest first (agent_group.h, NH_CP_ORDER_DEPTH)
  row_hint  the cone that dropped the LAST candidate of the same row (candidates (i, j), (i, j') lie on ray i) first,
            then depth order
  col_hint  the same per column (candidates of one column lie on line j)
  both      row hint, then column hint, then depth order

    python tests/tools/cp_order_model.py [--n 120] [--crowd 17]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests", "tools"))
sys.path.insert(0, ROOT)
import cp_model as M      # noqa: E402

f32 = np.float32


def depth_order(C, dwx, dwz):
    """cones by how deep des_v (world space) lies inside them: smaller of its distances to the two side lines,
    deepest first; cones that do not contain it after those, least outside first"""
    qx, qz = dwx - C["ax"], dwz - C["az"]
    dl = qz * C["Lx"] - qx * C["Lz"]           # > 0: right of the left side (inside)
    dr = -(qz * C["Rx"] - qx * C["Rz"])        # > 0: left of the right side (inside)
    depth = np.minimum(dl, dr)
    return np.argsort(-depth, kind="stable")


def count(C, order, cx, cz, ok, mode):
    nr = ok.shape[0]
    nc = len(order)
    inside = np.zeros((nc,) + ok.shape, bool)
    for c in range(nc):
        inside[c] = M.inside_cone(C, c, cx, cz)
    rank = np.empty(nc, int)
    rank[order] = np.arange(nc)
    tests = 0
    ncand = 0
    row_hint = np.full(nr, -1)
    # the kernel's candidate stream: column by column (here: ascending column index), rows inside a column
    for j in range(nr):
        col_hint = -1
        for i in range(nr):
            if not ok[i, j]:
                continue
            ncand += 1
            first = []
            if mode in ("row_hint", "both") and row_hint[i] >= 0:
                first.append(row_hint[i])
            if mode in ("col_hint", "both") and col_hint >= 0 and col_hint not in first:
                first.append(col_hint)
            hit = -1
            t = 0
            for c in first:
                t += 1
                if inside[c, i, j]:
                    hit = c
                    break
            if hit < 0:
                for c in order:
                    if c in first:
                        continue
                    t += 1
                    if inside[c, i, j]:
                        hit = c
                        break
            tests += t
            if hit >= 0:
                row_hint[i] = hit
                col_hint = hit
    return tests, ncand


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=120)
    ap.add_argument("--crowd", type=int, default=17)
    args = ap.parse_args()
    probs = M.jam_problems(args.n, args.crowd, 1, 0.0)
    tot = {m: 0 for m in ("depth", "row_hint", "col_hint", "both")}
    ncand = 0
    used = 0
    for ent, des, nbs in probs:
        C = M.make_cones(ent, nbs, np.ones(len(nbs), bool))
        n = len(C["ax"])
        if n < 17:
            continue
        ex, ez = f32(ent[0]), f32(ent[1])
        dwx, dwz = ex + f32(des[0]), ez + f32(des[1])
        if not M.inside_cone(C, np.arange(n), dwx, dwz).any():
            continue
        px, pz, dx, dz, s = M.rays_of(C)
        nr = 2 * n
        I, J = np.meshgrid(np.arange(nr), np.arange(nr), indexing="ij")
        ok, cx, cz = M.ray_isect(px[I], pz[I], dx[I], dz[I], s[I], px[J], pz[J], dx[J], dz[J], s[J])
        ok &= I != J
        order = depth_order(C, dwx, dwz)
        for m in tot:
            t, nc = count(C, order, cx, cz, ok, m)
            tot[m] += t
        ncand += nc
        used += 1
    print("%d problems with 17+ cones, %d valid candidates" % (used, ncand))
    for m, t in tot.items():
        print("%-9s %.2f cone tests per candidate" % (m, t / max(1, ncand)))


if __name__ == "__main__":
    main()
