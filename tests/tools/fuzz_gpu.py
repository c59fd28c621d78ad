#!/usr/bin/env python3
"""Randomised GPU-vs-oracle sweep (bit-exactness of velocities, fields, LOS, blockers) over many
seeds and world shapes -- looks for rare divergences (fast-path margins, caps, edge tiles) that
the fixed test cases might miss.  Prints one line per case and a summary; exit code 1 on a mismatch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge            # noqa: E402
ge.build_navhip()
from permafrost_engine_amd import navhip, synth     # noqa: E402
from oracle import navoracle                          # noqa: E402
from tests import cases                               # noqa: E402


def main(n_cases=24, first=0):
    bad = 0
    for case in range(first, first + n_cases):
        rng = np.random.RandomState(1000 + case)
        W = int(rng.choice([2, 3, 4]))
        frac = float(rng.choice([0.1, 0.2, 0.35]))
        grid = synth.cost_grid(W, W, seed=500 + case, frac_impassable=frac)
        blk = cases.random_blockers(grid, seed=case, frac=float(rng.choice([0.0, 0.02, 0.06])))
        chunks = synth.to_chunks(grid)
        onav = navoracle.OracleNav(chunks, blk)
        onav.set_layer(0, local_islands=onav.local_islands(0))
        ctx = navhip.NavContext(W, W)
        ctx.upload_plane(0, navhip.PLANE_COST_BASE, chunks)
        ctx.upload_plane(0, navhip.PLANE_BLOCKERS, blk)
        ctx.upload_plane(0, navhip.PLANE_LOCAL_ISLANDS, onav.plane(0, "local_islands"))
        # fields
        K = int(rng.choice([2, 3, 5]))
        liid = synth.from_chunks(onav.plane(0, "local_islands"))
        cells = synth.passable_cells(grid, synth.from_chunks(blk))
        dests = cells[rng.choice(len(cells), K, replace=False)]
        cols = synth.whole_map_requests(grid, dests, liid)
        reqs = cases.cols_to_reqs(cols, navhip.FIELD_REQ_DTYPE)
        dirs, integ = ctx.N_FlowFieldUpdate(reqs, want_integ=True)
        ed, ei = onav.build_fields(reqs.view(navoracle.FIELD_REQ_DTYPE), want_integ=True)
        ok_f = np.array_equal(dirs, ed) and np.array_equal(integ, ei)
        # agents: clustered or spread, sampling the fields on the device
        n = int(rng.choice([800, 2000, 4000]))
        world = cases.make_agents(grid, n, K, seed=case, clustered=bool(rng.rand() < 0.6),
                                  sigma=float(rng.choice([15.0, 40.0, 90.0])))
        world["flock_target_xz"] = synth.cell_centre(W, W, dests[:, 0], dests[:, 1])
        slot = -np.ones((K, W * W), np.int32)
        slot[cols["dest"], cols["chunk_r"] * W + cols["chunk_c"]] = np.arange(len(reqs))
        a = cases.step_arrays(world, None)
        a["flock_field_slot"], a["field_pool"] = slot, dirs.reshape(len(dirs), 4096)
        hz = int(rng.choice([20, 20, 10]))
        out = ctx.agent_step(a, hz=hz)
        exp = onav.agent_step(a, hz=hz, nthreads=8)
        same = out["vel_xz"].view(np.uint32) == exp["vel_xz"].view(np.uint32)
        nanboth = np.isnan(out["vel_xz"]) & np.isnan(exp["vel_xz"])
        ok_v = bool((same | nanboth).all()) and np.array_equal(out["status"], exp["status"]) \
            and np.array_equal(out["new_pos_xz"].view(np.uint32), exp["new_pos_xz"].view(np.uint32))
        d = np.linalg.norm(out["vel_xz"].astype(np.float64) - exp["vel_xz"], axis=1)
        rel = np.nanmax(d / np.maximum(np.linalg.norm(exp["vel_xz"].astype(np.float64), axis=1), 1e-3))
        print("case %2d: W=%d K=%d n=%d hz=%d  fields %s  velocities %s (max rel %.2g, %d/%d bit-identical)"
              % (case, W, K, n, hz, "ok" if ok_f else "DIFF", "ok" if ok_v else "DIFF", rel,
                 int((same | nanboth).all(1).sum()), n), flush=True)
        bad += (not ok_f) + (not ok_v)
        ctx.close()
    print("fuzz: %d mismatching cases of %d" % (bad, n_cases))
    return 1 if bad else 0


def state_sweep(n_cases=24, first=100):
    """The settle rule of the arrival overlay (navhip_arrival_settle) against the reference's own G_Arrival_ShouldSettle
    (oracle/_ref) over many seeds of tests/test_state_gpu.py's zone world: answers and the unit state left behind."""
    from oracle import pfref
    from tests import test_state_gpu as T
    if not pfref.available():
        print("state sweep: oracle/_ref is not present")
        return 0
    bad = 0
    for seed in range(first, first + n_cases):
        grid, nav, zones, units = T._zone_world(seed=seed)
        ref, keys, afters = [], [], []
        for z, u in zip(zones, units):
            s_, k_, a_ = pfref.arrival_should_settle(nav, z, u)
            ref.append(s_); keys.append(k_); afters.append(a_)
        ctx = T._upload(navhip, nav, layers=(0,))
        cat = {f: np.concatenate([u[f] for u in units]) for f in units[0]}
        nq = len(cat["zone"])
        cat["uid"] = np.arange(nq, dtype=np.int32)
        world = {"pos_xz": np.zeros((nq, 2), np.float32), "vel_xz": cat["vel_xz"], "radius": cat["radius"]}
        got, after = ctx.arrival_settle(world, zones, keys, cat)
        ctx.close()
        r = np.concatenate(ref)
        ok = np.array_equal(got, r) and all(np.array_equal(after[f], np.concatenate([a[f] for a in afters])) for f in after)
        print("settle seed %d: %s (%d of %d settle)" % (seed, "ok" if ok else "DIFF", int(r.sum()), nq), flush=True)
        bad += not ok
    print("state sweep: %d mismatching cases of %d" % (bad, n_cases))
    return 1 if bad else 0


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser(description="cases first .. first + n_cases - 1 of either sweep")
    ap.add_argument("n_cases", nargs="?", type=int, default=24)
    ap.add_argument("--first", type=int, default=None, help="first case / seed (default 0; --state: 100)")
    ap.add_argument("--state", action="store_true", help="the settle rule against the reference build")
    a = ap.parse_args()
    if a.state:
        sys.exit(state_sweep(a.n_cases, 100 if a.first is None else a.first))
    sys.exit(main(a.n_cases, 0 if a.first is None else a.first))
