#!/usr/bin/env python3
"""How many host cores does this box really give us?  (cpu_count vs affinity vs cgroup quota vs
measured scaling of the reference's N_FlowFieldUpdate over pthreads.)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import pfref
from permafrost_engine_amd import synth
info = {"cpu_count": os.cpu_count(), "affinity": len(os.sched_getaffinity(0))}
for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    if os.path.exists(p):
        info[p] = open(p).read().strip()
grid = synth.cost_grid(4, 4, seed=1234)
nav = pfref.RefNav(synth.to_chunks(grid))
reqs = np.zeros(2048, pfref.FIELD_REQ_DTYPE)
rng = np.random.RandomState(0)
cells = synth.passable_cells(grid)
c = cells[rng.randint(len(cells), size=2048)]
reqs["type"] = 1; reqs["faction_id"] = 15
reqs["chunk_r"], reqs["tile_r"] = c[:, 0] // 64, c[:, 0] % 64
reqs["chunk_c"], reqs["tile_c"] = c[:, 1] // 64, c[:, 1] % 64
for nt in (1, 4, 8, 16, 32, 64, 128, 256):
    t = nav.field_bench(reqs, reps=max(1, nt // 4), nthreads=nt)
    info["fields_per_s_%d" % nt] = 2048 * max(1, nt // 4) / t
print(json.dumps(info, indent=1))
