import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import pfref
from permafrost_engine_amd import navhip
from tests import test_agents_gpu as T
np.set_printoptions(precision=6, suppress=False, linewidth=160)
ctx = navhip.NavContext(1, 1)
for seed, md, ms, sp in [(1, 6, 3, 9.0), (5, 3, 3, 2.5), (3, 12, 0, 5.0), (4, 0, 12, 5.0)]:
    nq = 200
    ent, des, dyn, nd, stat, ns = T._cp_problems(seed, nq, md, ms, sp)
    exp = np.stack([pfref.clearpath_new_velocity(ent[i], des[i], dyn[i, :nd[i]], stat[i, :ns[i]]) for i in range(nq)])
    got = ctx.G_ClearPath_NewVelocity(ent, des, dyn, nd, stat, ns)
    err = T._vel_err(np.nan_to_num(got), np.nan_to_num(exp))
    bad = np.flatnonzero(~(err <= 1e-4))
    print("seed", seed, "bad", len(bad), "of", nq, "exact", int((got == exp).all(1).sum()))
    for b in bad[:6]:
        print("  q", b, "nd", nd[b], "ns", ns[b], "des", des[b], "exp", exp[b], "got", got[b],
              "exp==des", np.array_equal(exp[b], des[b]), "got==des", np.array_equal(got[b], des[b]))
