#!/usr/bin/env python3
"""What the jam looks like to ClearPath (developer tool): speeds and neighbour kinds of the crowded world."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from permafrost_engine_amd import tick
T = tick.NavTick(crowd_cells=17, los=True, flow_velocities=True)
for t in range(1, 41):
    T.step()
    if t in (5, 20, 40):
        T.sync()
        v = T.t["vel_xz"].cpu().numpy()
        sp = np.linalg.norm(v, axis=1)
        cnt = None
        print("tick", t, "speed mean %.3f median %.3f  frac<0.3: %.2f  frac==0: %.2f" % (sp.mean(), np.median(sp), (sp < 0.3).mean(), (sp == 0).mean()),
              "lists", T.ctx.last_step_lists(), flush=True)
T.close()
