import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from oracle import pfref
from permafrost_engine_amd import navhip, synth
from tests import cases, test_agents_gpu as T
np.set_printoptions(precision=7, suppress=False, linewidth=170)

def tile_of(w, h, p):
    mp = synth.map_pos(w, h)
    gc = int(abs(mp[0] - p[0]) / 4.0); gr = int(abs(mp[2] - p[1]) / 4.0)
    return gr, gc

# ---- flow sampling
grid, nav = cases.ref_nav_for(4, 4, seed=21)
world = cases.make_agents(grid, 800, 3, seed=77, clustered=False)
mv, dest_ids = cases.ref_move_for(nav, world)
exp_vel = mv.velocity(None); vdes = mv.vdes()
k = len(dest_ids); slots = -np.ones((k, 16), np.int32); pool = []
for f, did in enumerate(dest_ids):
    for cr in range(4):
        for cc in range(4):
            ff = nav.cached_field(did, cr, cc)
            if ff is not None:
                slots[f, cr * 4 + cc] = len(pool); pool.append(ff.reshape(-1))
a = T._step_arrays(world, mv, None); a["flock_field_slot"] = slots; a["field_pool"] = np.stack(pool).astype(np.uint8)
ctx = T._upload(navhip, nav); out = ctx.agent_step(a)
ps = np.isin(world["state"], (0, 5, 6)); clean = ps & ((out["status"] & 6) == 0)
err = T._vel_err(out["vdes_xz"], vdes)
bad = np.flatnonzero(clean & ~(err <= 1e-4))
print("flow: clean", clean.sum(), "bad", len(bad), "miss", int((out["status"] & 2 != 0).sum()), "none", int((out["status"] & 4 != 0).sum()))
pool_a = np.stack(pool).reshape(-1, 64, 64)
for b in bad[:8]:
    gr, gc = tile_of(4, 4, world["pos_xz"][b]); f = world["flock"][b]
    sl = slots[f, (gr // 64) * 4 + gc // 64]
    print(" uid", b, "pos", world["pos_xz"][b], "tile", gr, gc, "slot", sl, "exp", vdes[b], "got", out["vdes_xz"][b])
    print("   field 3x3:\n", pool_a[sl][max(gr % 64 - 1, 0):gr % 64 + 2, max(gc % 64 - 1, 0):gc % 64 + 2])
    # what does the reference say NOW for this agent?
    print("   ref again:", nav.desired_velocity(dest_ids[f], world["pos_xz"][b], world["flock_target_xz"][f]))
ctx.close(); pfref.RefMove.unload()

# ---- velocity mismatches
for clustered, n, k, blk in [(False, 1500, 4, False), (True, 1200, 3, False)]:
    grid = cases.synth.cost_grid(4, 4, seed=21)
    grid, nav = cases.ref_nav_for(4, 4, seed=21)
    world = cases.make_agents(grid, n, k, seed=31 + n, clustered=clustered)
    mv, dest_ids = cases.ref_move_for(nav, world)
    exp_vel = mv.velocity(None); vdes = mv.vdes()
    ctx = T._upload(navhip, nav); out = ctx.agent_step(T._step_arrays(world, mv, vdes)); ctx.close()
    moving = ~np.isin(world["state"], (2, 4))
    err = T._vel_err(out["vel_xz"], exp_vel)
    bad = np.flatnonzero(moving & ~(err <= 1e-4))
    psm = np.isin(world["state"], (0, 5, 6))
    vp = np.stack([mv.vpref(int(u), vdes[u]) if psm[u] else np.zeros(2, np.float32) for u in range(n)])
    verr = T._vel_err(out["vpref_xz"], vp)
    vbad = np.flatnonzero(psm & ~(verr <= 1e-4))
    print("world", clustered, n, "vel bad", bad, "vpref bad", vbad, "vpref exact frac", float((out["vpref_xz"][psm] == vp[psm]).all(1).mean()))
    for u in list(vbad[:4]):
        ar, co, se = mv.forces(int(u), vdes[u])
        print("  uid", u, "state", world["state"][u], "los", world["has_dest_los"][u], "pos", world["pos_xz"][u], "vel", world["vel_xz"][u])
        print("    ref arrive", ar, "coh", co, "sep", se, "vpref", vp[u], "got vpref", out["vpref_xz"][u])
        cnt, ids = pfref.spatial_query(navhip.grid_bounds(4, 4), world["pos_xz"], world["pos_xz"][u][None], 30.0, 128)
        print("    n30", cnt)
    for u in [b for b in bad if b not in vbad][:4]:
        dy, st = mv.neighbours(int(u))
        print("  uid", u, "vpref ok; exp vel", exp_vel[u], "got", out["vel_xz"][u], "ndyn", len(dy), "nstat", len(st), "vpref", vp[u])
    pfref.RefMove.unload()
