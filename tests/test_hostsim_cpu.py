"""The thread-per-agent bodies of the movement step (csrc/agent_thread.h) compiled for the host
(tests/hostsim) against the reference build: visiting order and caps of the combined neighbour walk,
separation sums, the priority ladder, admissibility, the sequential ClearPath search and the
position accept -- everything in them that is logic rather than a GPU intrinsic.  Runs without a GPU;
the same source is what the device kernels call (the -m gpu tests check those end to end)."""
import numpy as np
import pytest

from oracle import pfref
from tests import cases, hostsim

pytestmark = pytest.mark.skipif(not pfref.available(), reason="oracle/_ref (the reference build) not present")


@pytest.fixture(scope="module")
def navlib():
    from permafrost_engine_amd import navhip
    return navhip


@pytest.mark.parametrize("seed,max_dyn,max_stat,spread", [(1, 2, 2, 9.0), (2, 4, 0, 6.0), (3, 0, 4, 5.0),
                                                         (4, 3, 1, 2.5), (5, 1, 1, 4.0), (6, 8, 8, 9.0),
                                                         (7, 32, 32, 9.5)])
def test_serial_clearpath_search_matches_reference(seed, max_dyn, max_stat, spread):
    nq = 600 if max_dyn < 32 else 80
    ent, des, dyn, nd, stat, ns = cases.cp_problems(seed, nq, max_dyn, max_stat, spread)
    got, found = hostsim.clearpath_light(ent, des, dyn, nd, stat, ns)
    n_checked = 0
    for i in range(nq):
        if not found[i]:
            continue            # the step hands these to the wave path (remove_furthest + retry)
        exp = pfref.clearpath_new_velocity(ent[i], des[i], dyn[i, :nd[i]], stat[i, :ns[i]])
        both_nan = np.isnan(exp) & np.isnan(got[i])
        assert np.array_equal(np.where(both_nan, 0, got[i]).view(np.uint32),
                              np.where(both_nan, 0, exp).view(np.uint32)), (i, nd[i], ns[i], got[i], exp)
        n_checked += 1
    assert n_checked > nq * (0.8 if max_dyn < 32 else 0.3)


DISP_FULL = 7      # agent_thread.h: DISP_DONE, ROW0..3, WAVE, HEAVY, FULL (the irregular gather)

def _world(navlib, clustered, n, k, blk, seed=21, garrison=False, arrival=False):
    grid = cases.synth.cost_grid(4, 4, seed=seed)
    blockers = cases.random_blockers(grid, seed=8, frac=0.02) if blk else None
    grid, nav = cases.ref_nav_for(4, 4, seed=seed, blockers=blockers)
    world = cases.make_agents(grid, n, k, seed=31 + n, clustered=clustered)
    if garrison:
        g = np.random.RandomState(5).rand(n) < 0.02
        world["flags"] = np.where(g, world["flags"] | navlib.ENTITY_FLAG_GARRISONED, world["flags"]).astype(np.uint32)
    return grid, nav, world


@pytest.mark.parametrize("clustered,n,k,blk,garrison", [(False, 1500, 4, False, False), (True, 1200, 3, False, False),
                                                        (True, 1500, 2, True, False), (False, 1500, 4, False, True)])
def test_thread_step_matches_reference(navlib, clustered, n, k, blk, garrison):
    grid, nav, world = _world(navlib, clustered, n, k, blk, garrison=garrison)
    mv, dest_ids = cases.ref_move_for(nav, world)
    exp_vel = mv.velocity(None)
    vdes = mv.vdes()
    order = [mv.flock_order(f) for f in range(k)]
    a = cases.step_arrays(world, vdes, order)
    moving = ~np.isin(world["state"], (2, 4))
    coh = np.zeros((n, 2), np.float32)
    for uid in np.flatnonzero(np.isin(world["state"], (0, 5, 6))):
        coh[uid] = mv.forces(int(uid), vdes[uid])[1]
    out = hostsim.agent_step(navlib, 4, 4, nav.plane(0), nav.plane(1), a, coh)
    disp = out["disp"]
    computed = moving & (disp < DISP_FULL)
    assert computed.sum() > 0.6 * moving.sum(), np.bincount(disp[moving])
    # ClearPath neighbour lists (counts) against find_neighbours
    for uid in np.flatnonzero(moving & (disp != DISP_FULL))[:300]:
        dyn, stat = mv.neighbours(int(uid))
        assert (len(dyn), len(stat)) == tuple(out["counts"][uid]), uid
    # preferred velocity of the point-seek agents, then the final velocities, bit for bit
    ps = np.flatnonzero(np.isin(world["state"], (0, 5, 6)) & (disp != DISP_FULL))
    for uid in ps[:80]:
        ev = mv.vpref(int(uid), vdes[uid])
        assert np.array_equal(out["vpref_xz"][uid].view(np.uint32), ev.view(np.uint32)), ("vpref", uid)
    bad = np.flatnonzero(computed & ~(out["vel_xz"].view(np.uint32) == exp_vel.view(np.uint32)).all(1))
    assert len(bad) == 0, (bad[:10], disp[bad[:10]], out["vel_xz"][bad[:3]], exp_vel[bad[:3]])
    assert np.all(out["vel_xz"][~moving] == 0)
    if garrison:
        assert (disp[moving] == DISP_FULL).sum() > 0          # garrisoned neighbours -> the irregular list
    # position accept
    for uid in np.flatnonzero(computed)[:200]:
        v = exp_vel[uid]
        npos = world["pos_xz"][uid] + v
        on_blocked = nav.position_blocked(world["pos_xz"][uid])
        acc = (np.linalg.norm(v) > 0) and nav.position_pathable(npos) and (on_blocked or not nav.position_blocked(npos))
        if world["flags"][uid] & navlib.ENTITY_FLAG_GARRISONED:
            acc = False
        assert bool(out["status"][uid] & 1) == bool(acc), uid
    pfref.RefMove.unload()


def test_thread_step_with_arrival_state_matches_reference(navlib):
    """G_Arrival_SeekTarget / G_Arrival_NeighbourSettling (the reference's own arrival.c in the harness):
    committed units seek their slot, settling neighbours are static obstacles."""
    n, k = 1500, 4
    grid, nav, world = _world(navlib, False, n, k, False)
    sink, aflags = cases.arrival_inputs(world, seed=3)
    mv, dest_ids = cases.ref_move_for(nav, world)
    base_vel = mv.velocity(None)
    vdes = mv.vdes()
    mv.set_arrival(sink, aflags)
    exp_vel = mv.velocity(vdes)
    moving = ~np.isin(world["state"], (2, 4))
    assert (exp_vel[moving] != base_vel[moving]).any(1).sum() > 50      # the inputs matter
    a = cases.step_arrays(world, vdes, [mv.flock_order(f) for f in range(k)])
    a["arrival_sink_xz"], a["arrival_flags"] = sink, aflags
    coh = np.zeros((n, 2), np.float32)
    for uid in np.flatnonzero(np.isin(world["state"], (0, 5, 6))):
        coh[uid] = mv.forces(int(uid), vdes[uid])[1]
    out = hostsim.agent_step(navlib, 4, 4, nav.plane(0), nav.plane(1), a, coh)
    computed = moving & (out["disp"] < DISP_FULL)
    assert computed.sum() > 0.6 * moving.sum()
    for uid in np.flatnonzero(moving)[:300]:
        dyn, stat = mv.neighbours(int(uid))
        assert (len(dyn), len(stat)) == tuple(out["counts"][uid]), uid
    bad = np.flatnonzero(computed & ~(out["vel_xz"].view(np.uint32) == exp_vel.view(np.uint32)).all(1))
    assert len(bad) == 0, (bad[:10], out["vel_xz"][bad[:3]], exp_vel[bad[:3]])
    pfref.RefMove.unload()


def _cones_around(rng, n, spread):
    """n ClearPath cones as the search stores them (apex, slopes, unit side rays) around entities near `spread`."""
    apex = rng.uniform(-spread, spread, (n, 2)).astype(np.float32)
    ang = rng.uniform(0, 2 * np.pi, n)
    half = rng.uniform(0.02, 1.2, n)
    left = np.stack([np.cos(ang + half), np.sin(ang + half)], 1).astype(np.float32)
    right = np.stack([np.cos(ang - half), np.sin(ang - half)], 1).astype(np.float32)
    with np.errstate(divide="ignore", invalid="ignore"):
        sl = np.where(np.abs(left[:, 0]) < 1 / 1024, np.nan, left[:, 1] / left[:, 0]).astype(np.float32)
        sr = np.where(np.abs(right[:, 0]) < 1 / 1024, np.nan, right[:, 1] / right[:, 0]).astype(np.float32)
    return apex, left, right, sl, sr


@pytest.mark.parametrize("spread", [3.0, 40.0, 3000.0])
def test_branch_free_ray_intersection_equals_the_branching_one(spread):
    """ray_isect_bf (agent_math.h; C_RayRayIntersection2D collision.c:854 + the distance to des_v, all three shapes of
    C_InfiniteLineIntersection evaluated and selected): wherever it does not hand the lane back (`slow`), validity,
    point and distance are bit for bit those of ray_isect + vlen.  Axis-parallel and coincident rays included."""
    import ctypes as C
    rng = np.random.RandomState(int(spread))
    n = 200000
    apex, left, right, sl, sr = _cones_around(rng, 2 * n, spread)
    # a share of exactly vertical / horizontal rays and of rays from a common apex (static neighbours)
    k = n // 10
    left[:k] = [0.0, 1.0]; sl[:k] = np.nan
    left[k:2 * k] = [1.0, 0.0]; sl[k:2 * k] = 0.0
    apex[n:n + 3 * k] = apex[:3 * k]
    rays = np.zeros((n, 10), np.float32)
    rays[:, 0:2], rays[:, 2:4], rays[:, 4] = apex[:n], left[:n], sl[:n]
    rays[:, 5:7], rays[:, 7:9], rays[:, 9] = apex[n:], right[n:], sr[n:]
    des = rng.uniform(-1, 1, 2).astype(np.float32)
    ent = rng.uniform(-spread, spread, 2).astype(np.float32)
    bad = C.c_int(0)
    L = hostsim.lib()
    decided = L.hostsim_ray_isect_bf_check(n, rays.ctypes.data_as(C.c_void_p), des.ctypes.data_as(C.c_void_p),
                                           ent.ctypes.data_as(C.c_void_p), C.byref(bad))
    assert bad.value == 0, bad.value
    assert decided > 0.7 * n, decided           # (the expansions are the exception)


@pytest.mark.parametrize("spread", [3.0, 3000.0])
def test_branch_free_cone_test_equals_the_branching_one(spread):
    """cone_test_bf == cone_contains_fast verdict for verdict (0 outside, 1 inside, 2 undecided), and every decided
    verdict == cone_contains_exact (inside_pcr's test, clearpath.c:249-262); apex points and points on the side rays
    included."""
    import ctypes as C
    rng = np.random.RandomState(7 + int(spread))
    n = 300000
    apex, left, right, sl, sr = _cones_around(rng, n, spread)
    cones = np.concatenate([apex, sl[:, None], sr[:, None], left, right], 1).astype(np.float32)
    pts = (apex + rng.normal(0, 2.0, (n, 2))).astype(np.float32)
    k = n // 10
    pts[:k] = apex[:k]                                                    # the apex itself: "not inside" (:262)
    t = rng.uniform(0.001, 5.0, (k, 1)).astype(np.float32)
    pts[k:2 * k] = apex[k:2 * k] + left[k:2 * k] * t                      # on the left side ray
    pts[2 * k:3 * k] = apex[2 * k:3 * k] + right[2 * k:3 * k] * t         # on the right side ray
    pts[3 * k:4 * k] = apex[3 * k:4 * k] + rng.normal(0, 1e-3, (k, 2)).astype(np.float32)   # within EPS of the apex
    bad, bad_pair = C.c_int(0), C.c_int(0)
    decided = hostsim.lib().hostsim_cone_test_bf_check(n, cones.ctypes.data_as(C.c_void_p), pts.ctypes.data_as(C.c_void_p),
                                                       C.byref(bad), C.byref(bad_pair))
    assert bad_pair.value == 0 and bad.value == 0, (bad_pair.value, bad.value)
    assert decided > 0.6 * n, decided


# ---- the group code itself (csrc/agent_group.h) on the lockstep wave emulator (tests/hostsim/wave_emu.h) ------------
group_sim = pytest.mark.skipif(not hostsim.group_available(), reason="no clang++ (ROCm LLVM) to compile agent_group.h for the host")


def _same(got, exp):
    both = np.isnan(exp) & np.isnan(got)
    return np.array_equal(np.where(both, 0, got).view(np.uint32), np.where(both, 0, exp).view(np.uint32))


@group_sim
@pytest.mark.parametrize("G,seed,max_dyn,max_stat,spread,nq", [
    (16, 1, 4, 4, 6.0, 300), (16, 2, 8, 8, 9.0, 300), (16, 5, 16, 0, 4.0, 200), (16, 9, 8, 8, 2.0, 200),
    (64, 3, 8, 8, 9.0, 150), (64, 6, 16, 16, 9.0, 80), (64, 7, 32, 32, 9.5, 40), (64, 8, 32, 32, 3.0, 30)])
def test_group_clearpath_search_on_the_wave_emulator_matches_reference(G, seed, max_dyn, max_stat, spread, nq):
    """clearpath_grp<G> -- the source k_cp_rows (G = 16) and k_cp_heavy / k_agent_full (G = 64, a wave per problem)
    execute: cones, ranks, projections, the lane = row column phase, the branch-free queue, cone compaction, the retry
    shortcut and the replay of removals -- run lane by lane in lockstep on the host, == G_ClearPath_NewVelocity
    (clearpath.c:694) of the reference build, bit for bit, on every problem."""
    ent, des, dyn, nd, stat, ns = cases.cp_problems(seed, nq, max_dyn, max_stat, spread)
    if G == 16:
        keep = (nd + ns) <= 16
        ent, des, dyn, nd, stat, ns = [a[keep] for a in (ent, des, dyn, nd, stat, ns)]
    hostsim.group_attempts(reset=True)
    got, ops = hostsim.clearpath_group(G, ent, des, dyn, nd, stat, ns)
    assert ops > 20 * len(ent)                       # (the lanes really met at cross-lane operations)
    for i in range(len(ent)):
        exp = pfref.clearpath_new_velocity(ent[i], des[i], dyn[i, :nd[i]], stat[i, :ns[i]])
        assert _same(got[i], exp), (i, nd[i], ns[i], got[i], exp)


@group_sim
def test_group_search_takes_the_retry_shortcut_on_the_emulator():
    """Dense problems whose first attempt fails: the removal schedule (cp_jump), the replay and the attempt that
    succeeds, wave-wide -- and still the reference's answer."""
    ent, des, dyn, nd, stat, ns = cases.cp_problems(8, 60, 32, 32, 2.0)
    hostsim.group_attempts(reset=True)
    got, _ = hostsim.clearpath_group(64, ent, des, dyn, nd, stat, ns)
    att = hostsim.group_attempts()
    for i in range(len(ent)):
        exp = pfref.clearpath_new_velocity(ent[i], des[i], dyn[i, :nd[i]], stat[i, :ns[i]])
        assert _same(got[i], exp), (i, nd[i], ns[i], got[i], exp)
    # att[k], k >= 1: problems that returned in the attempt behind the shortcut; att[0]: no attempt succeeds at all
    assert sum(att[1:8]) >= 20 and att[0] >= 1, att


@group_sim
@pytest.mark.parametrize("clustered,n,k,blk", [(False, 1500, 4, False), (True, 1200, 3, False), (True, 1500, 2, True)])
def test_group_step_on_the_wave_emulator_matches_reference(navlib, clustered, n, k, blk):
    """The whole velocity step of a snapshot through the device's own per-agent sources on the host emulator --
    nbr_walk_row on 16 lanes (k_agent_nbr), mid_thread (k_agent_mid), cp_load_lists + clearpath_grp on 16 or 64 lanes
    with the retry logic (k_cp_small / k_cp_rows / k_cp_heavy), post_thread -- == move_velocity_work
    (movement.c:3395) of the reference build for every agent the regular path steps, clustered worlds (neighbour caps
    bind, searches retry) and blockers included."""
    grid, nav, world = _world(navlib, clustered, n, k, blk)
    mv, dest_ids = cases.ref_move_for(nav, world)
    exp_vel = mv.velocity(None)
    vdes = mv.vdes()
    order = [mv.flock_order(f) for f in range(k)]
    a = cases.step_arrays(world, vdes, order)
    moving = ~np.isin(world["state"], (2, 4))
    coh = np.zeros((n, 2), np.float32)
    for uid in np.flatnonzero(np.isin(world["state"], (0, 5, 6))):
        coh[uid] = mv.forces(int(uid), vdes[uid])[1]
    hostsim.group_attempts(reset=True)
    out = hostsim.group_agent_step(navlib, 4, 4, nav.plane(0), nav.plane(1), a, coh)
    disp = out["disp"]
    stepped = moving & (disp != DISP_FULL)
    assert stepped.sum() > 0.9 * moving.sum(), np.bincount(disp[moving])
    bad = np.flatnonzero(stepped & ~(out["vel_xz"].view(np.uint32) == exp_vel.view(np.uint32)).all(1))
    assert len(bad) == 0, (bad[:10], disp[bad[:10]], out["vel_xz"][bad[:3]], exp_vel[bad[:3]])
    assert np.all(out["vel_xz"][~moving] == 0)
    if clustered:
        assert (disp[moving] >= 5).sum() > 50        # wave-wide searches (17+ neighbours) among them
        assert sum(hostsim.group_attempts()[1:8]) > 0 # ... and searches that needed the retry shortcut
    pfref.RefMove.unload()


@group_sim
@pytest.mark.parametrize("seed,max_dyn,max_stat,spread,nq", [(3, 8, 8, 9.0, 60), (6, 16, 16, 9.0, 40), (7, 32, 32, 9.5, 25),
                                                             (8, 32, 32, 2.0, 25)])
def test_team_search_on_the_workgroup_emulator_matches_reference(seed, max_dyn, max_stat, spread, nq):
    """clearpath_grp<64, true>: four waves search one problem -- every wave its share of the columns and of the retry
    shortcut's candidates, the minima combined through LDS behind workgroup barriers (team_min) -- as k_cp_heavy runs
    the 17-64-neighbour problems outside a jam.  Four emulated waves + __syncthreads == the reference."""
    ent, des, dyn, nd, stat, ns = cases.cp_problems(seed, nq, max_dyn, max_stat, spread)
    hostsim.group_attempts(reset=True)
    got, ops = hostsim.clearpath_team(ent, des, dyn, nd, stat, ns)
    for i in range(len(ent)):
        exp = pfref.clearpath_new_velocity(ent[i], des[i], dyn[i, :nd[i]], stat[i, :ns[i]])
        assert _same(got[i], exp), (i, nd[i], ns[i], got[i], exp)
    if spread < 3.0:
        assert sum(hostsim.group_attempts()[1:8]) > 0      # the shared retry shortcut ran



@pytest.mark.skipif(not hostsim.group_available(), reason="no clang++ (ROCm LLVM) for the host build")
def test_emulated_runtime_rejects_copies_a_gpu_would_reject(tmp_path):
    """On the emulator device memory IS host memory: a copy with the wrong direction flag, or one that runs past the end
    of a device allocation, would just work.  The stand-in runtime (tests/hostsim/fakehip) remembers what hipMalloc
    handed out and fails such a copy the way a GPU does -- so the emulated runs of the GPU suite also check that."""
    import os
    import subprocess
    src = tmp_path / "copies.cpp"
    src.write_text(r'''
#include <hip/hip_runtime.h>
int main() {
    char host[256] = {0};
    char *dev = nullptr, *dev2 = nullptr;
    if(hipMalloc((void**)&dev, 128) || hipMalloc((void**)&dev2, 128)) return 10;
    int bad = 0;
    if(hipMemcpy(dev, host, 128, hipMemcpyHostToDevice) != hipSuccess) bad |= 1;                    /* fine */
    if(hipMemcpyAsync(host, dev + 64, 64, hipMemcpyDeviceToHost, 0) != hipSuccess) bad |= 2;         /* fine */
    if(hipMemcpy(dev2, dev, 128, hipMemcpyDeviceToDevice) != hipSuccess) bad |= 4;                   /* fine */
    if(hipMemcpy(host, dev, 128, hipMemcpyHostToDevice) == hipSuccess) bad |= 8;                     /* source is device memory */
    if(hipMemcpy(dev, host, 128, hipMemcpyDeviceToHost) == hipSuccess) bad |= 16;                    /* destination is device memory */
    if(hipMemcpyAsync(dev + 64, host, 128, hipMemcpyHostToDevice, 0) == hipSuccess) bad |= 32;       /* past the end */
    if(hipMemcpy(host, dev + 100, 64, hipMemcpyDeviceToHost) == hipSuccess) bad |= 64;               /* past the end */
    if(hipFree(dev) != hipSuccess || hipFree(dev2) != hipSuccess) bad |= 128;
    if(hipFree(host) == hipSuccess) bad |= 256;                                                      /* not a device allocation */
    return bad;
}''')
    exe = tmp_path / "copies"
    here = os.path.dirname(os.path.abspath(hostsim.__file__))
    subprocess.check_call([hostsim.CLANG, "-O1", "-std=c++17", "-w", "-DNH_HOSTSIM=1", "-I" + os.path.join(here, "fakehip"), "-I" + here,
                           "-I" + os.path.join(os.path.dirname(os.path.dirname(here)), "include"),
                           "-I" + os.path.join(os.path.dirname(os.path.dirname(here)), "permafrost-engine_amd", "csrc"),
                           str(src), "-o", str(exe)])
    r = subprocess.run([str(exe)], stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, (r.returncode, r.stderr)
    assert r.stderr.count("emulated HIP runtime:") == 5
