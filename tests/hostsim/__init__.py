"""tests/hostsim -- TEST INFRASTRUCTURE ONLY: the thread-per-agent bodies of the movement step that
run on the device (csrc/agent_thread.h: pool_record, mid_thread, post_thread) compiled with g++ and
driven serially -- behind serial statements of the group-parallel parts (serial_thread.h) -- to check
their logic against the reference on a machine without a GPU.  Never imported by the product."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "permafrost-engine_amd", "csrc")
LIB = os.path.join(HERE, "_hostsim.so")
_lib = None


def build():
    src = os.path.join(HERE, "hostsim.cpp")
    deps = [src] + [os.path.join(CSRC, f) for f in ("agent_thread.h", "agent_math.h", "agent_types.h", "map_view.h")]
    deps += [os.path.join(ROOT, "include", "navhip.h"), os.path.join(HERE, "serial_thread.h")]
    if os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    # same floating-point contract as the device build: no FMA contraction, IEEE everything
    cmd = ["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, src, "-o", LIB]
    subprocess.check_call(cmd)
    return LIB


class Map(C.Structure):
    _fields_ = [("chunk_w", C.c_int32), ("chunk_h", C.c_int32),
                ("cost", C.c_void_p * 12), ("blockers", C.c_void_p * 12)]


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


DISP_NAMES = {0: "done", 1: "row 1-2", 2: "row 3-4", 3: "row 5-8", 4: "row 9-16", 5: "wave", 6: "full"}


def agent_step(navhip, w, h, cost, blockers, arrays, coh_xz, hz=20, layer=0):
    """arrays: navhip_world member arrays (numpy).  Returns dict of outputs + 'disp' (per entity:
    DISP_* of agent_thread.h, +16 = the serial search found no admissible point: not computed here)."""
    world, keep = navhip.make_world(w, h, arrays, hz)
    n = world.n_ents
    m = Map()
    m.chunk_w, m.chunk_h = w, h
    cost = np.ascontiguousarray(cost, np.uint8)
    blockers = np.ascontiguousarray(blockers, np.uint16)
    m.cost[layer] = cost.ctypes.data
    m.blockers[layer] = blockers.ctypes.data
    out = {k: np.zeros((n, 2), np.float32) for k in ("vel_xz", "new_pos_xz", "vdes_xz", "vpref_xz")}
    out["status"] = np.zeros(n, np.uint8)
    so = navhip.StepOut()
    for k in out:
        setattr(so, k, out[k].ctypes.data)
    disp = np.zeros(n, np.uint8)
    counts = np.zeros((n, 2), np.int32)
    coh = np.ascontiguousarray(coh_xz, np.float32)
    rc = lib().hostsim_agent_step(C.byref(m), C.byref(world), coh.ctypes.data_as(C.c_void_p), C.byref(so),
                                  disp.ctypes.data_as(C.c_void_p), counts.ctypes.data_as(C.c_void_p))
    assert rc == 0
    out["disp"] = disp
    out["counts"] = counts
    return out


def clearpath_light(ent, des_v, dyn, n_dyn, stat, n_stat):
    ent = np.ascontiguousarray(ent, np.float32).reshape(-1, 5)
    nq = len(ent)
    des_v = np.ascontiguousarray(des_v, np.float32).reshape(nq, 2)
    dyn = np.ascontiguousarray(dyn, np.float32).reshape(nq, 32, 5)
    stat = np.ascontiguousarray(stat, np.float32).reshape(nq, 32, 5)
    n_dyn = np.ascontiguousarray(n_dyn, np.int32)
    n_stat = np.ascontiguousarray(n_stat, np.int32)
    out = np.zeros((nq, 2), np.float32)
    found = np.zeros(nq, np.int32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = lib().hostsim_clearpath_light(nq, p(ent), p(des_v), p(dyn), p(n_dyn), p(stat), p(n_stat), p(out), p(found))
    assert rc == 0
    return out, found


# ---- the group code itself (csrc/agent_group.h) on a lockstep emulator of a wave: wave_emu.h + group_sim.cpp ----
GROUP_LIB = os.path.join(HERE, "_groupsim.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"      # (agent_group.h uses clang vector types; the host compiler of the ROCm toolchain)
_glib = None


def group_available():
    return os.path.exists(CLANG)


def build_group():
    src = os.path.join(HERE, "group_sim.cpp")
    deps = [src, os.path.join(HERE, "wave_emu.h")] + [os.path.join(CSRC, f) for f in
            ("agent_group.h", "agent_thread.h", "agent_math.h", "agent_types.h", "map_view.h")]
    deps.append(os.path.join(ROOT, "include", "navhip.h"))
    if os.path.exists(GROUP_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(GROUP_LIB) for d in deps):
        return GROUP_LIB
    # -O1: the emulator orders lanes that wait at different operations by call-site address (wave_emu.h), which wants
    # the code laid out in source order; same floating-point contract as the device build
    cmd = [CLANG, "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-w",
           "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, src, "-o", GROUP_LIB]
    subprocess.check_call(cmd)
    return GROUP_LIB


def group_lib():
    global _glib
    if _glib is None:
        _glib = C.CDLL(build_group())
    return _glib


def clearpath_group(G, ent, des_v, dyn, n_dyn, stat, n_stat):
    """clearpath_grp<G> (G = 16: the row groups of k_cp_rows, at most 16 neighbours in total; G = 64: one wave per
    problem, k_cp_heavy / k_agent_full) for nq independent problems on the emulator.  Returns (velocities [nq][2],
    cross-lane operations executed)."""
    ent = np.ascontiguousarray(ent, np.float32).reshape(-1, 5)
    nq = len(ent)
    des_v = np.ascontiguousarray(des_v, np.float32).reshape(nq, 2)
    dyn = np.ascontiguousarray(dyn, np.float32).reshape(nq, 32, 5)
    stat = np.ascontiguousarray(stat, np.float32).reshape(nq, 32, 5)
    n_dyn = np.ascontiguousarray(n_dyn, np.int32)
    n_stat = np.ascontiguousarray(n_stat, np.int32)
    out = np.zeros((nq, 2), np.float32)
    ops = C.c_long(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = group_lib().groupsim_clearpath(int(G), nq, p(ent), p(des_v), p(dyn), p(n_dyn), p(stat), p(n_stat), p(out), C.byref(ops))
    if rc:
        raise RuntimeError("groupsim_clearpath failed (%d)" % rc)
    return out, ops.value


def group_attempts(reset=True):
    """[k] = problems that returned in attempt k (7: seven or more), [8] = attempts in total, since the last reset."""
    buf = (C.c_ulonglong * 9)()
    group_lib().groupsim_attempts(buf, 1 if reset else 0)
    return list(buf)


def group_agent_step(navhip, w, h, cost, blockers, arrays, coh_xz, hz=20, layer=0):
    """The velocity step of a snapshot through the device's own per-agent sources on the emulator: nbr_walk_row (16
    lanes), mid_thread, cp_load_lists + clearpath_grp (16 or 64 lanes, by neighbour count), post_thread.  Returns the
    outputs + 'disp' (DISP_* per entity; 7 = the irregular gather, not stepped) + 'ops' (cross-lane operations)."""
    world, keep = navhip.make_world(w, h, arrays, hz)
    n = world.n_ents
    m = Map()
    m.chunk_w, m.chunk_h = w, h
    cost = np.ascontiguousarray(cost, np.uint8)
    blockers = np.ascontiguousarray(blockers, np.uint16)
    m.cost[layer] = cost.ctypes.data
    m.blockers[layer] = blockers.ctypes.data
    out = {k: np.zeros((n, 2), np.float32) for k in ("vel_xz", "new_pos_xz", "vdes_xz", "vpref_xz")}
    out["status"] = np.zeros(n, np.uint8)
    so = navhip.StepOut()
    for k in out:
        setattr(so, k, out[k].ctypes.data)
    disp = np.zeros(n, np.uint8)
    coh = np.ascontiguousarray(coh_xz, np.float32)
    ops = C.c_long(0)
    rc = group_lib().groupsim_agent_step(C.byref(m), C.byref(world), coh.ctypes.data_as(C.c_void_p), C.byref(so),
                                         disp.ctypes.data_as(C.c_void_p), C.byref(ops))
    if rc:
        raise RuntimeError("groupsim_agent_step failed (%d)" % rc)
    out["disp"] = disp
    out["ops"] = ops.value
    return out


def clearpath_team(ent, des_v, dyn, n_dyn, stat, n_stat):
    """clearpath_grp<64, true>: one problem searched by a workgroup of four waves (k_cp_heavy below CP_SOLO_MIN
    problems; k_clearpath_team) on the emulator."""
    ent = np.ascontiguousarray(ent, np.float32).reshape(-1, 5)
    nq = len(ent)
    des_v = np.ascontiguousarray(des_v, np.float32).reshape(nq, 2)
    dyn = np.ascontiguousarray(dyn, np.float32).reshape(nq, 32, 5)
    stat = np.ascontiguousarray(stat, np.float32).reshape(nq, 32, 5)
    n_dyn = np.ascontiguousarray(n_dyn, np.int32)
    n_stat = np.ascontiguousarray(n_stat, np.int32)
    out = np.zeros((nq, 2), np.float32)
    ops = C.c_long(0)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    rc = group_lib().groupsim_clearpath_team(nq, p(ent), p(des_v), p(dyn), p(n_dyn), p(stat), p(n_stat), p(out), C.byref(ops))
    if rc:
        raise RuntimeError("groupsim_clearpath_team failed (%d)" % rc)
    return out, ops.value


# ---- the WHOLE library on the emulator: every translation unit of csrc/ compiled for the host against fakehip/ ----------
EMU_LIB = os.path.join(HERE, "_navhip_emu.so")
EMU_SOURCES = ["navhip_api", "pool_api", "field_kernels", "agent_kernels", "blocker_kernels", "los_kernels",
               "region_kernels", "comm_api", "state_kernels", "tick_api", "stream_set"]
# the two statements of the device sources a host compiler cannot take (a register clobber that pins the allocation of
# k_cp_rows; an unsized extern array for dynamic LDS) -- replaced in the copies that are compiled, nothing else is
_EMU_PATCHES = [
    ('    asm volatile("" ::: "v127");', "    /* (register pin: device only) */"),
    ("    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];",
     "    static __attribute__((aligned(16))) uint8_t smem[160 * 1024];"),
]


def build_navhip_emu():
    """tests/hostsim/_navhip_emu.so: libnavhip's own sources -- API layer and kernels -- built for the host on top of
    the lockstep emulator (wave_emu.h) and a stand-in for the HIP runtime (fakehip/: device memory is host memory,
    a launch runs the grid block by block).  Exports the C ABI of include/navhip.h.  TEST INFRASTRUCTURE: loaded only
    by tests that name it through NAVHIP_LIB; the product library has no CPU path."""
    src_dir = os.path.join(HERE, "_emu_src")
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    deps += [os.path.join(HERE, "wave_emu.h"), os.path.join(HERE, "fakehip", "hip", "hip_runtime.h"),
             os.path.join(ROOT, "include", "navhip.h"), os.path.abspath(__file__)]
    if os.path.exists(EMU_LIB) and all(os.path.getmtime(d) <= os.path.getmtime(EMU_LIB) for d in deps):
        return EMU_LIB
    os.makedirs(src_dir, exist_ok=True)
    objs = []
    for name in EMU_SOURCES:
        text = open(os.path.join(CSRC, name + ".hip")).read()
        for old, new in _EMU_PATCHES:
            text = text.replace(old, new)
        cpp = os.path.join(src_dir, name + ".cpp")
        open(cpp, "w").write(text)
        obj = os.path.join(src_dir, name + ".o")
        subprocess.check_call([CLANG, "-O1", "-std=c++17", "-fPIC", "-DNH_HOSTSIM=1", "-DNH_HOSTSIM_PROBEMASK=1", "-ffp-contract=off", "-fno-fast-math",
                               "-w", "-I" + os.path.join(HERE, "fakehip"), "-I" + HERE, "-I" + os.path.join(ROOT, "include"),
                               "-I" + CSRC, "-c", cpp, "-o", obj])
        objs.append(obj)
    subprocess.check_call([CLANG, "-shared", "-fPIC", "-o", EMU_LIB] + objs + ["-ldl"])
    return EMU_LIB

