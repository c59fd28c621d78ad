"""oracle/navoracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/libnavoracle.so, the plain-C restatement of the reference's algorithm
for the navigation hot path (oracle/navoracle.c).  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module.  It mirrors the PODs of include/navhip.h
on its own (it does not import the product package).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
NLAYERS = 12

# navhip_field_req, include/navhip.h
FIELD_REQ_DTYPE = np.dtype([
    ("layer", np.uint8), ("type", np.uint8), ("faction_id", np.uint8), ("flags", np.uint8),
    ("enemies", np.uint16), ("chunk_r", np.uint16), ("chunk_c", np.uint16),
    ("tile_r", np.uint8), ("tile_c", np.uint8),
    ("port_r0", np.uint8), ("port_c0", np.uint8), ("port_r1", np.uint8), ("port_c1", np.uint8),
    ("next_r0", np.uint8), ("next_c0", np.uint8), ("next_r1", np.uint8), ("next_c1", np.uint8),
    ("next_chunk_r", np.uint16), ("next_chunk_c", np.uint16),
    ("port_iid", np.uint16), ("next_iid", np.uint16), ("aux_iid", np.uint16), ("_pad", np.uint16),
], align=False)


CIRCLE_DTYPE = np.dtype([("x", np.float32), ("z", np.float32), ("radius", np.float32),
                         ("faction_id", np.int32), ("flags", np.uint32), ("delta", np.int32)])


REGION_REQ_DTYPE = np.dtype([("layer", np.uint8), ("out_mode", np.uint8), ("enemies", np.uint16),
                             ("base_abs_r", np.int16), ("base_abs_c", np.int16),
                             ("rdim", np.uint16), ("cdim", np.uint16), ("roff", np.uint16),
                             ("coff", np.uint16), ("seed_begin", np.uint32), ("seed_count", np.uint32),
                             ("overlay_begin", np.uint32), ("overlay_count", np.uint32)])
LOS_REQ_DTYPE = np.dtype([("layer", np.uint8), ("faction_id", np.uint8), ("enemies", np.uint16),
                          ("chunk_r", np.uint16), ("chunk_c", np.uint16),
                          ("target_chunk_r", np.uint16), ("target_chunk_c", np.uint16),
                          ("target_tile_r", np.uint8), ("target_tile_c", np.uint8),
                          ("prev_dr", np.int8), ("prev_dc", np.int8)])


class Map(C.Structure):
    _fields_ = [("w", C.c_int32), ("h", C.c_int32),
                ("cost", C.c_void_p * NLAYERS), ("blockers", C.c_void_p * NLAYERS),
                ("local_islands", C.c_void_p * NLAYERS), ("factions", C.c_void_p * NLAYERS),
                ("islands", C.c_void_p * NLAYERS)]


class World(C.Structure):
    """navhip_world, include/navhip.h"""
    _fields_ = [
        ("n_ents", C.c_int32), ("n_flocks", C.c_int32), ("hz", C.c_int32),
        ("n_field_slots", C.c_int32),
        ("pos_xz", C.c_void_p), ("vel_xz", C.c_void_p), ("radius", C.c_void_p),
        ("max_speed", C.c_void_p), ("speed", C.c_void_p), ("flags", C.c_void_p),
        ("state", C.c_void_p), ("has_dest_los", C.c_void_p), ("flock", C.c_void_p),
        ("vdes_xz", C.c_void_p), ("flock_target_xz", C.c_void_p), ("flock_offsets", C.c_void_p),
        ("flock_members", C.c_void_p), ("flock_field_slot", C.c_void_p), ("field_pool", C.c_void_p),
        ("map_pos_x", C.c_float), ("map_pos_z", C.c_float),
        ("grid_xmin", C.c_float), ("grid_xmax", C.c_float), ("grid_zmin", C.c_float),
        ("grid_zmax", C.c_float), ("work_begin", C.c_int32), ("work_end", C.c_int32),
        ("form_ready", C.c_void_p), ("cell_pos_xz", C.c_void_p), ("form_cohesion_xz", C.c_void_p),
        ("form_align_xz", C.c_void_p), ("form_drag_xz", C.c_void_p),
        # fine-arrival inputs: the restatement does not model them (always NULL here; the arrival
        # paths are checked against the reference build itself)
        ("arrival_sink_xz", C.c_void_p), ("arrival_flags", C.c_void_p),
        ("los_pool", C.c_void_p), ("flock_los_slot", C.c_void_p), ("los_pos_xz", C.c_void_p),
        ("n_los_slots", C.c_int32), ("static_epoch", C.c_uint32)]


class StepOut(C.Structure):
    _fields_ = [("vel_xz", C.c_void_p), ("new_pos_xz", C.c_void_p), ("vdes_xz", C.c_void_p),
                ("vpref_xz", C.c_void_p), ("status", C.c_void_p)]


_WORLD_ARRAYS = (
    ("pos_xz", np.float32), ("vel_xz", np.float32), ("radius", np.float32),
    ("max_speed", np.float32), ("speed", np.float32), ("flags", np.uint32), ("state", np.uint8),
    ("has_dest_los", np.uint8), ("flock", np.int32), ("vdes_xz", np.float32),
    ("flock_target_xz", np.float32), ("flock_offsets", np.int32), ("flock_members", np.int32),
    ("flock_field_slot", np.int32), ("field_pool", np.uint8), ("form_ready", np.uint8),
    ("cell_pos_xz", np.float32), ("form_cohesion_xz", np.float32), ("form_align_xz", np.float32),
    ("form_drag_xz", np.float32))


def available():
    return os.path.exists(os.path.join(_HERE, "libnavoracle.so")) or os.path.exists(
        os.path.join(_HERE, "navoracle.c"))


def lib():
    global _LIB
    if _LIB is None:
        from oracle import build_oracle
        L = C.CDLL(build_oracle.build())
        L.no_field_update.argtypes = [C.POINTER(Map), C.c_void_p, C.c_void_p, C.c_void_p]
        L.no_build_fields.argtypes = [C.POINTER(Map), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.no_spatial_query.argtypes = [C.POINTER(World), C.c_void_p, C.c_int, C.c_float, C.c_int,
                                       C.c_void_p, C.c_void_p]
        L.no_clearpath.argtypes = [C.c_int] + [C.c_void_p] * 7
        L.no_agent_step.argtypes = [C.POINTER(Map), C.POINTER(World), C.POINTER(StepOut), C.c_int]
        L.no_agent_forces.argtypes = [C.POINTER(Map), C.POINTER(World), C.c_int, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
        L.no_blockers_circles.argtypes = [C.POINTER(Map), C.c_void_p, C.c_int, C.c_float, C.c_float,
                                          C.c_void_p]
        L.no_local_islands.argtypes = [C.POINTER(Map), C.c_int, C.c_void_p, C.c_void_p]
        L.no_build_region_fields.argtypes = [C.POINTER(Map), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                             C.c_void_p, C.c_size_t]
        L.no_build_los.argtypes = [C.POINTER(Map), C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_float, C.c_float]
        L.no_field_bench.restype = C.c_double
        L.no_field_bench.argtypes = [C.POINTER(Map), C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.no_agent_bench.restype = C.c_double
        L.no_agent_bench.argtypes = [C.POINTER(Map), C.POINTER(World), C.POINTER(StepOut), C.c_int]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def make_world(chunk_w, chunk_h, arrays, hz=20, work=None):
    w = World()
    keep = {}
    w.n_ents = len(arrays["pos_xz"])
    ft = arrays.get("flock_target_xz")
    w.n_flocks = 0 if ft is None else len(ft)
    w.hz = hz
    for name, dt in _WORLD_ARRAYS:
        a = arrays.get(name)
        if a is None:
            setattr(w, name, None)
            continue
        a = np.ascontiguousarray(a, dtype=dt)
        keep[name] = a
        setattr(w, name, a.ctypes.data)
    fp = arrays.get("field_pool")
    w.n_field_slots = 0 if fp is None else int(np.asarray(fp).shape[0])
    w.map_pos_x = chunk_w * 128.0
    w.map_pos_z = -chunk_h * 128.0
    w.grid_xmin, w.grid_xmax = -chunk_w * 128.0, chunk_w * 128.0
    w.grid_zmin, w.grid_zmax = -chunk_h * 128.0, chunk_h * 128.0
    if work is not None:
        w.work_begin, w.work_end = work
    return w, keep


class OracleNav:
    """The planes of one map ([h][w][64][64] per layer) + the restated algorithms over them."""

    def __init__(self, cost, blockers=None, local_islands=None, factions=None, layer=0):
        cost = np.ascontiguousarray(cost, np.uint8)
        assert cost.ndim == 4 and cost.shape[2:] == (64, 64)
        self.h, self.w = cost.shape[:2]
        self._planes = {}
        self._map = Map()
        self._map.w, self._map.h = self.w, self.h
        self.set_layer(layer, cost, blockers, local_islands, factions)

    def set_layer(self, layer, cost=None, blockers=None, local_islands=None, factions=None,
                  islands=None):
        for name, arr, dt in (("cost", cost, np.uint8), ("blockers", blockers, np.uint16),
                              ("local_islands", local_islands, np.uint16),
                              ("factions", factions, np.uint8), ("islands", islands, np.uint16)):
            if arr is None:
                continue
            a = np.ascontiguousarray(arr, dt)
            self._planes[(layer, name)] = a
            getattr(self._map, name)[layer] = a.ctypes.data

    def plane(self, layer, name):
        return self._planes[(layer, name)]

    # -- dynamic obstacles ----------------------------------------------------------------
    def blockers_circles(self, circles):
        """N_BlockersIncref / N_BlockersDecref on this object's (mutable) blockers / factions
        planes; returns the reference-style dirty flags [12][h][w] (tile toggled occupied/free)."""
        c = np.ascontiguousarray(circles, CIRCLE_DTYPE)
        dirty = np.zeros((NLAYERS, self.h, self.w), np.uint8)
        lib().no_blockers_circles(C.byref(self._map), _p(c), len(c), self.w * 128.0, -self.h * 128.0,
                                  _p(dirty))
        return dirty

    def local_islands(self, layer=0, only=None, into=None):
        """n_update_local_island_field over the current cost + blockers planes."""
        out = np.zeros((self.h, self.w, 64, 64), np.uint16) if into is None else into
        o = None if only is None else np.ascontiguousarray(only, np.uint8)
        rc = lib().no_local_islands(C.byref(self._map), layer, _p(out), _p(o) if o is not None else None)
        if rc:
            raise ValueError("no_local_islands: layer has no cost plane")
        return out

    # -- fields ---------------------------------------------------------------------------
    def build_fields(self, reqs, inout=None, want_integ=False):
        reqs = np.ascontiguousarray(reqs, FIELD_REQ_DTYPE)
        n = len(reqs)
        dirs = np.zeros((n, 64, 64), np.uint8)
        if inout is not None:
            dirs[...] = np.asarray(inout, np.uint8).reshape(n, 64, 64)
        integ = np.zeros((n, 64, 64), np.float32) if want_integ else None
        rc = lib().no_build_fields(C.byref(self._map), _p(reqs), n, _p(dirs),
                                   _p(integ) if want_integ else None)
        if rc:
            raise ValueError("no_build_fields: malformed request")
        return dirs, integ

    def build_region_fields(self, reqs, seeds, overlay=None, inout=None, out_stride=8192):
        reqs = np.ascontiguousarray(reqs, REGION_REQ_DTYPE)
        n = len(reqs)
        seeds = np.ascontiguousarray(seeds, np.int16).reshape(-1, 2)
        ov = np.zeros((1, 2), np.int16) if overlay is None or len(overlay) == 0 else \
            np.ascontiguousarray(overlay, np.int16).reshape(-1, 2)
        buf = np.zeros((n, out_stride), np.uint8)
        if inout is not None:
            a = np.asarray(inout, np.uint8).reshape(n, -1)
            buf[:, :a.shape[1]] = a
        rc = lib().no_build_region_fields(C.byref(self._map), _p(reqs), n, _p(seeds), _p(ov), _p(buf),
                                          out_stride)
        if rc:
            raise ValueError("no_build_region_fields: bad request")
        return buf

    def build_los(self, reqs, prev=None):
        """N_LOSFieldCreate for each request; prev: [n,64,64] previous fields or None."""
        reqs = np.ascontiguousarray(reqs, LOS_REQ_DTYPE)
        n = len(reqs)
        out = np.zeros((n, 64, 64), np.uint8)
        p = None if prev is None else np.ascontiguousarray(prev, np.uint8).reshape(n, 64, 64)
        rc = lib().no_build_los(C.byref(self._map), _p(reqs), n, _p(p) if p is not None else None,
                                _p(out), self.w * 128.0, -self.h * 128.0)
        if rc:
            raise ValueError("no_build_los: bad request")
        return out

    def field_bench(self, reqs, reps=1, nthreads=1):
        reqs = np.ascontiguousarray(reqs, FIELD_REQ_DTYPE)
        return lib().no_field_bench(C.byref(self._map), _p(reqs), len(reqs), reps, nthreads)

    # -- agents ---------------------------------------------------------------------------
    def agent_step(self, arrays, hz=20, work=None, nthreads=1, timed=False):
        w, keep = make_world(self.w, self.h, arrays, hz, work)
        n = w.n_ents
        out = {k: np.zeros((n, 2), np.float32) for k in ("vel_xz", "new_pos_xz", "vdes_xz", "vpref_xz")}
        out["status"] = np.zeros(n, np.uint8)
        so = StepOut(*[out[k].ctypes.data for k in ("vel_xz", "new_pos_xz", "vdes_xz", "vpref_xz", "status")])
        if timed:
            out["seconds"] = lib().no_agent_bench(C.byref(self._map), C.byref(w), C.byref(so), nthreads)
        else:
            rc = lib().no_agent_step(C.byref(self._map), C.byref(w), C.byref(so), nthreads)
            if rc:
                raise ValueError("no_agent_step: bad input")
        return out

    def forces(self, arrays, uid, vdes, hz=20):
        w, keep = make_world(self.w, self.h, arrays, hz)
        a, c, s = (np.zeros(2, np.float32) for _ in range(3))
        v = np.ascontiguousarray(vdes, np.float32)
        lib().no_agent_forces(C.byref(self._map), C.byref(w), uid, _p(v), _p(a), _p(c), _p(s))
        return a, c, s


def spatial_query(chunk_w, chunk_h, pos_xz, query_xz, rng, maxout):
    w, keep = make_world(chunk_w, chunk_h, {"pos_xz": np.ascontiguousarray(pos_xz, np.float32)})
    q = np.ascontiguousarray(query_xz, np.float32).reshape(-1, 2)
    counts = np.zeros(len(q), np.int32)
    ids = np.zeros((len(q), maxout), np.uint32)
    rc = lib().no_spatial_query(C.byref(w), _p(q), len(q), rng, maxout, _p(counts), _p(ids))
    if rc:
        raise ValueError("no_spatial_query failed")
    return counts, ids


def clearpath(ent, des_v, dyn, n_dyn, stat, n_stat):
    ent = np.ascontiguousarray(ent, np.float32).reshape(-1, 5)
    nq = len(ent)
    des_v = np.ascontiguousarray(des_v, np.float32).reshape(nq, 2)
    dyn = np.ascontiguousarray(dyn, np.float32).reshape(nq, 32, 5)
    stat = np.ascontiguousarray(stat, np.float32).reshape(nq, 32, 5)
    n_dyn = np.ascontiguousarray(n_dyn, np.int32)
    n_stat = np.ascontiguousarray(n_stat, np.int32)
    out = np.zeros((nq, 2), np.float32)
    rc = lib().no_clearpath(nq, _p(ent), _p(des_v), _p(dyn), _p(n_dyn), _p(stat), _p(n_stat), _p(out))
    if rc:
        raise ValueError("no_clearpath: bad counts")
    return out
