"""oracle/pfref.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding of oracle/_ref/libpfref.so (the reference engine's own nav /
ClearPath / movement code, built by oracle/ref/build_ref.py from the sources
in place under /root/reference).  Only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

TARGET_PORTAL, TARGET_TILE = 0, 1
FACTION_ID_NONE = 0xF
ISLAND_NONE = 0xFFFF
PLANE_COST, PLANE_BLOCKERS, PLANE_ISLANDS, PLANE_LOCAL_ISLANDS, PLANE_FACTIONS = range(5)


class FieldReq(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "layer", "type", "faction_id", "inout", "chunk_r", "chunk_c", "tile_r", "tile_c",
        "port_r0", "port_c0", "port_r1", "port_c1", "next_chunk_r", "next_chunk_c",
        "next_r0", "next_c0", "next_r1", "next_c1", "port_iid", "next_iid")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


FIELD_REQ_DTYPE = np.dtype([(n, np.int32) for n, _ in FieldReq._fields_])


class Portal(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "chunk_r", "chunk_c", "r0", "c0", "r1", "c1", "conn_chunk_r", "conn_chunk_c",
        "conn_r0", "conn_c0", "conn_r1", "conn_c1", "component_id")]


ASYNC_REQ_DTYPE = np.dtype([("kind", np.int32), ("layer", np.int32), ("faction_id", np.int32), ("x", np.float32),
                            ("z", np.float32), ("ent", np.uint32), ("radius", np.int32)])
ASYNC_ENEMY_SEEK, ASYNC_SURROUND, ASYNC_GROUP_ARRIVAL = 0, 1, 2


class MoveWorld(C.Structure):
    _fields_ = [
        ("n", C.c_int), ("pos_xz", C.c_void_p), ("vel_xz", C.c_void_p), ("radius", C.c_void_p),
        ("max_speed", C.c_void_p), ("speed", C.c_void_p), ("flags", C.c_void_p),
        ("state", C.c_void_p), ("flock", C.c_void_p), ("has_dest_los", C.c_void_p),
        ("n_flocks", C.c_int), ("flock_target_xz", C.c_void_p), ("flock_dest_id", C.c_void_p),
        ("hz", C.c_int)]


def available():
    return os.path.exists(os.path.join(_HERE, "_ref", "libpfref.so"))


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_ref", "libpfref.so")
        if not os.path.exists(path):
            from oracle.ref import build_ref
            if build_ref.build(release=True) is None:
                raise RuntimeError("oracle/_ref/libpfref.so missing and /root/reference absent")
        L = C.CDLL(path)
        L.pfref_nav_create.restype = C.c_void_p
        L.pfref_nav_create.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_uint]
        L.pfref_nav_destroy.argtypes = [C.c_void_p]
        L.pfref_nav_set_blockers.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.pfref_nav_blockers_circle.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float,
                                                C.c_int, C.c_uint32, C.c_int]
        L.pfref_nav_flush_dirty.argtypes = [C.c_void_p]
        L.pfref_nav_copy_plane.restype = C.c_size_t
        L.pfref_nav_copy_plane.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.pfref_nav_num_portals.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.pfref_nav_get_portal.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.POINTER(Portal)]
        L.pfref_field_update.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pfref_flow_field_id.restype = C.c_uint64
        L.pfref_flow_field_id.argtypes = [C.c_void_p, C.c_void_p]
        L.pfref_region_field_id.restype = C.c_uint64
        L.pfref_region_field_id.argtypes = [C.c_int] * 4 + [C.c_uint32, C.c_int, C.c_int]
        L.pfref_game_load.argtypes = [C.c_float] * 4 + [C.c_int] + [C.c_void_p] * 4
        L.pfref_async_batch.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.pfref_desired_region_velocities.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pfref_cached_field_by_id.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p]
        L.pfref_hip_async_stats.argtypes = [C.c_void_p]
        L.pfref_hip_los_stats.argtypes = [C.c_void_p]
        L.pfref_hip_blockers_stats.argtypes = [C.c_void_p]
        L.pfref_hip_ctx.restype = C.c_void_p
        L.pfref_nav_dirty_chunks.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.pfref_field_nearest_pathable.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p]
        L.pfref_field_island_to_nearest.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.pfref_cell_arrival_field.argtypes = [C.c_void_p] + [C.c_int] * 7 + [C.c_void_p, C.c_int, C.c_void_p]
        L.pfref_group_arrival_field.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                                C.c_float, C.c_float, C.c_void_p, C.c_int, C.c_void_p]
        L.pfref_zone_field.argtypes = [C.c_void_p] + [C.c_int] * 6 + [C.c_void_p, C.c_void_p, C.c_int,
                                                                      C.c_void_p]
        L.pfref_los_field.argtypes = [C.c_void_p] + [C.c_int] * 10 + [C.c_void_p, C.c_void_p]
        L.pfref_hip_init.argtypes = [C.c_void_p]
        L.pfref_hip_mode.argtypes = [C.c_int, C.c_int]
        L.pfref_hip_sync_layer.argtypes = [C.c_void_p, C.c_int]
        L.pfref_hip_stats.argtypes = [C.c_void_p]
        L.pfref_desired_velocities.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                               C.c_void_p]
        L.pfref_dest_id.restype = C.c_uint32
        L.pfref_dest_id.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
        L.pfref_cache_clear.argtypes = [C.c_void_p]
        L.pfref_move_velocity_hip.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.pfref_move_velocity_hip.restype = C.c_int
        L.pfref_field_update_many.restype = C.c_double
        L.pfref_field_update_many.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.pfref_field_bench.restype = C.c_double
        L.pfref_field_bench.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.pfref_request_path.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float,
                                         C.c_float, C.c_float, C.c_int, C.POINTER(C.c_uint32)]
        L.pfref_trace_get.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pfref_desired_point_seek_velocity.argtypes = [C.c_void_p, C.c_uint32, C.c_float,
                                                        C.c_float, C.c_float, C.c_float,
                                                        C.c_void_p]
        L.pfref_cached_field.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
        L.pfref_move_state_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.pfref_closest_pathable.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p]
        L.pfref_dest_island_tiles.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_int]
        L.pfref_cached_ffid.restype = C.c_uint64
        L.pfref_cached_ffid.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int]
        L.pfref_cached_los.argtypes = [C.c_void_p, C.c_uint32, C.c_int, C.c_int, C.c_void_p]
        L.pfref_has_dest_los.argtypes = [C.c_void_p, C.c_uint32, C.c_float, C.c_float,
                                         C.c_float, C.c_float]
        L.pfref_position_pathable.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float]
        L.pfref_position_blocked.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float]
        L.pfref_map_pos.argtypes = [C.c_void_p, C.c_void_p]
        for name, at, rt in (
            ("pfref_clearpath_new_velocity", [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                              C.c_void_p, C.c_int, C.c_void_p], None),
            ("pfref_spatial_query", [C.c_float] * 4 + [C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                                       C.c_float, C.c_int, C.c_void_p,
                                                       C.c_void_p], None),
            ("pfref_move_load", [C.c_void_p, C.POINTER(MoveWorld)], C.c_int),
            ("pfref_move_velocity", [C.c_void_p, C.c_int, C.c_int, C.c_void_p], None),
            ("pfref_move_unload", [], None),
            ("pfref_move_set_formation", [C.c_void_p] * 5, None),
            ("pfref_move_set_arrival", [C.c_void_p] * 2, None),
            ("pfref_move_set_arrival_zone", [C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int,
                                             C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int], C.c_int),
            ("pfref_move_set_arrival_units", [C.c_void_p] * 7, None),
            ("pfref_move_get_arrival_units", [C.c_void_p] * 4, None),
            ("pfref_move_hip_settle_stats", [C.c_void_p], None),
            ("pfref_move_hip_wait_differ", [], C.c_long),
            ("pfref_move_hip_state_work_seconds", [], C.c_double),
            ("pfref_move_hip_state_times", [C.c_void_p], None),
            ("pfref_move_set_state_aux", [C.c_void_p] * 3, None),
            ("pfref_move_get_wait_ticks", [C.c_void_p], None),
            ("pfref_move_set_turning", [C.c_void_p] * 2, None),
            ("pfref_move_set_range_targets", [C.c_void_p] * 3, None),
            ("pfref_move_set_surround", [C.c_void_p] * 3, None),
            ("pfref_move_hip_surround_differ", [], C.c_long),
            ("pfref_move_hip_resident_state_pass", [C.c_int], None),
            ("pfref_move_hip_resident_passes", [], C.c_long),
            ("pfref_move_get_out", [C.c_void_p] * 2, None),
            ("pfref_move_get_surround", [C.c_void_p] * 3, None),
            ("pfref_move_surround_queries", [C.c_void_p] * 3, None),
            ("pfref_move_set_next_rot", [C.c_void_p], None),
            ("pfref_move_set_interp", [C.c_void_p] * 2, None),
            ("pfref_move_heading_gate", [C.c_void_p] * 3 + [C.c_int] * 2 + [C.c_void_p] * 2, None),
            ("pfref_move_dir_quat", [C.c_void_p, C.c_int, C.c_void_p], None),
            ("pfref_move_settled_count", [C.c_void_p, C.c_int, C.c_void_p], None),
            ("pfref_arrival_should_settle", [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                             C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                             C.c_void_p, C.c_int] + [C.c_void_p] * 12, C.c_int),
            ("pfref_move_vpref", [C.c_int, C.c_void_p, C.c_void_p], None),
            ("pfref_move_forces", [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
             None),
            ("pfref_move_neighbours", [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
             C.c_int),
            ("pfref_move_get_vdes", [C.c_void_p], None),
            ("pfref_move_flock_order", [C.c_int, C.c_void_p], C.c_int),
            ("pfref_move_bench", [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p],
             C.c_double),
        ):
            if hasattr(L, name):
                f = getattr(L, name)
                f.argtypes = at
                f.restype = rt
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class RefNav:
    """The reference's `struct nav_private` built from explicit cost planes."""

    def __init__(self, cost_base, layer_mask=1):
        cost_base = np.ascontiguousarray(cost_base, dtype=np.uint8)
        assert cost_base.ndim == 4 and cost_base.shape[2:] == (64, 64)
        self.h, self.w = cost_base.shape[:2]
        self.layer_mask = layer_mask
        self._h = lib().pfref_nav_create(self.w, self.h, _p(cost_base), layer_mask)
        if not self._h:
            raise RuntimeError("pfref_nav_create failed")
        mp = np.zeros(3, np.float32)
        lib().pfref_map_pos(self._h, _p(mp))
        self.map_pos = mp

    def close(self):
        if self._h:
            lib().pfref_nav_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- planes ---------------------------------------------------------
    def plane(self, which, layer=0):
        dt = np.uint8 if which in (PLANE_COST, PLANE_FACTIONS) else np.uint16
        shape = (self.h, self.w, 64, 64) if which != PLANE_FACTIONS else (self.h, self.w, 15, 64, 64)
        out = np.zeros(shape, dt)
        lib().pfref_nav_copy_plane(self._h, layer, which, _p(out))
        return out

    def set_blockers(self, blockers, layer=0):
        b = np.ascontiguousarray(blockers, dtype=np.uint16)
        assert b.shape == (self.h, self.w, 64, 64)
        lib().pfref_nav_set_blockers(self._h, layer, _p(b))

    def blockers_circle(self, x, z, rng, faction_id=0, flags=0, incref=True):
        lib().pfref_nav_blockers_circle(self._h, x, z, rng, faction_id, flags, int(incref))

    def flush_dirty(self):
        lib().pfref_nav_flush_dirty(self._h)

    def portals(self, chunk_r, chunk_c, layer=0):
        n = lib().pfref_nav_num_portals(self._h, layer, chunk_r, chunk_c)
        out = []
        for i in range(n):
            p = Portal()
            lib().pfref_nav_get_portal(self._h, layer, chunk_r, chunk_c, i, C.byref(p))
            out.append(p)
        return out

    # -- fields ---------------------------------------------------------
    def field_update(self, req, inout=None, want_integ=False):
        """req: FieldReq or a FIELD_REQ_DTYPE record.  Returns (dirs[64,64] u8, integ|None)."""
        r = _to_req(req)
        dirs = np.zeros((64, 64), np.uint8) if inout is None else \
            np.ascontiguousarray(inout, dtype=np.uint8).reshape(64, 64).copy()
        integ = np.zeros((64, 64), np.float32) if want_integ else None
        rc = lib().pfref_field_update(self._h, C.byref(r), _p(dirs),
                                      _p(integ) if want_integ else None)
        if rc != 0:
            raise ValueError("pfref_field_update: bad request")
        return dirs, integ

    def flow_field_id(self, req):
        """N_FlowFieldID (field.c:1952) of a TILE / PORTAL request."""
        r = _to_req(req)
        return int(lib().pfref_flow_field_id(self._h, C.byref(r)))

    @staticmethod
    def region_field_id(kind, layer, chunk_r, chunk_c, a, b=0, c=0):
        """N_FlowFieldID of an ENEMIES (2: a = faction) / ENTITY (4: a = uid) / ZONE (5: a, b = centre in
        absolute tiles, c = radius) target."""
        return int(lib().pfref_region_field_id(kind, layer, chunk_r, chunk_c, a, b, c))

    def field_nearest_pathable(self, chunk_r, chunk_c, start_r, start_c, existing, layer=0,
                               faction_id=FACTION_ID_NONE):
        """N_FlowFieldUpdateToNearestPathable on a copy of `existing` ([64,64] u8)."""
        dirs = np.ascontiguousarray(existing, np.uint8).reshape(64, 64).copy()
        lib().pfref_field_nearest_pathable(self._h, layer, chunk_r, chunk_c, start_r, start_c,
                                           faction_id, _p(dirs))
        return dirs

    def field_island_to_nearest(self, req, local_iid, existing):
        """N_FlowFieldUpdateIslandToNearest(local_iid) on a copy of `existing`, whose target is req."""
        r = _to_req(req)
        dirs = np.ascontiguousarray(existing, np.uint8).reshape(64, 64).copy()
        rc = lib().pfref_field_island_to_nearest(self._h, C.byref(r), int(local_iid), _p(dirs))
        if rc != 0:
            raise ValueError("pfref_field_island_to_nearest: bad request")
        return dirs

    def cell_arrival_field(self, dim, target_abs, center_abs, blocked=None, layer=0, enemies=0):
        out = np.zeros(dim * dim // 2, np.uint8)
        b = np.zeros((0, 2), np.int16) if blocked is None else np.ascontiguousarray(blocked, np.int16)
        lib().pfref_cell_arrival_field(self._h, dim, layer, enemies, target_abs[0], target_abs[1],
                                       center_abs[0], center_abs[1], _p(b) if len(b) else None, len(b),
                                       _p(out))
        return out

    def group_arrival_field(self, dim, targets_xz, center_xz, blocked=None, layer=0, enemies=0):
        out = np.zeros(dim * dim // 2, np.uint8)
        t = np.ascontiguousarray(targets_xz, np.float32).reshape(-1, 2)
        b = np.zeros((0, 2), np.int16) if blocked is None else np.ascontiguousarray(blocked, np.int16)
        lib().pfref_group_arrival_field(self._h, dim, layer, enemies, _p(t), len(t), center_xz[0],
                                        center_xz[1], _p(b) if len(b) else None, len(b), _p(out))
        return out

    def zone_field(self, chunk, center_abs, radius, existing, layer=0):
        """TARGET_ZONE chunk field on a copy of `existing`; returns (dirs, seeds [k,2] i16, geometry
        dict of the padded region field_update_zone integrates over)."""
        dirs = np.ascontiguousarray(existing, np.uint8).reshape(64, 64).copy()
        seeds = np.zeros((128 * 128, 2), np.int16)
        geom = (C.c_int * 6)()
        n = lib().pfref_zone_field(self._h, layer, chunk[0], chunk[1], center_abs[0], center_abs[1],
                                   radius, _p(dirs), _p(seeds), len(seeds), geom)
        g = dict(base_abs_r=geom[0], base_abs_c=geom[1], rdim=geom[2], roff=geom[3], coff=geom[4],
                 cdim=geom[5])
        return dirs, seeds[:n].copy(), g

    def los_field(self, chunk, target, prev=None, prev_d=(0, 0), layer=0, faction_id=FACTION_ID_NONE):
        """N_LOSFieldCreate: chunk=(r,c), target=(chunk_r,chunk_c,tile_r,tile_c); prev = previous
        chunk's field ([64,64] u8) with prev_d = its chunk offset.  Returns [64,64] u8."""
        out = np.zeros((64, 64), np.uint8)
        p = None if prev is None else np.ascontiguousarray(prev, np.uint8)
        lib().pfref_los_field(self._h, layer, faction_id, chunk[0], chunk[1], target[0], target[1],
                              target[2], target[3], prev_d[0], prev_d[1],
                              _p(p) if p is not None else None, _p(out))
        return out

    def field_bench(self, reqs, reps=1, nthreads=1):
        reqs = np.ascontiguousarray(reqs, dtype=FIELD_REQ_DTYPE)
        return lib().pfref_field_bench(self._h, _p(reqs), len(reqs), reps, nthreads)

    def field_update_many(self, reqs, nthreads=8):
        """N_FlowFieldInit + N_FlowFieldUpdate for every request (none in place), threaded; [n,64,64] u8."""
        reqs = np.ascontiguousarray(reqs, dtype=FIELD_REQ_DTYPE)
        out = np.zeros((len(reqs), 64, 64), np.uint8)
        t = lib().pfref_field_update_many(self._h, _p(reqs), len(reqs), nthreads, _p(out))
        assert t >= 0
        return out

    # -- the reference-side binding of libnavhip.so (bindings/permafrost/nav_hip.c) -------------------
    def hip_init(self):
        return bool(lib().pfref_hip_init(self._h))

    @staticmethod
    def hip_shutdown():
        lib().pfref_hip_shutdown()

    @staticmethod
    def hip_mode(use_binding, backend=1):
        lib().pfref_hip_mode(int(use_binding), int(backend))

    def hip_sync_layer(self, layer=0):
        return bool(lib().pfref_hip_sync_layer(self._h, layer))

    @staticmethod
    def hip_stats():
        out = (C.c_long * 3)()
        lib().pfref_hip_stats(out)
        return {"device_builds": out[0], "batches": out[1], "requests": out[2]}

    def dest_id(self, dst, layer=0, faction_id=FACTION_ID_NONE):
        return int(lib().pfref_dest_id(self._h, layer, faction_id, float(dst[0]), float(dst[1])))

    def cache_put_fields(self, reqs, dest_ids, dirs):
        """N_FC_PutFlowField + N_FC_PutDestFFMapping for every (request, dest id, 4096 direction bytes): the state
        n_request_path leaves in the reference's own field cache."""
        reqs = np.ascontiguousarray(reqs, dtype=FIELD_REQ_DTYPE)
        ids = np.ascontiguousarray(dest_ids, np.uint32)
        d = np.ascontiguousarray(dirs, np.uint8).reshape(len(reqs), 4096)
        L = lib()
        L.pfref_cache_put_fields.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        return int(L.pfref_cache_put_fields(self._h, len(reqs), _p(reqs), _p(ids), _p(d)))

    def cache_clear(self):
        lib().pfref_cache_clear(self._h)

    def desired_velocities(self, dest_ids, pos, dst, batched=False):
        """N_DesiredPointSeekVelocity for every agent: serial calls in order, or the binding's
        miss-collecting batched form."""
        ids = np.ascontiguousarray(dest_ids, np.uint32)
        p = np.ascontiguousarray(pos, np.float32).reshape(-1, 2)
        d = np.ascontiguousarray(dst, np.float32).reshape(-1, 2)
        out = np.zeros_like(p)
        lib().pfref_desired_velocities(self._h, len(ids), _p(ids), _p(p), _p(d), int(batched), _p(out))
        return out

    def cache_dump(self, dest_ids):
        """{(dest_id, chunk_r, chunk_c): dirs} of every field the cache maps for these destinations."""
        out = {}
        for did in sorted(set(int(x) for x in dest_ids)):
            for cr in range(self.h):
                for cc in range(self.w):
                    ff = self.cached_field(did, cr, cc)
                    if ff is not None:
                        out[(did, cr, cc)] = ff.copy()
        return out

    # -- the asynchronous field batch / LOS / blockers seams of the binding ---------------
    def game_load(self, pos_xz, radius, faction, flags):
        """Explicit game state for the reference's enemy / entity frontier extraction (uid == index)."""
        p = np.ascontiguousarray(pos_xz, np.float32).reshape(-1, 2)
        r = np.ascontiguousarray(radius, np.float32)
        f = np.ascontiguousarray(faction, np.int32)
        g = np.ascontiguousarray(flags, np.uint32)
        hx, hz = self.w * 128.0, self.h * 128.0
        lib().pfref_game_load(-hx, hx, -hz, hz, len(p), _p(p), _p(r), _p(f), _p(g))

    @staticmethod
    def game_unload():
        lib().pfref_game_unload()

    def async_batch(self, reqs):
        """compute_async_fields of one tick (N_PrepareAsyncWork, the N_RequestAsync*Field calls,
        N_AwaitAsyncFields): returns {ff_id: dirs [64,64]} of the jobs the batch took."""
        reqs = np.ascontiguousarray(reqs, ASYNC_REQ_DTYPE)
        ids = np.zeros(256, np.uint64)
        n = lib().pfref_async_batch(self._h, len(reqs), _p(reqs), _p(ids), len(ids))
        out = {}
        for i in ids[:n]:
            d = np.zeros((64, 64), np.uint8)
            assert lib().pfref_cached_field_by_id(self._h, int(i), _p(d)), hex(int(i))
            out[int(i)] = d
        return out

    def desired_region_velocities(self, reqs):
        """N_DesiredEnemySeekVelocity / N_DesiredSurroundVelocity / N_DesiredGroupArrivalVelocity per agent
        (ASYNC_REQ_DTYPE records; kind 2 carries the zone centre as float bits in ent / faction_id).
        Returns (vel [n,2] f32, flags [n] u8: bit 0 lookup ok, bit 1 at_slot)."""
        reqs = np.ascontiguousarray(reqs, ASYNC_REQ_DTYPE)
        out = np.zeros((len(reqs), 2), np.float32)
        fl = np.zeros(len(reqs), np.uint8)
        lib().pfref_desired_region_velocities(self._h, len(reqs), _p(reqs), _p(out), _p(fl))
        return out, fl

    @staticmethod
    def hip_seam_stats():
        a, l, b = (C.c_long * 3)(), (C.c_long * 2)(), (C.c_long * 2)()
        lib().pfref_hip_async_stats(a)
        lib().pfref_hip_los_stats(l)
        lib().pfref_hip_blockers_stats(b)
        return {"async_device_jobs": a[0], "async_batches": a[1], "async_cpu_jobs": a[2],
                "los_device_fields": l[0], "los_batches": l[1], "blocker_circles": b[0], "blocker_batches": b[1]}

    @staticmethod
    def hip_pool(n_slots=0, n_rows=0):
        """N_HIP_PoolEnable / Disable: the field cache's device image."""
        if n_slots:
            return bool(lib().pfref_hip_pool_enable(int(n_slots), int(n_rows)))
        lib().pfref_hip_pool_disable()
        return True

    @staticmethod
    def hip_pool_stats():
        p, m = (C.c_long * 3)(), (C.c_long * 3)()
        lib().pfref_hip_pool_stats(p)
        lib().pfref_move_hip_stats(m)
        return {"puts": p[0], "maps": p[1], "built_resident": p[2], "device_sampled": m[0], "host_fallbacks": m[1],
                "steps": m[2]}

    @staticmethod
    def hip_device_sampling(on):
        lib().pfref_move_hip_sampling(int(bool(on)))

    @staticmethod
    def hip_blockers_flush():
        return bool(lib().pfref_hip_blockers_flush())

    @staticmethod
    def hip_ctx():
        """The binding's navhip_ctx* (so that a test can read the device planes through libnavhip)."""
        return lib().pfref_hip_ctx()

    def dirty_chunks(self, layer=0):
        out = np.zeros((self.h, self.w), np.uint8)
        lib().pfref_nav_dirty_chunks(self._h, layer, _p(out))
        return out

    def los_dump(self, dest_ids):
        out = {}
        for did in sorted(set(int(x) for x in dest_ids)):
            for cr in range(self.h):
                for cc in range(self.w):
                    lf = self.cached_los(did, cr, cc)
                    if lf is not None:
                        out[(did, cr, cc)] = lf.copy()
        return out

    # -- planner ----------------------------------------------------------
    def request_path(self, src, dst, layer=0, faction_id=FACTION_ID_NONE, clear_cache=False):
        did = C.c_uint32(0)
        ok = lib().pfref_request_path(self._h, layer, faction_id, src[0], src[1], dst[0], dst[1],
                                      int(clear_cache), C.byref(did))
        return bool(ok), did.value

    @staticmethod
    def trace(clear=True):
        L = lib()
        n = L.pfref_trace_count()
        reqs = np.zeros(n, FIELD_REQ_DTYPE)
        before = np.zeros((n, 64, 64), np.uint8)
        after = np.zeros((n, 64, 64), np.uint8)
        for i in range(n):
            L.pfref_trace_get(i, _p(reqs[i:i + 1]), _p(before[i]), _p(after[i]))
        if clear:
            L.pfref_trace_clear()
        return reqs, before, after

    @staticmethod
    def los_trace(clear=True):
        """The planner's N_LOSFieldCreate calls since the last clear: [n][6] = dest id, chunk r, c, has_prev,
        prev chunk r, c (the chain of nav.c:1843 / :2026-2039)."""
        L = lib()
        L.pfref_los_trace_get.argtypes = [C.c_int, C.c_void_p]
        n = L.pfref_los_trace_count()
        out = np.zeros((n, 6), np.int32)
        for i in range(n):
            L.pfref_los_trace_get(i, _p(out[i]))
        if clear:
            L.pfref_los_trace_clear()
        return out

    def desired_velocity(self, dest_id, pos, dst):
        out = np.zeros(2, np.float32)
        lib().pfref_desired_point_seek_velocity(self._h, dest_id, pos[0], pos[1], dst[0], dst[1],
                                                _p(out))
        return out

    def cached_field(self, dest_id, chunk_r, chunk_c):
        out = np.zeros((64, 64), np.uint8)
        ok = lib().pfref_cached_field(self._h, dest_id, chunk_r, chunk_c, _p(out))
        return out if ok else None

    def closest_pathable(self, xz, layer=0):
        """N_ClosestPathable (nav.c:4126): (x, z) or None."""
        out = np.zeros(2, np.float32)
        ok = lib().pfref_closest_pathable(self._h, layer, float(xz[0]), float(xz[1]), _p(out))
        return out if ok else None

    def dest_island_tiles(self, xz, layer=0):
        """The tiles N_IsMaximallyClose (nav.c:4707) tests for this destination: [k, 2] int16 absolute (row, col)."""
        out = np.zeros((256, 2), np.int16)
        n = lib().pfref_dest_island_tiles(self._h, layer, float(xz[0]), float(xz[1]), _p(out), len(out))
        return out[:n].copy()

    def cached_ffid(self, dest_id, chunk_r, chunk_c):
        """N_FC_GetDestFFMapping: id of the flow field mapped for (dest, chunk), 0 = none."""
        return int(lib().pfref_cached_ffid(self._h, dest_id, chunk_r, chunk_c))

    def cached_los(self, dest_id, chunk_r, chunk_c):
        """The LOS field the field cache holds for (dest, chunk), [64,64] u8 (bit 0 visible), or None."""
        out = np.zeros((64, 64), np.uint8)
        ok = lib().pfref_cached_los(self._h, dest_id, chunk_r, chunk_c, _p(out))
        return out if ok else None

    def has_dest_los(self, dest_id, pos, dst):
        return bool(lib().pfref_has_dest_los(self._h, dest_id, pos[0], pos[1], dst[0], dst[1]))

    def position_pathable(self, pos, layer=0):
        return bool(lib().pfref_position_pathable(self._h, layer, pos[0], pos[1]))

    def position_blocked(self, pos, layer=0):
        return bool(lib().pfref_position_blocked(self._h, layer, pos[0], pos[1]))


def _to_req(req):
    if isinstance(req, FieldReq):
        return req
    r = FieldReq()
    if isinstance(req, dict):
        for k, v in req.items():
            setattr(r, k, int(v))
    else:
        for n in FIELD_REQ_DTYPE.names:
            setattr(r, n, int(req[n]))
    return r


def set_enemy_factions(faction_id, mask):
    """What G_GetEnemyFactions(faction_id) returns inside the reference (game.c:2744)."""
    L = lib()
    L.pfref_set_enemy_factions.argtypes = [C.c_int, C.c_uint]
    L.pfref_set_enemy_factions(int(faction_id), int(mask))


def clearpath_new_velocity(ent, des_v, dyn, stat):
    ent = np.ascontiguousarray(ent, np.float32)
    des_v = np.ascontiguousarray(des_v, np.float32)
    dyn = np.ascontiguousarray(dyn, np.float32).reshape(-1, 5)
    stat = np.ascontiguousarray(stat, np.float32).reshape(-1, 5)
    out = np.zeros(2, np.float32)
    lib().pfref_clearpath_new_velocity(_p(ent), _p(des_v), _p(dyn), len(dyn), _p(stat), len(stat),
                                       _p(out))
    return out


def spatial_query(bounds, pos_xz, query_xz, rng, maxout):
    pos_xz = np.ascontiguousarray(pos_xz, np.float32).reshape(-1, 2)
    query_xz = np.ascontiguousarray(query_xz, np.float32).reshape(-1, 2)
    counts = np.zeros(len(query_xz), np.int32)
    ids = np.zeros((len(query_xz), maxout), np.uint32)
    lib().pfref_spatial_query(bounds[0], bounds[1], bounds[2], bounds[3], _p(pos_xz), len(pos_xz),
                              _p(query_xz), len(query_xz), rng, maxout, _p(counts), _p(ids))
    return counts, ids


STATE_MOVING, STATE_MOVING_IN_FORMATION, STATE_ARRIVED, STATE_SEEK_ENEMIES, STATE_WAITING, \
    STATE_SURROUND_ENTITY, STATE_ENTER_ENTITY_RANGE, STATE_TURNING, STATE_ARRIVING_TO_CELL = range(9)
ENTITY_FLAG_MOVABLE = 1 << 3
ENTITY_FLAG_AIR = 1 << 15
ENTITY_FLAG_GARRISONED = 1 << 18
ENTITY_FLAG_COMBAT_HELD = 1 << 21


class RefMove:
    """movement.c's static velocity pipeline loaded with an explicit world snapshot."""

    def __init__(self, nav, pos, vel, radius, max_speed, speed, flags, state, flock,
                 has_dest_los, flock_target_xz, flock_dest_id, hz=20):
        n = len(pos)
        self.n = n
        self._keep = dict(
            pos=np.ascontiguousarray(pos, np.float32).reshape(n, 2),
            vel=np.ascontiguousarray(vel, np.float32).reshape(n, 2),
            radius=np.ascontiguousarray(radius, np.float32),
            max_speed=np.ascontiguousarray(max_speed, np.float32),
            speed=np.ascontiguousarray(speed, np.float32),
            flags=np.ascontiguousarray(flags, np.uint32),
            state=np.ascontiguousarray(state, np.int32),
            flock=np.ascontiguousarray(flock, np.int32),
            los=np.ascontiguousarray(has_dest_los, np.uint8),
            ftgt=np.ascontiguousarray(flock_target_xz, np.float32).reshape(-1, 2),
            fdest=np.ascontiguousarray(flock_dest_id, np.uint32))
        k = self._keep
        w = MoveWorld(n, _p(k["pos"]).value, _p(k["vel"]).value, _p(k["radius"]).value,
                      _p(k["max_speed"]).value, _p(k["speed"]).value, _p(k["flags"]).value,
                      _p(k["state"]).value, _p(k["flock"]).value, _p(k["los"]).value,
                      len(k["fdest"]), _p(k["ftgt"]).value, _p(k["fdest"]).value, hz)
        self.nav = nav
        lib().pfref_move_load(nav._h, C.byref(w))
        if hasattr(lib(), "pfref_move_set_next_rot"):
            lib().pfref_move_set_next_rot(None)          # (an input of the previous world must not leak into this one)

    def velocity(self, vdes=None, begin=0, end=None):
        end = self.n if end is None else end
        out = np.zeros((self.n, 2), np.float32)
        v = None if vdes is None else np.ascontiguousarray(vdes, np.float32).reshape(self.n, 2)
        lib().pfref_move_velocity(_p(v) if v is not None else None, begin, end, _p(out))
        return out

    def set_formation(self, ready, cell_pos, cohesion, align, drag):
        k = self._keep
        k["f_ready"] = np.ascontiguousarray(ready, np.uint8)
        for name, a in (("f_cell", cell_pos), ("f_coh", cohesion), ("f_align", align), ("f_drag", drag)):
            k[name] = np.ascontiguousarray(a, np.float32).reshape(self.n, 2)
        lib().pfref_move_set_formation(_p(k["f_ready"]), _p(k["f_cell"]), _p(k["f_coh"]),
                                       _p(k["f_align"]), _p(k["f_drag"]))

    def velocity_hip(self, vdes=None, begin=0, end=None):
        """move_velocity_work through the WORK_TYPE_HIP arm of the binding (bindings/permafrost/move_hip.c)."""
        end = self.n if end is None else end
        out = np.zeros((self.n, 2), np.float32)
        v = None if vdes is None else np.ascontiguousarray(vdes, np.float32)
        ok = lib().pfref_move_velocity_hip(_p(v) if v is not None else None, begin, end, _p(out))
        return out if ok else None

    def state_update(self, new_vel, vdes, begin=0, end=None):
        """entity_compute_update (movement.c:2303) per unit: (next_state [n] u8, flags [n] u8: bit 0 state
        set, bit 1 next_block, bit 2 UPDATE_SET_MOVING -- next_state = wait_prev --, bit 3 UPDATE_SET_TARGET_DIR, bit 4 UPDATE_SET_DEST)."""
        end = self.n if end is None else end
        v = np.ascontiguousarray(new_vel, np.float32).reshape(self.n, 2)
        d = np.ascontiguousarray(vdes, np.float32).reshape(self.n, 2)
        st, fl = np.zeros(self.n, np.uint8), np.zeros(self.n, np.uint8)
        lib().pfref_move_state_update(_p(v), _p(d), begin, end, _p(st), _p(fl))
        return st, fl

    def state_update_hip(self, new_vel, vdes, begin=0, end=None):
        """The same through bindings/permafrost/move_hip.c (move_hip_state_work + move_hip_update_work):
        (next_state, flags, device flags NAVHIP_SU_*), or None when the device arm declined."""
        end = self.n if end is None else end
        v = np.ascontiguousarray(new_vel, np.float32).reshape(self.n, 2)
        d = np.ascontiguousarray(vdes, np.float32).reshape(self.n, 2)
        st, fl, dv = np.zeros(self.n, np.uint8), np.zeros(self.n, np.uint8), np.zeros(self.n, np.uint8)
        if not lib().pfref_move_state_update_hip(_p(v), _p(d), begin, end, _p(st), _p(fl), _p(dv)):
            return None
        return st, fl, dv

    def hip_state_stats(self):
        """(units decided on the device, units left to the host, passes) of the state binding."""
        out = (C.c_long * 3)()
        lib().pfref_move_hip_state_stats(out)
        return tuple(out)

    def set_arrival_zone(self, flock, zone):
        """A real arrival zone (struct arrival_state) for (flock, zone["layer"]); zone as in arrival_should_settle."""
        f32 = lambda a: np.ascontiguousarray(a, np.float32)
        slots, ring, reg = f32(zone["slots_xz"]).reshape(-1, 2), np.ascontiguousarray(zone["slot_ring"], np.int32), \
            f32(zone["region_xz"]).reshape(-1, 2)
        cen = f32(zone["centre_xz"])
        nk = lib().pfref_move_set_arrival_zone(int(flock), int(zone["layer"]), _p(cen), int(zone["radius"]),
                                               float(zone["unit_radius"]), float(zone["fill_frac"]), int(zone["active_row"]),
                                               int(zone["num_rows"]), _p(slots), _p(ring), len(ring), _p(reg), len(reg))
        assert nk >= 0
        return nk

    def set_arrival_units(self, u):
        """Every unit's struct arrival_unit_state from a dict of n-row arrays (substate, sink_valid, sink_xz,
        order_pos_xz, progress_anchor_xz, progress_anchored, stuck)."""
        k = [np.ascontiguousarray(u["substate"], np.uint8), np.ascontiguousarray(u["sink_valid"], np.uint8),
             np.ascontiguousarray(u["sink_xz"], np.float32), np.ascontiguousarray(u["order_pos_xz"], np.float32),
             np.ascontiguousarray(u["progress_anchor_xz"], np.float32), np.ascontiguousarray(u["progress_anchored"], np.uint8),
             np.ascontiguousarray(u["stuck"], np.int32)]
        assert all(len(a) == self.n for a in k)
        lib().pfref_move_set_arrival_units(*[_p(a) for a in k])

    def get_arrival_units(self):
        out = {"substate": np.zeros(self.n, np.uint8), "progress_anchor_xz": np.zeros((self.n, 2), np.float32),
               "progress_anchored": np.zeros(self.n, np.uint8), "stuck": np.zeros(self.n, np.int32)}
        lib().pfref_move_get_arrival_units(_p(out["substate"]), _p(out["progress_anchor_xz"]), _p(out["progress_anchored"]),
                                           _p(out["stuck"]))
        return out

    def hip_state_work_seconds(self):
        """Wall time of the last move_hip_state_work (snapshot fill + the device calls of the state pass)."""
        return float(lib().pfref_move_hip_state_work_seconds())

    def hip_state_times(self):
        """Where the last move_hip_state_work spent its time, milliseconds."""
        out = (C.c_double * 6)()
        lib().pfref_move_hip_state_times(out)
        return dict(zip(("snapshot", "unit_inputs", "flock_queries", "state_pass_call", "settle_pass", "scatter"),
                        [x * 1e3 for x in out]))

    def hip_wait_differ(self):
        """Wait counters the device's pass left different from the reference's (the state binding's check)."""
        return int(lib().pfref_move_hip_wait_differ())

    def hip_settle_stats(self):
        """(units the device's settle rule decided, of those settled, unit states that differ from the reference's
        afterwards, units whose heading gate the device left to the host) of the state binding."""
        out = (C.c_long * 4)()
        lib().pfref_move_hip_settle_stats(out)
        return tuple(out)

    def set_state_aux(self, fstate, wait_ticks_left, wait_prev):
        """move_work_in.fstate as bits (1 member, 2 ready, 4 assigned, 8 in range, 16 arrived at cell),
        movestate.wait_ticks_left, .wait_prev -- the inputs of the flag / counter arms of the state switch."""
        k = [np.ascontiguousarray(fstate, np.uint8), np.ascontiguousarray(wait_ticks_left, np.int32),
             np.ascontiguousarray(wait_prev, np.uint8)]
        assert all(len(a) == self.n for a in k)
        lib().pfref_move_set_state_aux(*[_p(a) for a in k])

    def set_turning(self, ent_rot, target_dir):
        """STATE_TURNING inputs: the rotation Entity_GetRot answers and movestate.target_dir, [n][4] each; TURNING
        units are driven by state_update from then on."""
        a, b = np.ascontiguousarray(ent_rot, np.float32), np.ascontiguousarray(target_dir, np.float32)
        assert a.shape == b.shape == (self.n, 4)
        lib().pfref_move_set_turning(_p(a), _p(b))

    def set_surround(self, target_uid, target_prev_xz, nearest_prev_xz):
        """STATE_SURROUND_ENTITY inputs (movement.c:2509-2567): movestate.surround_target_uid (-1 = NULL_UID),
        .surround_target_prev, .surround_nearest_prev per unit."""
        k = [np.ascontiguousarray(target_uid, np.int32), np.ascontiguousarray(target_prev_xz, np.float32).reshape(self.n, 2),
             np.ascontiguousarray(nearest_prev_xz, np.float32).reshape(self.n, 2)]
        lib().pfref_move_set_surround(*[_p(a) for a in k])

    def get_surround(self):
        """(surround_target_prev [n][2], surround_nearest_prev [n][2], patch.next_dest [n][2]) after a state update."""
        out = [np.zeros((self.n, 2), np.float32) for _ in range(3)]
        lib().pfref_move_get_surround(*[_p(a) for a in out])
        return out

    def surround_queries(self, new_vel):
        """The two unit-query answers per surround unit as the binding hands them to the device: (query bits [n]:
        1 adjacent-or-gone, 2 / 4 a reachable position exists from pos + new_vel / from pos; dest_xz [n][2][2])."""
        v = np.ascontiguousarray(new_vel, np.float32).reshape(self.n, 2)
        q, d = np.zeros(self.n, np.uint8), np.zeros((self.n, 2, 2), np.float32)
        lib().pfref_move_surround_queries(_p(v), _p(q), _p(d))
        return q, d

    def hip_resident_state_pass(self, on):
        """The binding's state pass on the device-resident snapshot of the velocity pass that precedes it in the tick
        (navhip_state_pass_resident); off: every state pass uploads its own snapshot."""
        lib().pfref_move_hip_resident_state_pass(int(bool(on)))

    def hip_resident_passes(self):
        return int(lib().pfref_move_hip_resident_passes())

    def get_out(self):
        """(ent_vel [n][2], ent_des_v [n][2]) the last velocity pass left in the work items."""
        v, d = np.zeros((self.n, 2), np.float32), np.zeros((self.n, 2), np.float32)
        lib().pfref_move_get_out(_p(v), _p(d))
        return v, d

    def hip_surround_differ(self):
        """Surround positions the device's pass returned that differ from what the reference's switch stored."""
        return int(lib().pfref_move_hip_surround_differ())

    def set_next_rot(self, next_rot):
        """movestate.next_rot [n][4] as an input of state_update / state_update_hip (None: facing on the heading)."""
        if next_rot is None:
            lib().pfref_move_set_next_rot(None)
        else:
            r = np.ascontiguousarray(next_rot, np.float32).reshape(self.n, 4)
            lib().pfref_move_set_next_rot(_p(r))

    def set_interp(self, next_pos_xz, step):
        """movestate.next_pos (x, z) and .step: what a rate below 20 Hz interpolates from (movement.c:2372)."""
        k = [np.ascontiguousarray(next_pos_xz, np.float32).reshape(self.n, 2), np.ascontiguousarray(step, np.float32)]
        lib().pfref_move_set_interp(*[_p(a) for a in k])

    def set_range_targets(self, target_uid, target_range, target_prev_xz):
        """STATE_ENTER_ENTITY_RANGE inputs: movestate.surround_target_uid (-1 = NULL_UID), .target_range, .target_prev_pos."""
        k = [np.ascontiguousarray(target_uid, np.int32), np.ascontiguousarray(target_range, np.float32),
             np.ascontiguousarray(target_prev_xz, np.float32)]
        assert len(k[0]) == len(k[1]) == self.n and k[2].shape == (self.n, 2)
        lib().pfref_move_set_range_targets(*[_p(a) for a in k])

    def get_wait_ticks(self):
        out = np.zeros(self.n, np.int32)
        lib().pfref_move_get_wait_ticks(_p(out))
        return out

    def heading_gate(self, new_vel, vdes, next_rot, begin=0, end=None):
        """entity_compute_update with movestate.next_rot given (the heading gate, movement.c:2319-2336):
        (turn_to_move [n] u8 = UPDATE_TURNING_IN_PLACE of the patch, next_velocity [n][2])."""
        end = self.n if end is None else end
        v = np.ascontiguousarray(new_vel, np.float32).reshape(self.n, 2)
        d = np.ascontiguousarray(vdes, np.float32).reshape(self.n, 2)
        r = np.ascontiguousarray(next_rot, np.float32).reshape(self.n, 4)
        turn, vel = np.zeros(self.n, np.uint8), np.zeros((self.n, 2), np.float32)
        lib().pfref_move_heading_gate(_p(v), _p(d), _p(r), begin, end, _p(turn), _p(vel))
        return turn, vel

    @staticmethod
    def dir_quat(heading_xz):
        """dir_quat_from_velocity (movement.c:1411) per row."""
        h = np.ascontiguousarray(heading_xz, np.float32).reshape(-1, 2)
        out = np.zeros((len(h), 4), np.float32)
        lib().pfref_move_dir_quat(_p(h), len(h), _p(out))
        return out

    def settled_count(self, uids):
        """adjacent_settled_count (movement.c:982) of the loaded snapshot."""
        u = np.ascontiguousarray(uids, np.int32)
        out = np.zeros(len(u), np.int32)
        lib().pfref_move_settled_count(_p(u), len(u), _p(out))
        return out

    def set_arrival(self, sink_xz, flags):
        """Fine-arrival inputs: per-unit slot + flags (bit 0 committed to a valid slot, bit 1 the
        flock's arrival region for the unit's layer is filling)."""
        s = np.ascontiguousarray(sink_xz, np.float32).reshape(self.n, 2)
        f = np.ascontiguousarray(flags, np.uint8)
        lib().pfref_move_set_arrival(_p(s), _p(f))

    def bench(self, vdes, reps=1, nthreads=1, begin=0, end=None):
        end = self.n if end is None else end
        v = np.ascontiguousarray(vdes, np.float32).reshape(self.n, 2)
        out = np.zeros((self.n, 2), np.float32)
        return lib().pfref_move_bench(_p(v), begin, end, reps, nthreads, _p(out)), out

    def hip_dry_run(self, on):
        """Host side of the WORK_TYPE_HIP arm only (no device; the step's outputs read as zero): for timing the fill."""
        lib().pfref_move_hip_dry_run(1 if on else 0)

    def hip_snapshot(self):
        """The snapshot tables bindings/permafrost/move_hip.c builds for the library (dense = ascending uid order), without
        a device: dict of pos, vel, radius, max_speed, flags, state, flock, flock_offsets, flock_members."""
        n = self.n
        o = {"pos": np.zeros((n, 2), np.float32), "vel": np.zeros((n, 2), np.float32), "radius": np.zeros(n, np.float32),
             "max_speed": np.zeros(n, np.float32), "flags": np.zeros(n, np.uint32), "state": np.zeros(n, np.uint8),
             "flock": np.zeros(n, np.int32), "flock_offsets": np.zeros(4096, np.int32), "flock_members": np.zeros(n, np.int32)}
        nf = C.c_int32(0)
        got = lib().pfref_move_hip_snapshot(n, _p(o["pos"]), _p(o["vel"]), _p(o["radius"]), _p(o["max_speed"]), _p(o["flags"]),
                                            _p(o["state"]), _p(o["flock"]), _p(o["flock_offsets"]), _p(o["flock_members"]),
                                            C.byref(nf))
        assert got == n, got
        o["flock_offsets"] = o["flock_offsets"][:nf.value + 1].copy()
        o["flock_members"] = o["flock_members"][:o["flock_offsets"][-1]].copy()
        return o

    def hip_threads(self, nthreads, min_items=0):
        """Fork-join width of the binding's host-side loops (the engine's worker tasks; 1 = the calling task only);
        loops shorter than min_items (0: the binding's default, 8 192) stay on the calling task."""
        lib().pfref_move_hip_threads(int(nthreads), int(min_items))

    def bench_hip(self, vdes, reps=1, begin=0, end=None):
        """The movement tick's velocity half through the binding's WORK_TYPE_HIP arm (bindings/permafrost/move_hip.c),
        `reps` times: (wall seconds, {fill, device, scatter} seconds) or None when the arm declined."""
        end = self.n if end is None else end
        v = np.ascontiguousarray(vdes, np.float32)
        t = (C.c_double * 4)()
        L = lib()
        L.pfref_move_bench_hip.restype = C.c_double
        L.pfref_move_bench_hip.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        dt = L.pfref_move_bench_hip(_p(v), begin, end, reps, t)
        if dt < 0:
            return None
        return dt, {"fill": t[0], "device": t[1], "scatter": t[2]}

    def vpref(self, uid, vdes):
        out = np.zeros(2, np.float32)
        v = np.ascontiguousarray(vdes, np.float32)
        lib().pfref_move_vpref(uid, _p(v), _p(out))
        return out

    def forces(self, uid, vdes):
        a, c, s = (np.zeros(2, np.float32) for _ in range(3))
        v = np.ascontiguousarray(vdes, np.float32)
        lib().pfref_move_forces(uid, _p(v), _p(a), _p(c), _p(s))
        return a, c, s

    def neighbours(self, uid):
        dyn = np.zeros((32, 5), np.float32)
        stat = np.zeros((32, 5), np.float32)
        nd, ns = C.c_int(0), C.c_int(0)
        lib().pfref_move_neighbours(uid, _p(dyn), C.byref(nd), _p(stat), C.byref(ns))
        return dyn[:nd.value].copy(), stat[:ns.value].copy()

    def vdes(self):
        out = np.zeros((self.n, 2), np.float32)
        lib().pfref_move_get_vdes(_p(out))
        return out

    def flock_order(self, f):
        buf = np.zeros(self.n, np.uint32)
        k = lib().pfref_move_flock_order(f, _p(buf))
        return buf[:k].astype(np.int32)

    @staticmethod
    def unload():
        lib().pfref_move_unload()


def arrival_should_settle(nav, zone, units):
    """G_Arrival_ShouldSettle (arrival.c:946) for the units of one zone.  zone: dict(layer, centre_xz, radius,
    unit_radius, fill_frac, active_row, num_rows, slots_xz [S][2], slot_ring [S], region_xz [R][2] positions whose
    tiles are the footprint); units: dict(new_pos_xz, vel_xz, radius, nsettled, substate, sink_valid, sink_xz,
    order_pos_xz, progress_anchor_xz, progress_anchored, stuck), arrays of nq rows.  Returns (settle [nq] u8,
    region_keys u64 sorted, dict of the unit state after the call)."""
    f32 = lambda a, w: np.ascontiguousarray(a, np.float32).reshape(-1, w) if w > 1 else np.ascontiguousarray(a, np.float32)
    slots, ring = f32(zone["slots_xz"], 2), np.ascontiguousarray(zone["slot_ring"], np.int32)
    reg = f32(zone["region_xz"], 2)
    keys = np.zeros(max(1, len(reg)), np.uint64)
    cen = f32(zone["centre_xz"], 1)
    nq = len(units["nsettled"])
    sub = np.array(units["substate"], np.uint8)
    anc = np.array(f32(units["progress_anchor_xz"], 2))
    anced = np.array(units["progress_anchored"], np.uint8)
    stuck = np.array(units["stuck"], np.int32)
    keep = [f32(units["new_pos_xz"], 2), f32(units["vel_xz"], 2), f32(units["radius"], 1),
            np.ascontiguousarray(units["nsettled"], np.int32), np.ascontiguousarray(units["sink_valid"], np.uint8),
            f32(units["sink_xz"], 2), f32(units["order_pos_xz"], 2)]
    out = np.zeros(nq, np.uint8)
    nk = lib().pfref_arrival_should_settle(
        nav._h, int(zone["layer"]), _p(cen), int(zone["radius"]), float(zone["unit_radius"]), float(zone["fill_frac"]),
        int(zone["active_row"]), int(zone["num_rows"]), _p(slots), _p(ring), len(ring), _p(reg), len(reg), _p(keys),
        nq, _p(keep[0]), _p(keep[1]), _p(keep[2]), _p(keep[3]), _p(sub), _p(keep[4]), _p(keep[5]), _p(keep[6]),
        _p(anc), _p(anced), _p(stuck), _p(out))
    assert nk >= 0
    return out, keys[:nk].copy(), {"substate": sub, "progress_anchor_xz": anc, "progress_anchored": anced, "stuck": stuck}
