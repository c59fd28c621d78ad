"""Host-side mirror of the reference's navigation interface over the C ABI of libnavhip.so.

The product boundary is the C ABI (include/navhip.h); this module only loads it with ctypes
and gives the entry points the names / argument meaning of the reference functions they stand
in for (N_FlowFieldInit / N_FlowFieldUpdate / N_FlowFieldID, field.c:2020,2030,1952; the
packed plane uploads of nav.c:2408-2490), so parity tests read like calls into the reference.
PyTorch is used for device buffers / streams only (see `dev_ptr`).

There is NO CPU fallback: if libnavhip.so is missing or no GPU is visible, calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# NAVHIP_LIB: load another build of the same library (A/B runs of kernel variants, scripts/ab_lib.py)
LIB_PATH = os.environ.get("NAVHIP_LIB") or os.path.join(_HERE, "libnavhip.so")

OK = 0
ERR_INVALID, ERR_DEVICE, ERR_NOMEM, ERR_NOT_UPLOADED = -1, -2, -3, -4        # NAVHIP_ERR_*
FIELD_RES = 64
FIELD_CELLS = 4096
COST_IMPASSABLE = 0xFF
ISLAND_NONE = 0xFFFF
FACTION_ID_NONE = 0xF
TARGET_PORTAL, TARGET_TILE, TARGET_NEAREST_PATHABLE = 0, 1, 2
PLANE_COST_BASE, PLANE_BLOCKERS, PLANE_LOCAL_ISLANDS, PLANE_FACTIONS, PLANE_ISLANDS = 0, 1, 2, 3, 4
REQ_INOUT, REQ_IF_CHANGED, REQ_LIVE_IIDS, REQ_ISLAND_NEAREST = 0x1, 0x2, 0x4, 0x8
FD_NONE, FD_NW, FD_N, FD_NE, FD_W, FD_E, FD_SW, FD_S, FD_SE = range(9)

# navhip_field_req, include/navhip.h (32 bytes)
FIELD_REQ_DTYPE = np.dtype([
    ("layer", np.uint8), ("type", np.uint8), ("faction_id", np.uint8), ("flags", np.uint8),
    ("enemies", np.uint16), ("chunk_r", np.uint16), ("chunk_c", np.uint16),
    ("tile_r", np.uint8), ("tile_c", np.uint8),
    ("port_r0", np.uint8), ("port_c0", np.uint8), ("port_r1", np.uint8), ("port_c1", np.uint8),
    ("next_r0", np.uint8), ("next_c0", np.uint8), ("next_r1", np.uint8), ("next_c1", np.uint8),
    ("next_chunk_r", np.uint16), ("next_chunk_c", np.uint16),
    ("port_iid", np.uint16), ("next_iid", np.uint16), ("aux_iid", np.uint16), ("_pad", np.uint16),
], align=False)
assert FIELD_REQ_DTYPE.itemsize == 32

# navhip_circle, include/navhip.h (24 bytes)
CIRCLE_DTYPE = np.dtype([("x", np.float32), ("z", np.float32), ("radius", np.float32),
                         ("faction_id", np.int32), ("flags", np.uint32), ("delta", np.int32)])
assert CIRCLE_DTYPE.itemsize == 24

# navhip_los_req, include/navhip.h (16 bytes)
LOS_REQ_DTYPE = np.dtype([("layer", np.uint8), ("faction_id", np.uint8), ("enemies", np.uint16),
                          ("chunk_r", np.uint16), ("chunk_c", np.uint16),
                          ("target_chunk_r", np.uint16), ("target_chunk_c", np.uint16),
                          ("target_tile_r", np.uint8), ("target_tile_c", np.uint8),
                          ("prev_dr", np.int8), ("prev_dc", np.int8)])
assert LOS_REQ_DTYPE.itemsize == 16

# navhip_region_req, include/navhip.h (32 bytes)
REGION_REQ_DTYPE = np.dtype([("layer", np.uint8), ("out_mode", np.uint8), ("enemies", np.uint16),
                             ("base_abs_r", np.int16), ("base_abs_c", np.int16),
                             ("rdim", np.uint16), ("cdim", np.uint16), ("roff", np.uint16),
                             ("coff", np.uint16), ("seed_begin", np.uint32), ("seed_count", np.uint32),
                             ("overlay_begin", np.uint32), ("overlay_count", np.uint32)])
assert REGION_REQ_DTYPE.itemsize == 32

# exported symbols, checked by the CPU test-suite against include/navhip.h
_SIGS = {
    "navhip_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int]),
    "navhip_ctx_destroy": (None, [C.c_void_p]),
    "navhip_last_error": (C.c_char_p, [C.c_void_p]),
    "navhip_device": (C.c_int, [C.c_void_p]),
    "navhip_stream": (C.c_void_p, [C.c_void_p]),
    "navhip_sync": (C.c_int, [C.c_void_p]),
    "navhip_upload_plane": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "navhip_upload_chunk": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_size_t]),
    "navhip_plane_dev": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int]),
    "navhip_download_plane": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "navhip_blockers_circles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float]),
    "navhip_blockers_circles_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                              C.c_void_p]),
    "navhip_relabel_local_islands": (C.c_int, [C.c_void_p, C.c_int]),
    "navhip_changed_chunks": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int]),
    "navhip_clear_changed": (C.c_int, [C.c_void_p, C.c_void_p]),
    "navhip_build_region_fields": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t,
                                             C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "navhip_build_region_fields_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p,
                                                 C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "navhip_build_los": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_float, C.c_float]),
    "navhip_build_los_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                       C.c_float, C.c_float, C.c_void_p]),
    "navhip_build_fields": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "navhip_build_fields_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "navhip_flow_field_id": (C.c_uint64, [C.c_void_p]),
    "navhip_region_field_id": (C.c_uint64, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_int]),
    "navhip_set_field_kernel": (C.c_int, [C.c_void_p, C.c_int]),
    "navhip_debug_cp_attempts": (C.c_int, [C.c_void_p, C.c_int]),
    "navhip_comm_unique_id": (C.c_int, [C.c_void_p]),
    "navhip_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "navhip_comm_init_mailbox": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "navhip_comm_destroy": (None, [C.c_void_p]),
    "navhip_comm_rank": (C.c_int, [C.c_void_p]),
    "navhip_comm_world": (C.c_int, [C.c_void_p]),
    "navhip_comm_allgather_step_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "navhip_comm_allgather_rows_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
}

COMM_ID_BYTES = 128


def comm_unique_id():
    """ncclGetUniqueId through the library (rank 0); 128 bytes to hand to the other ranks."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    rc = lib().navhip_comm_unique_id(buf)
    if rc != 0:
        raise NavHipError("navhip_comm_unique_id failed (%d): is librccl present?" % rc)
    return bytes(buf)

_lib = None


class NavHipError(RuntimeError):
    pass


def lib():
    """Load libnavhip.so (built in-tree by build.py).  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NavHipError("libnavhip.so not built (run __graft_entry__.build()); "
                              "there is no CPU fallback")
        if os.environ.get("NAVHIP_LIB"):
            # never silent: a development / test build (an A/B variant, the host emulator of tests/hostsim) stands
            # in for the in-tree library -- bench.py names it in config.library
            import sys
            sys.stderr.write("navhip: NAVHIP_LIB=%s replaces the in-tree libnavhip.so\n" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, (rt, at) in _SIGS.items():
            if os.environ.get("NAVHIP_LIB") and not hasattr(L, name):
                continue                 # (an A/B build of an older revision lacks the newer entry points)
            f = getattr(L, name)
            f.restype = rt
            f.argtypes = at
        _lib = L
    return _lib


def _hp(a):
    return a.ctypes.data_as(C.c_void_p)


def dev_ptr(t):
    """torch CUDA tensor -> void* for the *_dev entry points."""
    return C.c_void_p(t.data_ptr())


def make_reqs(n):
    r = np.zeros(n, FIELD_REQ_DTYPE)
    r["faction_id"] = FACTION_ID_NONE
    return r


class NavContext:
    """Device-resident navigation state of one map: the GPU counterpart of the planes of
    `struct nav_private` (nav_private.h:52) that the hot path reads."""

    def __init__(self, chunk_w, chunk_h, device=0):
        self._h = C.c_void_p()
        rc = lib().navhip_ctx_create(C.byref(self._h), chunk_w, chunk_h, device)
        if rc != OK:
            self._h = None
            raise NavHipError("navhip_ctx_create failed (%d): no MI355X visible?" % rc)
        self.w, self.h, self.device = chunk_w, chunk_h, device

    def close(self):
        if getattr(self, "_h", None):
            lib().navhip_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != OK:
            msg = lib().navhip_last_error(self._h)
            raise NavHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))

    def last_error(self):
        msg = lib().navhip_last_error(self._h)
        return msg.decode() if msg else ""

    @property
    def stream(self):
        return lib().navhip_stream(self._h)

    def sync(self):
        self._chk(lib().navhip_sync(self._h), "navhip_sync")

    # -- map state (N_CopyCostBasePacked / N_CopyBlockersPacked layouts, nav.c:2432,2470) ------
    def upload_plane(self, layer, plane, array):
        dt = np.uint8 if plane in (PLANE_COST_BASE, PLANE_FACTIONS) else np.uint16
        a = np.ascontiguousarray(array, dtype=dt)
        self._chk(lib().navhip_upload_plane(self._h, layer, plane, _hp(a), a.nbytes),
                  "navhip_upload_plane")

    def upload_chunk(self, layer, plane, chunk_r, chunk_c, array):
        dt = np.uint8 if plane in (PLANE_COST_BASE, PLANE_FACTIONS) else np.uint16
        a = np.ascontiguousarray(array, dtype=dt)
        self._chk(lib().navhip_upload_chunk(self._h, layer, plane, chunk_r, chunk_c, _hp(a),
                                            a.nbytes), "navhip_upload_chunk")

    def download_plane(self, layer, plane):
        dt = np.uint8 if plane in (PLANE_COST_BASE, PLANE_FACTIONS) else np.uint16
        shape = (self.h, self.w, 64, 64) if plane != PLANE_FACTIONS else (self.h, self.w, 15, 64, 64)
        out = np.zeros(shape, dt)
        self._chk(lib().navhip_download_plane(self._h, layer, plane, _hp(out), out.nbytes),
                  "navhip_download_plane")
        return out

    # -- dynamic obstacles (N_BlockersIncref / N_BlockersDecref, nav.c:4663,4685) -----------------
    def map_pos(self):
        return self.w * 128.0, -self.h * 128.0

    def N_BlockersUpdate(self, circles):
        """circles: CIRCLE_DTYPE records (delta +1 = N_BlockersIncref, -1 = N_BlockersDecref)."""
        c = np.ascontiguousarray(circles, dtype=CIRCLE_DTYPE)
        mx, mz = self.map_pos()
        self._chk(lib().navhip_blockers_circles(self._h, _hp(c), len(c), mx, mz),
                  "navhip_blockers_circles")

    def blockers_circles_dev(self, d_circles, n, stream=None):
        mx, mz = self.map_pos()
        self._chk(lib().navhip_blockers_circles_dev(self._h, dev_ptr(d_circles), n, mx, mz,
                                                    C.c_void_p(stream) if stream else None),
                  "navhip_blockers_circles_dev")

    def relabel_local_islands(self, layer=0):
        self._chk(lib().navhip_relabel_local_islands(self._h, layer), "navhip_relabel_local_islands")

    def changed_chunks(self, layer=0, clear=False):
        out = np.zeros(self.w * self.h, np.uint8)
        self._chk(lib().navhip_changed_chunks(self._h, layer, _hp(out), int(clear)),
                  "navhip_changed_chunks")
        return out.reshape(self.h, self.w)

    def clear_changed(self, stream=None):
        self._chk(lib().navhip_clear_changed(self._h, C.c_void_p(stream) if stream else None),
                  "navhip_clear_changed")

    def set_field_kernel(self, mode):
        self._chk(lib().navhip_set_field_kernel(self._h, mode), "navhip_set_field_kernel")

    # -- flow fields ----------------------------------------------------------------------------
    def N_FlowFieldUpdate(self, reqs, inout=None, want_integ=False):
        """Batched N_FlowFieldInit + N_FlowFieldUpdate (field.c:2020,2030) through host buffers.
        reqs: FIELD_REQ_DTYPE array.  inout: [n,64,64] u8 existing fields (rows used only for
        requests flagged REQ_INOUT).  Returns (dirs [n,64,64] u8, integ [n,64,64] f32 | None)."""
        reqs = np.ascontiguousarray(reqs, dtype=FIELD_REQ_DTYPE)
        n = len(reqs)
        dirs = np.zeros((n, 64, 64), np.uint8)
        if inout is not None:
            dirs[...] = np.asarray(inout, np.uint8).reshape(n, 64, 64)
        integ = np.zeros((n, 64, 64), np.float32) if want_integ else None
        self._chk(lib().navhip_build_fields(self._h, _hp(reqs), n, _hp(dirs),
                                            _hp(integ) if want_integ else None),
                  "navhip_build_fields")
        return dirs, integ

    def build_region_fields(self, reqs, seeds, overlay=None, inout=None, out_stride=None):
        """Region flow fields (N_CellArrivalFieldCreate / N_GroupArrivalFieldCreate in mode 0, the
        padded-region builders behind TARGET_ENEMIES / ENTITY / ZONE in mode 1).  seeds / overlay:
        [k, 2] int16 absolute (row, col) tiles.  Returns [n, out_stride] u8."""
        reqs = np.ascontiguousarray(reqs, dtype=REGION_REQ_DTYPE)
        n = len(reqs)
        seeds = np.ascontiguousarray(seeds, np.int16).reshape(-1, 2)
        ov = np.zeros((0, 2), np.int16) if overlay is None else \
            np.ascontiguousarray(overlay, np.int16).reshape(-1, 2)
        if out_stride is None:
            out_stride = 8192
        buf = np.zeros((n, out_stride), np.uint8)
        if inout is not None:
            a = np.asarray(inout, np.uint8).reshape(n, -1)
            buf[:, :a.shape[1]] = a
        self._chk(lib().navhip_build_region_fields(self._h, _hp(reqs), n, _hp(seeds), len(seeds),
                                                   _hp(ov), len(ov), _hp(buf), out_stride),
                  "navhip_build_region_fields")
        return buf

    def N_LOSFieldCreate(self, reqs, prev=None):
        """Batched N_LOSFieldCreate (field.c:2085).  reqs: LOS_REQ_DTYPE; prev: [n,64,64] u8 previous
        fields (bit0 visible, bit1 wavefront_blocked) or None.  Returns [n,64,64] u8."""
        reqs = np.ascontiguousarray(reqs, dtype=LOS_REQ_DTYPE)
        n = len(reqs)
        out = np.zeros((n, 64, 64), np.uint8)
        p = None if prev is None else np.ascontiguousarray(prev, np.uint8).reshape(n, 64, 64)
        mx, mz = self.map_pos()
        self._chk(lib().navhip_build_los(self._h, _hp(reqs), n, _hp(p) if p is not None else None,
                                         _hp(out), mx, mz), "navhip_build_los")
        return out

    def build_fields_dev(self, d_reqs, n, d_dirs, d_integ=None, stream=None):
        """Everything resident in HBM (torch tensors); asynchronous on `stream`."""
        self._chk(lib().navhip_build_fields_dev(
            self._h, dev_ptr(d_reqs), n, dev_ptr(d_dirs),
            dev_ptr(d_integ) if d_integ is not None else None,
            C.c_void_p(stream) if stream else None), "navhip_build_fields_dev")


def N_FlowFieldID(req):
    """N_FlowFieldID (field.c:1952) for one navhip_field_req record."""
    r = np.ascontiguousarray(np.asarray(req, dtype=FIELD_REQ_DTYPE).reshape(1))
    return int(lib().navhip_flow_field_id(_hp(r)))


FFID_ENEMIES, FFID_ENTITY, FFID_ZONE = 2, 4, 5


def N_RegionFieldID(kind, layer, chunk_r, chunk_c, a, b=0, c=0):
    """N_FlowFieldID for an ENEMIES / ENTITY / ZONE target (field.c:1976-2003)."""
    return int(lib().navhip_region_field_id(kind, layer, chunk_r, chunk_c, a, b, c))


# ---------------------------------------------------------------------------------------------
# per-agent movement step
# ---------------------------------------------------------------------------------------------
STATE_MOVING, STATE_MOVING_IN_FORMATION, STATE_ARRIVED, STATE_SEEK_ENEMIES, STATE_WAITING, \
    STATE_SURROUND_ENTITY, STATE_ENTER_ENTITY_RANGE, STATE_TURNING, STATE_ARRIVING_TO_CELL = range(9)
ENTITY_FLAG_MOVABLE = 1 << 3
ENTITY_FLAG_WATER = 1 << 14
ENTITY_FLAG_AIR = 1 << 15
ENTITY_FLAG_GARRISONED = 1 << 18
ENTITY_FLAG_COMBAT_HELD = 1 << 21
ST_MOVED, ST_FIELD_MISS, ST_FIELD_NONE, ST_LOS_MISS, ST_UNSUPPORTED = 0x01, 0x02, 0x04, 0x08, 0x80
LOS_LOOKUP = 0xFF


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
        ("arrival_sink_xz", C.c_void_p), ("arrival_flags", C.c_void_p),
        ("los_pool", C.c_void_p), ("flock_los_slot", C.c_void_p), ("los_pos_xz", C.c_void_p),
        ("n_los_slots", C.c_int32), ("static_epoch", C.c_uint32),
        ("region_row", C.c_void_p), ("region_field_slot", C.c_void_p), ("n_region_rows", C.c_int32)]


class StepOut(C.Structure):
    """navhip_step_out, include/navhip.h"""
    _fields_ = [("vel_xz", C.c_void_p), ("new_pos_xz", C.c_void_p), ("vdes_xz", C.c_void_p),
                ("vpref_xz", C.c_void_p), ("status", C.c_void_p)]


class StateIn(C.Structure):
    """navhip_state_in, include/navhip.h"""
    _fields_ = [("new_pos_xz", C.c_void_p), ("vdes_xz", C.c_void_p), ("skip", C.c_void_p),
                ("flock_layer", C.c_void_p), ("flock_nearest_xz", C.c_void_p), ("flock_tiles_off", C.c_void_p),
                ("flock_tiles", C.c_void_p)]


SU_SET_STATE, SU_BLOCK, SU_HOST = 0x01, 0x02, 0x80
GATE_TURN, GATE_HOST = 0x01, 0x80
SU_SET_MOVING, SU_TARGET_DIR, SU_SET_DEST, SU_SURROUND_DEST, SU_SURROUND_PREV = 0x04, 0x08, 0x10, 0x20, 0x40
SQ_ADJACENT, SQ_HAS_DEST_0, SQ_HAS_DEST_1 = 0x01, 0x02, 0x04
FS_MEMBER, FS_READY, FS_ASSIGNED, FS_IN_RANGE, FS_ARRIVED = 0x01, 0x02, 0x04, 0x08, 0x10


class StateAuxIn(C.Structure):
    """navhip_state_aux_in, include/navhip.h"""
    _fields_ = [("fstate", C.c_void_p), ("wait_ticks_left", C.c_void_p), ("wait_prev", C.c_void_p), ("new_pos_xz", C.c_void_p),
                ("ent_rot", C.c_void_p), ("target_dir", C.c_void_p), ("range_target", C.c_void_p), ("target_range", C.c_void_p),
                ("target_prev_xz", C.c_void_p), ("range_tiles_row", C.c_void_p), ("range_tiles_off", C.c_void_p),
                ("range_tiles", C.c_void_p), ("n_range_rows", C.c_int32),
                ("surround_target", C.c_void_p), ("surround_query", C.c_void_p), ("surround_target_prev_xz", C.c_void_p),
                ("surround_nearest_prev_xz", C.c_void_p), ("surround_dest_xz", C.c_void_p), ("vdes_xz", C.c_void_p),
                ("out_surround_dest_xz", C.c_void_p), ("sparse_units", C.c_void_p), ("n_sparse", C.c_int32)]


class GateIn(C.Structure):
    """navhip_gate_in, include/navhip.h"""
    _fields_ = [("next_rot", C.c_void_p), ("new_vel_xz", C.c_void_p), ("vdes_xz", C.c_void_p),
                ("interp_from_xz", C.c_void_p), ("interp_step", C.c_void_p)]


class StatePassIn(C.Structure):
    """navhip_state_pass_in, include/navhip.h"""
    _fields_ = [("gate", GateIn), ("state", StateIn), ("aux", StateAuxIn)]


class StatePassOut(C.Structure):
    """navhip_state_pass_out, include/navhip.h"""
    _fields_ = [("state", C.c_void_p), ("flags", C.c_void_p), ("gate", C.c_void_p), ("new_pos_xz", C.c_void_p),
                ("vel_xz", C.c_void_p), ("wait_ticks_left", C.c_void_p)]



class ArrivalZone(C.Structure):
    """navhip_arrival_zone, include/navhip.h"""
    _fields_ = [("centre_x", C.c_float), ("centre_z", C.c_float), ("unit_radius", C.c_float), ("fill_frac", C.c_float),
                ("radius", C.c_int32), ("layer", C.c_int32), ("active_row", C.c_int32), ("num_rows", C.c_int32),
                ("slot_begin", C.c_int32), ("slot_end", C.c_int32), ("key_begin", C.c_int32), ("key_end", C.c_int32)]


class SettleIn(C.Structure):
    """navhip_settle_in, include/navhip.h"""
    _fields_ = [("n_zones", C.c_int32), ("nq", C.c_int32), ("zones", C.c_void_p), ("slots_xz", C.c_void_p),
                ("slot_ring", C.c_void_p), ("region_keys", C.c_void_p), ("uid", C.c_void_p), ("zone", C.c_void_p),
                ("new_pos_xz", C.c_void_p), ("nsettled", C.c_void_p), ("substate", C.c_void_p),
                ("sink_valid", C.c_void_p), ("sink_xz", C.c_void_p), ("order_pos_xz", C.c_void_p),
                ("progress_anchor_xz", C.c_void_p), ("progress_anchored", C.c_void_p), ("stuck", C.c_void_p)]


class SettleOut(C.Structure):
    """navhip_settle_out, include/navhip.h"""
    _fields_ = [("settle", C.c_void_p), ("substate", C.c_void_p), ("progress_anchor_xz", C.c_void_p),
                ("progress_anchored", C.c_void_p), ("stuck", C.c_void_p), ("nsettled", C.c_void_p)]


_SIGS.update({
    "navhip_state_update": (C.c_int, [C.c_void_p, C.POINTER(World), C.POINTER(StateIn), C.c_void_p, C.c_void_p]),
    "navhip_state_update_dev": (C.c_int, [C.c_void_p, C.POINTER(World), C.POINTER(StateIn), C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "navhip_heading_gate": (C.c_int, [C.c_void_p, C.POINTER(World), C.POINTER(GateIn), C.c_void_p, C.c_void_p,
                                      C.c_void_p]),
    "navhip_heading_gate_dev": (C.c_int, [C.c_void_p, C.POINTER(World), C.POINTER(GateIn), C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_void_p]),
    "navhip_state_update_aux": (C.c_int, [C.c_void_p, C.POINTER(World), C.POINTER(StateAuxIn), C.c_void_p, C.c_void_p,
                                          C.c_void_p]),
    "navhip_state_update_aux_dev": (C.c_int, [C.c_void_p, C.POINTER(World), C.POINTER(StateAuxIn), C.c_void_p, C.c_void_p,
                                              C.c_void_p, C.c_void_p]),
    "navhip_state_pass": (C.c_int, [C.c_void_p, C.POINTER(World), C.POINTER(StatePassIn), C.POINTER(StatePassOut)]),
    "navhip_settled_count": (C.c_int, [C.c_void_p, C.POINTER(World), C.c_int, C.c_void_p, C.c_void_p]),
    "navhip_arrival_settle": (C.c_int, [C.c_void_p, C.POINTER(World), C.POINTER(SettleIn), C.POINTER(SettleOut)]),
    "navhip_arrival_settle_dev": (C.c_int, [C.c_void_p, C.POINTER(World), C.POINTER(SettleIn), C.POINTER(SettleOut),
                                            C.c_void_p]),
    "navhip_agent_step": (C.c_int, [C.c_void_p, C.POINTER(World), C.POINTER(StepOut)]),
    "navhip_agent_step_dev": (C.c_int, [C.c_void_p, C.POINTER(World), C.POINTER(StepOut), C.c_void_p]),
    "navhip_agent_prefetch_dev": (C.c_int, [C.c_void_p, C.POINTER(World), C.c_void_p]),
    "navhip_agent_prefetch_dev_ex": (C.c_int, [C.c_void_p, C.POINTER(World), C.c_void_p, C.c_uint32]),
    "navhip_spatial_query": (C.c_int, [C.c_void_p, C.POINTER(World), C.c_void_p, C.c_int, C.c_float,
                                       C.c_int, C.c_void_p, C.c_void_p]),
    "navhip_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "navhip_region_lookup": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                       C.c_int, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "navhip_stream_wait_stage": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "navhip_get_counters": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "navhip_stream_create_partial": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "navhip_stream_beside": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "navhip_stream_main": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "navhip_last_step_ms": (C.c_int, [C.c_void_p, C.POINTER(C.c_float * 5)]),
    "navhip_last_step_lists": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32 * 6)]),
    "navhip_step_lists_peek": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32 * 6)]),
    "navhip_clearpath_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "navhip_clearpath_team": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "navhip_clearpath": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
})

_SIGS.update({
    "navhip_pool_create": (C.c_int, [C.c_void_p, C.c_int, C.c_int]),
    "navhip_pool_destroy": (None, [C.c_void_p]),
    "navhip_pool_clear": (C.c_int, [C.c_void_p]),
    "navhip_pool_contains": (C.c_int, [C.c_void_p, C.c_uint64]),
    "navhip_pool_put": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "navhip_pool_get": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p]),
    "navhip_pool_invalidate": (C.c_int, [C.c_void_p, C.c_uint64]),
    "navhip_pool_build": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "navhip_pool_map": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "navhip_host_alloc": (C.c_void_p, [C.c_size_t]),
    "navhip_host_free": (None, [C.c_void_p]),
    "navhip_agent_step_submit": (C.c_int, [C.c_void_p, C.POINTER(World), C.POINTER(StepOut)]),
    "navhip_agent_step_poll": (C.c_int, [C.c_void_p]),
    "navhip_agent_step_wait": (C.c_int, [C.c_void_p]),
})
POOL_RESIDENT = -1

_WORLD_ARRAYS = (
    ("pos_xz", np.float32), ("vel_xz", np.float32), ("radius", np.float32),
    ("max_speed", np.float32), ("speed", np.float32), ("flags", np.uint32), ("state", np.uint8),
    ("has_dest_los", np.uint8), ("flock", np.int32), ("vdes_xz", np.float32),
    ("flock_target_xz", np.float32), ("flock_offsets", np.int32), ("flock_members", np.int32),
    ("flock_field_slot", np.int32), ("field_pool", np.uint8), ("form_ready", np.uint8),
    ("cell_pos_xz", np.float32), ("form_cohesion_xz", np.float32), ("form_align_xz", np.float32),
    ("form_drag_xz", np.float32), ("arrival_sink_xz", np.float32), ("arrival_flags", np.uint8),
    ("los_pool", np.uint8), ("flock_los_slot", np.int32), ("los_pos_xz", np.float32),
    ("region_row", np.int32), ("region_field_slot", np.int32))


def flock_csr(flock, n_flocks, order=None):
    """CSR member lists from a per-entity flock index (members in ascending uid order unless
    `order` -- a list of per-flock uid arrays, e.g. the reference's kh_foreach order -- is given)."""
    flock = np.asarray(flock)
    lists = order if order is not None else [np.flatnonzero(flock == f) for f in range(n_flocks)]
    offs = np.zeros(n_flocks + 1, np.int32)
    offs[1:] = np.cumsum([len(l) for l in lists])
    members = np.concatenate(lists).astype(np.int32) if n_flocks else np.zeros(0, np.int32)
    return offs, members


def _ctx_build_los_dev(self, d_reqs, n, d_prev, d_out, stream=None):
    """navhip_build_los_dev: n LOS fields, every buffer a torch CUDA tensor (d_prev may be None)."""
    mx, mz = self.map_pos()
    self._chk(lib().navhip_build_los_dev(self._h, dev_ptr(d_reqs), int(n), dev_ptr(d_prev) if d_prev is not None else None,
                                         dev_ptr(d_out), mx, mz, C.c_void_p(stream) if stream else None),
              "navhip_build_los_dev")


NavContext.build_los_dev = _ctx_build_los_dev


def grid_bounds(chunk_w, chunk_h):
    """bg_ent_init bounds the engine uses (position.c:276-283): the map, centred on the origin."""
    hx, hz = chunk_w * 128.0, chunk_h * 128.0
    return (-hx, hx, -hz, hz)


def make_world(chunk_w, chunk_h, arrays, hz=20, xp=None):
    """Build a navhip_world from a dict of numpy arrays (host form) or torch CUDA tensors
    (device form).  Returns (World, keepalive)."""
    w = World()
    keep = {}
    n = len(arrays["pos_xz"])
    w.n_ents = n
    w.n_flocks = len(arrays["flock_target_xz"]) if arrays.get("flock_target_xz") is not None else 0
    w.hz = hz
    for name, dt in _WORLD_ARRAYS:
        a = arrays.get(name)
        if a is None:
            setattr(w, name, None)
            continue
        if hasattr(a, "data_ptr"):          # torch tensor on the GPU
            keep[name] = a
            setattr(w, name, a.data_ptr())
        else:
            a = np.ascontiguousarray(a, dtype=dt)
            keep[name] = a
            setattr(w, name, a.ctypes.data)
    fp = arrays.get("field_pool")
    w.n_field_slots = 0 if fp is None else int(fp.shape[0])
    lp = arrays.get("los_pool")
    w.n_los_slots = 0 if lp is None else int(lp.shape[0])
    rs = arrays.get("region_field_slot")
    w.n_region_rows = int(arrays.get("n_region_rows") or (0 if rs is None else int(rs.shape[0])))
    w.map_pos_x = chunk_w * 128.0
    w.map_pos_z = -chunk_h * 128.0
    w.grid_xmin, w.grid_xmax, w.grid_zmin, w.grid_zmax = grid_bounds(chunk_w, chunk_h)
    w.static_epoch = int(arrays.get("static_epoch") or 0)
    return w, keep


def _ctx_agent_step(self, arrays, hz=20, want=("vel_xz", "new_pos_xz", "vdes_xz", "vpref_xz", "status")):
    """Host-buffer velocity step: move_velocity_work (movement.c:3395) for every non-still entity.
    arrays: dict of numpy arrays named after navhip_world members.  Returns dict of outputs."""
    w, keep = make_world(self.w, self.h, arrays, hz)
    if arrays.get("field_pool") is None and arrays.get("use_resident_pool"):
        w.n_field_slots = POOL_RESIDENT
    n = w.n_ents
    out = {}
    so = StepOut()
    for name in ("vel_xz", "new_pos_xz", "vdes_xz", "vpref_xz"):
        if name in want or name == "vel_xz":
            out[name] = np.zeros((n, 2), np.float32)
            setattr(so, name, out[name].ctypes.data)
    if "status" in want:
        out["status"] = np.zeros(n, np.uint8)
        so.status = out["status"].ctypes.data
    self._chk(lib().navhip_agent_step(self._h, C.byref(w), C.byref(so)), "navhip_agent_step")
    return out


def _ctx_agent_step_dev(self, world, stepout, stream=None):
    self._chk(lib().navhip_agent_step_dev(self._h, C.byref(world), C.byref(stepout),
                                          C.c_void_p(stream) if stream else None),
              "navhip_agent_step_dev")


PREFETCH_FRONT_INLINE, PREFETCH_SNAPSHOT_HELD, PREFETCH_FOLLOWS_STEP = 1, 2, 4


def _ctx_agent_prefetch_dev(self, world, stream=None, flags=0):
    self._chk(lib().navhip_agent_prefetch_dev_ex(self._h, C.byref(world),
                                                 C.c_void_p(stream) if stream else None, flags),
              "navhip_agent_prefetch_dev")


def _ctx_spatial_query(self, pos_xz, query_xz, rng, maxout):
    """G_Pos_EntsInCircleFrom candidate lists (bitmap_grid.h:1376 order) for each query."""
    w, keep = make_world(self.w, self.h, {"pos_xz": np.ascontiguousarray(pos_xz, np.float32)})
    q = np.ascontiguousarray(query_xz, np.float32).reshape(-1, 2)
    counts = np.zeros(len(q), np.int32)
    ids = np.zeros((len(q), maxout), np.uint32)
    self._chk(lib().navhip_spatial_query(self._h, C.byref(w), _hp(q), len(q), rng, maxout,
                                         _hp(counts), _hp(ids)), "navhip_spatial_query")
    return counts, ids


def _ctx_clearpath(self, ent, des_v, dyn, n_dyn, stat, n_stat, rows=False):
    """G_ClearPath_NewVelocity (clearpath.c:694) for a batch of independent problems, one wave per
    problem; rows=True: one row of 16 lanes per problem (<= 16 neighbours); rows="team": the waves of a
    workgroup per problem."""
    ent = np.ascontiguousarray(ent, np.float32).reshape(-1, 5)
    nq = len(ent)
    des_v = np.ascontiguousarray(des_v, np.float32).reshape(nq, 2)
    dyn = np.ascontiguousarray(dyn, np.float32).reshape(nq, 32, 5)
    stat = np.ascontiguousarray(stat, np.float32).reshape(nq, 32, 5)
    n_dyn = np.ascontiguousarray(n_dyn, np.int32)
    n_stat = np.ascontiguousarray(n_stat, np.int32)
    out = np.zeros((nq, 2), np.float32)
    if rows == "team":
        self._chk(lib().navhip_clearpath_team(self._h, nq, _hp(ent), _hp(des_v), _hp(dyn), _hp(n_dyn),
                                              _hp(stat), _hp(n_stat), _hp(out)), "navhip_clearpath_team")
        return out
    if rows:
        self._chk(lib().navhip_clearpath_rows(self._h, nq, _hp(ent), _hp(des_v), _hp(dyn), _hp(n_dyn),
                                              _hp(stat), _hp(n_stat), _hp(out)), "navhip_clearpath_rows")
        return out
    self._chk(lib().navhip_clearpath(self._h, nq, _hp(ent), _hp(des_v), _hp(dyn), _hp(n_dyn),
                                     _hp(stat), _hp(n_stat), _hp(out)), "navhip_clearpath")
    return out


STAGE_NEIGHBOURS, STAGE_LISTS, STAGE_START, STAGE_END = 0, 1, 2, 3


def _ctx_stream_wait_stage(self, stream, stage, check=True):
    """Make `stream` (a hipStream_t value) wait for a stage of the agent step in flight.  check=False: return whether
    the library could do so instead of raising (NAVHIP_STAGE_END after a step that ran on one stream: it cannot)."""
    rc = lib().navhip_stream_wait_stage(self._h, C.c_void_p(stream), stage)
    if check:
        self._chk(rc, "navhip_stream_wait_stage")
    return rc == 0


COUNTER_NAMES = ("field_calls", "chunk_fields", "step_calls", "agent_steps", "los_fields", "region_fields",
                 "blocker_circles")


def _ctx_counters(self, reset=False):
    """navhip_get_counters: work counters of the context as a dict."""
    out = (C.c_uint64 * len(COUNTER_NAMES))()
    self._chk(lib().navhip_get_counters(self._h, out, int(bool(reset))), "navhip_get_counters")
    return dict(zip(COUNTER_NAMES, [int(x) for x in out]))


def _ctx_stream_create_partial(self, cu_begin, cu_count):
    """A hipStream_t value restricted to the compute units [cu_begin, cu_begin + cu_count)."""
    out = C.c_void_p()
    self._chk(lib().navhip_stream_create_partial(self._h, cu_begin, cu_count, C.byref(out)), "navhip_stream_create_partial")
    return out.value


def _ctx_stream_beside(self, main_stream, cu_begin=0, cu_count=0):
    """navhip_stream_beside: the library's stream for wide work beside a step on `main_stream` (a hipStream_t value);
    cu_count > 0 restricts it to the compute units [cu_begin, cu_begin + cu_count)."""
    out = C.c_void_p()
    self._chk(lib().navhip_stream_beside(self._h, C.c_void_p(main_stream), cu_begin, cu_count, C.byref(out)), "navhip_stream_beside")
    return out.value


def _ctx_stream_main(self):
    """navhip_stream_main: the library's own stream for the agent chain (a hipStream_t value)."""
    out = C.c_void_p()
    self._chk(lib().navhip_stream_main(self._h, C.byref(out)), "navhip_stream_main")
    return out.value


def _ctx_set_profiling(self, on):
    self._chk(lib().navhip_set_profiling(self._h, int(bool(on))), "navhip_set_profiling")


STEP_PHASES = ("sp_build", "agent_nbr", "cohesion", "coh_regroup", "agent_finish")


def _ctx_last_step_ms(self):
    """Milliseconds of the kernel groups STEP_PHASES of the last profiled agent step."""
    out = (C.c_float * 5)()
    self._chk(lib().navhip_last_step_ms(self._h, C.byref(out)), "navhip_last_step_ms")
    return tuple(float(x) for x in out)


def _ctx_last_step_lists(self):
    """Agents per ClearPath work list of the last step: light 1..4 neighbours, wave, full-wave."""
    out = (C.c_int32 * 6)()
    self._chk(lib().navhip_last_step_lists(self._h, C.byref(out)), "navhip_last_step_lists")
    return tuple(int(x) for x in out)


def _ids_of(reqs):
    reqs = np.ascontiguousarray(reqs, dtype=FIELD_REQ_DTYPE)
    return np.array([lib().navhip_flow_field_id(_hp(reqs[i:i + 1])) for i in range(len(reqs))], np.uint64)


def _ctx_pool_create(self, n_slots, n_dests):
    self._chk(lib().navhip_pool_create(self._h, n_slots, n_dests), "navhip_pool_create")


def _ctx_pool_build(self, reqs, ff_ids=None, base_ids=None, readback=True):
    """Batched N_FlowFieldInit + N_FlowFieldUpdate + N_FC_PutFlowField into the resident pool; ids
    default to N_FlowFieldID of every request.  Returns (ids, dirs [n,64,64] | None)."""
    reqs = np.ascontiguousarray(reqs, dtype=FIELD_REQ_DTYPE)
    n = len(reqs)
    ids = _ids_of(reqs) if ff_ids is None else np.ascontiguousarray(ff_ids, np.uint64)
    base = None if base_ids is None else np.ascontiguousarray(base_ids, np.uint64)
    out = np.zeros((n, 64, 64), np.uint8) if readback else None
    self._chk(lib().navhip_pool_build(self._h, _hp(reqs), _hp(ids), _hp(base) if base is not None else None, n,
                                      _hp(out) if readback else None), "navhip_pool_build")
    return ids, out


def _ctx_pool_put(self, ff_id, dirs):
    d = np.ascontiguousarray(dirs, np.uint8).reshape(4096)
    self._chk(lib().navhip_pool_put(self._h, int(ff_id), _hp(d)), "navhip_pool_put")


def _ctx_pool_get(self, ff_id):
    d = np.zeros((64, 64), np.uint8)
    rc = lib().navhip_pool_get(self._h, int(ff_id), _hp(d))
    return None if rc != OK else d


def _ctx_region_lookup(self, pos_xz, rows, region_field_slot=None, field_pool=None, centre_abs=None, radius=None):
    """N_DesiredGroupArrivalVelocity for many points: (dir [nq] u8 with 0xff = no field, at_slot [nq] | None)."""
    p = np.ascontiguousarray(pos_xz, np.float32).reshape(-1, 2)
    nq = len(p)
    r = np.ascontiguousarray(rows, np.int32)
    tbl = None if region_field_slot is None else np.ascontiguousarray(region_field_slot, np.int32)
    fp = None if field_pool is None else np.ascontiguousarray(field_pool, np.uint8).reshape(-1, 4096)
    cen = None if centre_abs is None else np.ascontiguousarray(centre_abs, np.int32).reshape(nq, 2)
    rad = None if radius is None else np.ascontiguousarray(radius, np.int32)
    out = np.zeros(nq, np.uint8)
    at = np.zeros(nq, np.uint8) if cen is not None else None
    self._chk(lib().navhip_region_lookup(
        self._h, nq, _hp(p), _hp(r), _hp(tbl) if tbl is not None else None, 0 if tbl is None else len(tbl),
        _hp(fp) if fp is not None else None, 0 if fp is None else len(fp), _hp(cen) if cen is not None else None,
        _hp(rad) if rad is not None else None, self.w * 128.0, -self.h * 128.0, _hp(out),
        _hp(at) if at is not None else None), "navhip_region_lookup")
    return out, at


def _ctx_state_update(self, arrays, new_pos_xz, vdes_xz, flock_layer, flock_nearest_xz, flock_tiles, skip=None,
                      hz=20, work=None):
    """The arrival arm of entity_compute_update (movement.c:2303) for every unit of the snapshot `arrays`.
    flock_tiles: list of [k, 2] int16 arrays (absolute (row, col) tiles per flock).  Returns (next_state, flags)."""
    w, keep = make_world(self.w, self.h, arrays, hz)
    if work is not None:
        w.work_begin, w.work_end = work
    n = w.n_ents
    si = StateIn()
    k = {"np": np.ascontiguousarray(new_pos_xz, np.float32).reshape(n, 2),
         "vd": np.ascontiguousarray(vdes_xz, np.float32).reshape(n, 2),
         "fl": np.ascontiguousarray(flock_layer, np.uint8),
         "fn": np.ascontiguousarray(flock_nearest_xz, np.float32).reshape(-1, 2)}
    offs = np.zeros(len(flock_tiles) + 1, np.int32)
    offs[1:] = np.cumsum([len(t) for t in flock_tiles])
    tiles = np.concatenate([np.asarray(t, np.int16).reshape(-1, 2) for t in flock_tiles] + [np.zeros((1, 2), np.int16)])
    k["to"], k["tt"] = offs, np.ascontiguousarray(tiles)
    si.new_pos_xz, si.vdes_xz = k["np"].ctypes.data, k["vd"].ctypes.data
    si.flock_layer, si.flock_nearest_xz = k["fl"].ctypes.data, k["fn"].ctypes.data
    si.flock_tiles_off, si.flock_tiles = k["to"].ctypes.data, k["tt"].ctypes.data
    if skip is not None:
        k["sk"] = np.ascontiguousarray(skip, np.uint8)
        si.skip = k["sk"].ctypes.data
    st, fl = np.zeros(n, np.uint8), np.zeros(n, np.uint8)
    self._chk(lib().navhip_state_update(self._h, C.byref(w), C.byref(si), _hp(st), _hp(fl)), "navhip_state_update")
    return st, fl


def _surround_arrays(n, su):
    """su: dict(target [n] row or -1 / -2, query [n] SQ_*, target_prev_xz [n][2], nearest_prev_xz [n][2], dest_xz [n][2][2])
    -> the five contiguous arrays of navhip_state_aux_in + the output array."""
    return [np.ascontiguousarray(su["target"], np.int32), np.ascontiguousarray(su["query"], np.uint8),
            np.ascontiguousarray(su["target_prev_xz"], np.float32).reshape(n, 2),
            np.ascontiguousarray(su["nearest_prev_xz"], np.float32).reshape(n, 2),
            np.ascontiguousarray(su["dest_xz"], np.float32).reshape(n, 2, 2), np.zeros((n, 2), np.float32)]


def _ctx_heading_gate(self, arrays, next_rot, new_vel_xz, vdes_xz, work=None, hz=20, interp=None):
    """The heading gate of entity_compute_update (movement.c:2319-2336) for the units of the snapshot `arrays`
    (pos_xz, vel_xz, state).  Returns (velocity after the gate [n][2], new_pos [n][2], gate flags [n])."""
    w, keep = make_world(self.w, self.h, arrays, hz=hz)
    if work is not None:
        w.work_begin, w.work_end = work
    n = w.n_ents
    k = [np.ascontiguousarray(next_rot, np.float32).reshape(n, 4), np.ascontiguousarray(new_vel_xz, np.float32).reshape(n, 2),
         np.ascontiguousarray(vdes_xz, np.float32).reshape(n, 2)]
    gi = GateIn(k[0].ctypes.data, k[1].ctypes.data, k[2].ctypes.data)
    if interp is not None:          # (movestate.next_pos xz [n][2], movestate.step [n]): a rate below 20 Hz
        k += [np.ascontiguousarray(interp[0], np.float32).reshape(n, 2), np.ascontiguousarray(interp[1], np.float32)]
        gi.interp_from_xz, gi.interp_step = k[-2].ctypes.data, k[-1].ctypes.data
    vel, pos, gate = np.zeros((n, 2), np.float32), np.zeros((n, 2), np.float32), np.zeros(n, np.uint8)
    self._chk(lib().navhip_heading_gate(self._h, C.byref(w), C.byref(gi), _hp(vel), _hp(pos), _hp(gate)),
              "navhip_heading_gate")
    return vel, pos, gate


def _ctx_state_update_aux(self, arrays, fstate, wait_ticks_left, wait_prev, new_pos_xz, state, flags, work=None,
                          ent_rot=None, target_dir=None, range_in=None, surround=None, vdes_xz=None, hz=20):
    """The flag / counter arms of the state switch, after state_update on the same slab: returns (state, flags,
    wait_ticks_left) with the rows this pass decides overwritten -- and, with `surround` (see _surround_arrays; needs
    vdes_xz), the positions of NAVHIP_SU_SURROUND_PREV rows as a fourth value."""
    w, keep = make_world(self.w, self.h, arrays, hz=hz)
    if work is not None:
        w.work_begin, w.work_end = work
    n = w.n_ents
    k = [np.ascontiguousarray(fstate, np.uint8), np.ascontiguousarray(wait_ticks_left, np.int32),
         np.ascontiguousarray(wait_prev, np.uint8), np.ascontiguousarray(new_pos_xz, np.float32).reshape(n, 2)]
    ai = StateAuxIn(*[a.ctypes.data for a in k])
    if ent_rot is not None:
        k += [np.ascontiguousarray(ent_rot, np.float32).reshape(n, 4), np.ascontiguousarray(target_dir, np.float32).reshape(n, 4)]
        ai.ent_rot, ai.target_dir = k[-2].ctypes.data, k[-1].ctypes.data
    if range_in is not None:
        # range_in: dict(target [n] row or -1 / -2, range [n], prev_xz [n][2], tiles_row [n], tiles: list of [k, 2] int16)
        offs = np.zeros(len(range_in["tiles"]) + 1, np.int32)
        offs[1:] = np.cumsum([len(t) for t in range_in["tiles"]])
        tiles = np.concatenate([np.asarray(t, np.int16).reshape(-1, 2) for t in range_in["tiles"]] + [np.zeros((1, 2), np.int16)])
        r = [np.ascontiguousarray(range_in["target"], np.int32), np.ascontiguousarray(range_in["range"], np.float32),
             np.ascontiguousarray(range_in["prev_xz"], np.float32).reshape(n, 2), np.ascontiguousarray(range_in["tiles_row"], np.int32),
             offs, np.ascontiguousarray(tiles)]
        k += r
        ai.range_target, ai.target_range, ai.target_prev_xz, ai.range_tiles_row, ai.range_tiles_off, ai.range_tiles = \
            [a.ctypes.data for a in r]
        ai.n_range_rows = len(range_in["tiles"])
    if surround is not None:
        sa = _surround_arrays(n, surround) + [np.ascontiguousarray(vdes_xz, np.float32).reshape(n, 2)]
        k += sa
        ai.surround_target, ai.surround_query, ai.surround_target_prev_xz, ai.surround_nearest_prev_xz, ai.surround_dest_xz, \
            ai.out_surround_dest_xz, ai.vdes_xz = [a.ctypes.data for a in sa]
    st, fl, ticks = np.array(state, np.uint8), np.array(flags, np.uint8), np.zeros(n, np.int32)
    self._chk(lib().navhip_state_update_aux(self._h, C.byref(w), C.byref(ai), _hp(st), _hp(fl), _hp(ticks)),
              "navhip_state_update_aux")
    if surround is not None:
        return st, fl, ticks, sa[5]
    return st, fl, ticks


def _ctx_state_pass(self, arrays, next_rot, new_vel_xz, vdes_xz, flock_layer, flock_nearest_xz, flock_tiles, skip=None,
                    aux=None, work=None, hz=20, interp=None):
    """The state half of the tick in one call (navhip_state_pass): heading gate -> state update -> flag / counter arms.
    aux: dict(fstate, wait_ticks_left, wait_prev[, ent_rot, target_dir][, range_in]) or None.  Returns a dict of the
    outputs (state, flags, gate, new_pos_xz, vel_xz, wait_ticks_left[, surround_dest_xz with aux["surround"]]).
    interp: (movestate.next_pos xz, movestate.step) at a rate below 20 Hz."""
    w, keep = make_world(self.w, self.h, arrays, hz=hz)
    if work is not None:
        w.work_begin, w.work_end = work
    n = w.n_ents
    f32 = lambda a, width: np.ascontiguousarray(a, np.float32).reshape(n, width)
    k = [f32(next_rot, 4), f32(new_vel_xz, 2), f32(vdes_xz, 2), np.ascontiguousarray(flock_layer, np.uint8),
         np.ascontiguousarray(flock_nearest_xz, np.float32).reshape(-1, 2)]
    offs = np.zeros(len(flock_tiles) + 1, np.int32)
    offs[1:] = np.cumsum([len(t) for t in flock_tiles])
    tiles = np.ascontiguousarray(np.concatenate([np.asarray(t, np.int16).reshape(-1, 2) for t in flock_tiles] + [np.zeros((1, 2), np.int16)]))
    k += [offs, tiles]
    pi = StatePassIn()
    pi.gate = GateIn(k[0].ctypes.data, k[1].ctypes.data, k[2].ctypes.data)
    if interp is not None:
        k += [f32(interp[0], 2), np.ascontiguousarray(interp[1], np.float32)]
        pi.gate.interp_from_xz, pi.gate.interp_step = k[-2].ctypes.data, k[-1].ctypes.data
    pi.state.flock_layer, pi.state.flock_nearest_xz = k[3].ctypes.data, k[4].ctypes.data
    pi.state.flock_tiles_off, pi.state.flock_tiles = offs.ctypes.data, tiles.ctypes.data
    if skip is not None:
        k.append(np.ascontiguousarray(skip, np.uint8))
        pi.state.skip = k[-1].ctypes.data
    if aux is not None:
        a = [np.ascontiguousarray(aux["fstate"], np.uint8), np.ascontiguousarray(aux["wait_ticks_left"], np.int32),
             np.ascontiguousarray(aux["wait_prev"], np.uint8)]
        k += a
        pi.aux.fstate, pi.aux.wait_ticks_left, pi.aux.wait_prev = [x.ctypes.data for x in a]
        if aux.get("ent_rot") is not None:
            r = [f32(aux["ent_rot"], 4), f32(aux["target_dir"], 4)]
            k += r
            pi.aux.ent_rot, pi.aux.target_dir = r[0].ctypes.data, r[1].ctypes.data
        ri = aux.get("range_in")
        if ri is not None:
            ro = np.zeros(len(ri["tiles"]) + 1, np.int32)
            ro[1:] = np.cumsum([len(t) for t in ri["tiles"]])
            rt = np.ascontiguousarray(np.concatenate([np.asarray(t, np.int16).reshape(-1, 2) for t in ri["tiles"]] + [np.zeros((1, 2), np.int16)]))
            r = [np.ascontiguousarray(ri["target"], np.int32), np.ascontiguousarray(ri["range"], np.float32), f32(ri["prev_xz"], 2),
                 np.ascontiguousarray(ri["tiles_row"], np.int32), ro, rt]
            k += r
            pi.aux.range_target, pi.aux.target_range, pi.aux.target_prev_xz, pi.aux.range_tiles_row, pi.aux.range_tiles_off, \
                pi.aux.range_tiles = [x.ctypes.data for x in r]
            pi.aux.n_range_rows = len(ri["tiles"])
        su_out = None
        if aux.get("surround") is not None:
            sa = _surround_arrays(n, aux["surround"])
            k += sa
            pi.aux.surround_target, pi.aux.surround_query, pi.aux.surround_target_prev_xz, pi.aux.surround_nearest_prev_xz, \
                pi.aux.surround_dest_xz, pi.aux.out_surround_dest_xz = [x.ctypes.data for x in sa]
            su_out = sa[5]
    res = {"state": np.zeros(n, np.uint8), "flags": np.zeros(n, np.uint8), "gate": np.zeros(n, np.uint8),
           "new_pos_xz": np.zeros((n, 2), np.float32), "vel_xz": np.zeros((n, 2), np.float32),
           "wait_ticks_left": np.zeros(n, np.int32)}
    po = StatePassOut(*[res[f].ctypes.data for f in ("state", "flags", "gate", "new_pos_xz", "vel_xz", "wait_ticks_left")])
    self._chk(lib().navhip_state_pass(self._h, C.byref(w), C.byref(pi), C.byref(po)), "navhip_state_pass")
    if aux is not None and aux.get("surround") is not None:
        res["surround_dest_xz"] = su_out
    return res


def _ctx_settled_count(self, arrays, uids):
    """adjacent_settled_count (movement.c:982) for the units `uids` of the snapshot `arrays` (pos_xz, radius,
    flags, state); -1 = the host counts (radius > 12.5)."""
    w, keep = make_world(self.w, self.h, arrays)
    u = np.ascontiguousarray(uids, np.int32)
    out = np.zeros(len(u), np.int32)
    self._chk(lib().navhip_settled_count(self._h, C.byref(w), len(u), _hp(u), _hp(out)), "navhip_settled_count")
    return out


def _ctx_arrival_settle(self, arrays, zones, region_keys, units):
    """G_Arrival_ShouldSettle (arrival.c:946).  zones: list of dicts (layer, centre_xz, radius, unit_radius,
    fill_frac, active_row, num_rows, slots_xz, slot_ring), region_keys: list of sorted u64 arrays per zone;
    units: dict of nq-row arrays (uid, zone, new_pos_xz, nsettled, substate, sink_valid, sink_xz, order_pos_xz,
    progress_anchor_xz, progress_anchored, stuck).  Returns (settle [nq], dict of the unit state after)."""
    w, keep = make_world(self.w, self.h, arrays)
    zs = (ArrivalZone * len(zones))()
    slots, rings, keys = [], [], []
    so = ko = 0
    for i, z in enumerate(zones):
        sl = np.asarray(z["slots_xz"], np.float32).reshape(-1, 2)
        kk = np.asarray(region_keys[i], np.uint64)
        zs[i] = ArrivalZone(float(z["centre_xz"][0]), float(z["centre_xz"][1]), float(z["unit_radius"]),
                            float(z["fill_frac"]), int(z["radius"]), int(z["layer"]), int(z["active_row"]),
                            int(z["num_rows"]), so, so + len(sl), ko, ko + len(kk))
        so += len(sl); ko += len(kk)
        slots.append(sl); rings.append(np.asarray(z["slot_ring"], np.int32)); keys.append(kk)
    cat = lambda parts, dt, shape: np.ascontiguousarray(np.concatenate(parts + [np.zeros(shape, dt)]))
    k = {"slots": cat(slots, np.float32, (1, 2)), "ring": cat(rings, np.int32, (1,)), "keys": cat(keys, np.uint64, (1,))}
    nq = len(units["uid"])
    spec = (("uid", np.int32, 1), ("zone", np.int32, 1), ("new_pos_xz", np.float32, 2), ("nsettled", np.int32, 1),
            ("substate", np.uint8, 1), ("sink_valid", np.uint8, 1), ("sink_xz", np.float32, 2),
            ("order_pos_xz", np.float32, 2), ("progress_anchor_xz", np.float32, 2), ("progress_anchored", np.uint8, 1),
            ("stuck", np.int32, 1))
    si = SettleIn()
    si.n_zones, si.nq = len(zones), nq
    si.zones = C.addressof(zs)
    si.slots_xz, si.slot_ring, si.region_keys = k["slots"].ctypes.data, k["ring"].ctypes.data, k["keys"].ctypes.data
    for name, dt, width in spec:
        a = np.ascontiguousarray(units[name], dt).reshape((nq, width) if width > 1 else (nq,))
        k[name] = a
        setattr(si, name, a.ctypes.data)
    res = {"settle": np.zeros(nq, np.uint8), "substate": np.zeros(nq, np.uint8),
           "progress_anchor_xz": np.zeros((nq, 2), np.float32), "progress_anchored": np.zeros(nq, np.uint8),
           "stuck": np.zeros(nq, np.int32)}
    so_ = SettleOut(*[res[f].ctypes.data for f in ("settle", "substate", "progress_anchor_xz", "progress_anchored", "stuck")])
    self._chk(lib().navhip_arrival_settle(self._h, C.byref(w), C.byref(si), C.byref(so_)), "navhip_arrival_settle")
    settle = res.pop("settle")
    return settle, res


def _ctx_pool_invalidate(self, ff_id):
    self._chk(lib().navhip_pool_invalidate(self._h, int(ff_id)), "navhip_pool_invalidate")


def _ctx_pool_contains(self, ff_id):
    return bool(lib().navhip_pool_contains(self._h, int(ff_id)))


def _ctx_pool_map(self, dest, chunk_r, chunk_c, ff_ids):
    d = np.ascontiguousarray(dest, np.int32)
    r = np.ascontiguousarray(chunk_r, np.uint16)
    c = np.ascontiguousarray(chunk_c, np.uint16)
    i = np.ascontiguousarray(ff_ids, np.uint64)
    self._chk(lib().navhip_pool_map(self._h, len(d), _hp(d), _hp(r), _hp(c), _hp(i)), "navhip_pool_map")


def host_alloc(nbytes):
    """navhip_host_alloc: pinned host memory as a writable buffer (freed with host_free(buf))."""
    p = lib().navhip_host_alloc(nbytes)
    if not p:
        raise MemoryError("navhip_host_alloc(%d)" % nbytes)
    buf = (C.c_uint8 * nbytes).from_address(p)
    buf._navhip_ptr = p
    return buf


def host_free(buf):
    lib().navhip_host_free(C.c_void_p(buf._navhip_ptr))


def _ctx_agent_step_async(self, arrays, hz=20, work=None, want=("vel_xz", "new_pos_xz", "status"), spin=None):
    """navhip_agent_step_submit + _poll: returns the outputs once the step has completed; `spin`
    is called while it is still running (what the nav task does between submit and join)."""
    w, keep = make_world(self.w, self.h, arrays, hz)
    if arrays.get("field_pool") is None and arrays.get("use_resident_pool"):
        w.n_field_slots = POOL_RESIDENT
    if work is not None:
        w.work_begin, w.work_end = work
    w.static_epoch = int(arrays.get("static_epoch") or 0)
    n = w.n_ents
    out = {}
    so = StepOut()
    for name in ("vel_xz", "new_pos_xz", "vdes_xz", "vpref_xz"):
        if name in want or name == "vel_xz":
            out[name] = np.zeros((n, 2), np.float32)
            setattr(so, name, out[name].ctypes.data)
    if "status" in want:
        out["status"] = np.zeros(n, np.uint8)
        so.status = out["status"].ctypes.data
    self._chk(lib().navhip_agent_step_submit(self._h, C.byref(w), C.byref(so)), "navhip_agent_step_submit")
    polls = 0
    while True:
        rc = lib().navhip_agent_step_poll(self._h)
        if rc == 0:
            break
        if rc < 0:
            self._chk(rc, "navhip_agent_step_poll")
        polls += 1
        if spin:
            spin()
    out["polls"] = polls
    return out


NavContext.pool_create = _ctx_pool_create
NavContext.pool_build = _ctx_pool_build
NavContext.pool_put = _ctx_pool_put
NavContext.pool_get = _ctx_pool_get
def _ctx_comm_init(self, rank, world, uid):
    """navhip_comm_init: this context joins the RCCL communicator named by `uid` (comm_unique_id())."""
    buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(uid)
    self._chk(lib().navhip_comm_init(self._h, int(rank), int(world), buf), "navhip_comm_init")


def _ctx_comm_allgather_step_dev(self, d_new_pos, d_vel, bounds, stream=None):
    b = np.ascontiguousarray(bounds, np.int32)
    self._chk(lib().navhip_comm_allgather_step_dev(self._h, dev_ptr(d_new_pos), dev_ptr(d_vel), _hp(b),
                                                   C.c_void_p(stream) if stream else None),
              "navhip_comm_allgather_step_dev")


def _ctx_comm_allgather_rows_dev(self, d_rows, row_bytes, bounds, stream=None):
    b = np.ascontiguousarray(bounds, np.int32)
    self._chk(lib().navhip_comm_allgather_rows_dev(self._h, dev_ptr(d_rows), int(row_bytes), _hp(b),
                                                   C.c_void_p(stream) if stream else None),
              "navhip_comm_allgather_rows_dev")


def _ctx_comm_init_mailbox(self, rank, world, d_mailbox):
    """navhip_comm_init_mailbox: the exchange step over a device buffer instead of RCCL (bring-up, tests)."""
    self._chk(lib().navhip_comm_init_mailbox(self._h, int(rank), int(world), dev_ptr(d_mailbox),
                                             int(d_mailbox.numel() * d_mailbox.element_size())), "navhip_comm_init_mailbox")


NavContext.comm_init = _ctx_comm_init
NavContext.comm_init_mailbox = _ctx_comm_init_mailbox
NavContext.comm_destroy = lambda self: lib().navhip_comm_destroy(self._h)
NavContext.comm_world = lambda self: int(lib().navhip_comm_world(self._h))
NavContext.comm_allgather_step_dev = _ctx_comm_allgather_step_dev
NavContext.comm_allgather_rows_dev = _ctx_comm_allgather_rows_dev
NavContext.pool_contains = _ctx_pool_contains
NavContext.pool_invalidate = _ctx_pool_invalidate
NavContext.region_lookup = _ctx_region_lookup
NavContext.state_update = _ctx_state_update
NavContext.heading_gate = _ctx_heading_gate
NavContext.state_pass = _ctx_state_pass
NavContext.state_update_aux = _ctx_state_update_aux
NavContext.settled_count = _ctx_settled_count
NavContext.arrival_settle = _ctx_arrival_settle
NavContext.pool_map = _ctx_pool_map
NavContext.agent_step_async = _ctx_agent_step_async
NavContext.set_profiling = _ctx_set_profiling
NavContext.last_step_ms = _ctx_last_step_ms
def _ctx_step_lists_peek(self):
    """Like last_step_lists, without waiting: the counts of the latest step whose copy has arrived."""
    out = (C.c_int32 * 6)()
    self._chk(lib().navhip_step_lists_peek(self._h, C.byref(out)), "navhip_step_lists_peek")
    return list(out)


NavContext.last_step_lists = _ctx_last_step_lists
NavContext.step_lists_peek = _ctx_step_lists_peek
NavContext.stream_wait_stage = _ctx_stream_wait_stage
NavContext.stream_create_partial = _ctx_stream_create_partial
NavContext.stream_beside = _ctx_stream_beside
NavContext.stream_main = _ctx_stream_main
NavContext.counters = _ctx_counters
NavContext.agent_step = _ctx_agent_step
NavContext.agent_step_dev = _ctx_agent_step_dev
NavContext.agent_prefetch_dev = _ctx_agent_prefetch_dev
NavContext.spatial_query = _ctx_spatial_query
NavContext.G_ClearPath_NewVelocity = _ctx_clearpath


# ---------------------------------------------------------------------------------------------
# the whole tick behind one call (navhip_tick_*, csrc/tick_api.hip)
# ---------------------------------------------------------------------------------------------
TICK_SERIAL, TICK_TIME_FIELDS, TICK_OWNS_SNAPSHOT = 0x2, 0x8, 0x10


class TickDesc(C.Structure):
    """navhip_tick_desc, include/navhip.h"""
    _fields_ = [("world", World), ("pos_xz_1", C.c_void_p), ("vel_xz_1", C.c_void_p), ("status", C.c_void_p),
                ("vdes_xz", C.c_void_p), ("vpref_xz", C.c_void_p), ("dev_reqs", C.c_void_p), ("n_reqs", C.c_int32),
                ("req_slot0", C.c_int32), ("field_pool_1", C.c_void_p), ("field_cus", C.c_int32),
                ("fields_stage", C.c_int32), ("dev_moves", C.c_void_p), ("n_moves", C.c_int32),
                ("n_move_ticks", C.c_int32), ("move_tick0", C.c_int32), ("bounds", C.c_void_p), ("stream", C.c_void_p),
                ("field_stream", C.c_void_p), ("comm_stream", C.c_void_p), ("flags", C.c_uint32)]


class TickInfo(C.Structure):
    """navhip_tick_info, include/navhip.h"""
    _fields_ = [("ticks", C.c_int64),
                ("host_enqueue_ms", C.c_double), ("stream", C.c_void_p), ("field_stream", C.c_void_p),
                ("comm_stream", C.c_void_p), ("fields_ms", C.c_double), ("fields_samples", C.c_int32), ("_pad", C.c_int32)]


_SIGS.update({
    "navhip_arrival_settle_resident": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "navhip_settled_count_resident": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "navhip_state_pass_resident": (C.c_int, [C.c_void_p, C.POINTER(StatePassIn), C.POINTER(StatePassOut)]),
    "navhip_tick_create": (C.c_int, [C.c_void_p, C.POINTER(TickDesc), C.POINTER(C.c_void_p)]),
    "navhip_tick_run": (C.c_int, [C.c_void_p, C.c_int]),
    "navhip_tick_compute": (C.c_int, [C.c_void_p]),
    "navhip_tick_advance": (C.c_int, [C.c_void_p]),
    "navhip_tick_sync": (C.c_int, [C.c_void_p]),
    "navhip_tick_get_info": (C.c_int, [C.c_void_p, C.POINTER(TickInfo)]),
    "navhip_tick_destroy": (None, [C.c_void_p]),
})


class Tick:
    """One navhip_tick: the per-tick loop of a device-resident world inside the library (the reference's
    navigation_tick_task, movement.c:4263).  `desc` is a filled TickDesc; `keep` whatever owns the device arrays."""

    def __init__(self, ctx, desc, keep=None):
        self.ctx, self._keep = ctx, (keep, desc)
        self._h = C.c_void_p()
        ctx._chk(lib().navhip_tick_create(ctx._h, C.byref(desc), C.byref(self._h)), "navhip_tick_create")
        self._run = lib().navhip_tick_run

    def run(self, n=1):
        rc = self._run(self._h, n)
        if rc != OK:
            self.ctx._chk(rc, "navhip_tick_run")

    def compute(self):
        self.ctx._chk(lib().navhip_tick_compute(self._h), "navhip_tick_compute")

    def advance(self):
        self.ctx._chk(lib().navhip_tick_advance(self._h), "navhip_tick_advance")

    def sync(self):
        self.ctx._chk(lib().navhip_tick_sync(self._h), "navhip_tick_sync")

    def info(self):
        out = TickInfo()
        self.ctx._chk(lib().navhip_tick_get_info(self._h, C.byref(out)), "navhip_tick_get_info")
        return out

    def close(self):
        if getattr(self, "_h", None):
            lib().navhip_tick_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
