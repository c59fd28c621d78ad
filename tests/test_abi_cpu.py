"""CPU checks of the drop-in boundary: libnavhip.so builds, loads, exports every symbol that
include/navhip.h declares, its PODs have the layout the ctypes mirror assumes, its host-only
entry points answer, and -- without a GPU -- every compute entry point fails LOUDLY (no CPU
fallback inside the product)."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "navhip.h")


def _declared():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(navhip_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound(navlib):
    L = navlib.lib()
    names = _declared()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, "declared in include/navhip.h but not exported: %s" % missing
    unbound = [n for n in names if n not in navlib._SIGS]
    assert not unbound, "exported but not mirrored in navhip.py: %s" % unbound
    ghost = [n for n in navlib._SIGS if n not in names]
    assert not ghost, "navhip.py binds undeclared symbols: %s" % ghost
    # ... and nothing else leaves the library: kernels, launch helpers and C++ internals stay local
    # (csrc/navhip.map), every dynamic symbol it defines is declared in the header
    out = subprocess.run(["nm", "-D", "--defined-only", navlib.LIB_PATH], stdout=subprocess.PIPE, text=True,
                         check=True).stdout
    exported = sorted(line.split()[-1] for line in out.splitlines() if line.split()[1] in "TDBW")
    extra = [n for n in exported if n not in names]
    assert not extra, "exported but not declared in include/navhip.h: %s" % extra


def test_header_is_plain_c_and_layouts_match(navlib, tmp_path):
    """The header must compile as C99 (the reference's host language) and the PODs crossing the
    boundary must have the sizes/offsets the Python mirror uses."""
    prog = tmp_path / "layout.c"
    prog.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "navhip.h"
#define P(T, f) printf(#T "." #f " %zu\n", offsetof(T, f))
int main(void){
    printf("sizeof.navhip_field_req %zu\n", sizeof(navhip_field_req));
    printf("sizeof.navhip_world %zu\n", sizeof(navhip_world));
    printf("sizeof.navhip_step_out %zu\n", sizeof(navhip_step_out));
    printf("sizeof.navhip_circle %zu\n", sizeof(navhip_circle));
    printf("sizeof.navhip_gate_in %zu\n", sizeof(navhip_gate_in));
    printf("sizeof.navhip_arrival_zone %zu\n", sizeof(navhip_arrival_zone));
    printf("sizeof.navhip_settle_in %zu\n", sizeof(navhip_settle_in));
    printf("sizeof.navhip_settle_out %zu\n", sizeof(navhip_settle_out));
    printf("sizeof.navhip_state_aux_in %zu\n", sizeof(navhip_state_aux_in));
    printf("sizeof.navhip_state_pass_in %zu\n", sizeof(navhip_state_pass_in));
    printf("sizeof.navhip_state_pass_out %zu\n", sizeof(navhip_state_pass_out));
    P(navhip_state_pass_in, state); P(navhip_state_pass_in, aux);
    P(navhip_state_aux_in, ent_rot); P(navhip_state_aux_in, range_tiles); P(navhip_state_aux_in, n_range_rows);
    P(navhip_arrival_zone, radius); P(navhip_arrival_zone, key_end);
    P(navhip_settle_in, zones); P(navhip_settle_in, uid); P(navhip_settle_in, stuck);
    P(navhip_circle, radius); P(navhip_circle, faction_id); P(navhip_circle, delta);
    P(navhip_field_req, enemies); P(navhip_field_req, chunk_r); P(navhip_field_req, tile_r);
    P(navhip_field_req, port_r0); P(navhip_field_req, next_r0); P(navhip_field_req, next_chunk_r);
    P(navhip_field_req, port_iid); P(navhip_field_req, next_iid);
    P(navhip_world, pos_xz); P(navhip_world, vdes_xz); P(navhip_world, field_pool);
    P(navhip_world, map_pos_x); P(navhip_world, grid_xmin); P(navhip_world, work_begin);
    P(navhip_step_out, status);
    return 0;
}''')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I",
                           os.path.join(ROOT, "include"), str(prog), "-o", str(exe)])
    got = dict(l.split() for l in subprocess.check_output([str(exe)], text=True).splitlines())
    got = {k: int(v) for k, v in got.items()}
    dt = navlib.FIELD_REQ_DTYPE
    assert got["sizeof.navhip_field_req"] == dt.itemsize == 32
    for f in ("enemies", "chunk_r", "tile_r", "port_r0", "next_r0", "next_chunk_r", "port_iid",
              "next_iid"):
        assert got["navhip_field_req." + f] == dt.fields[f][1], f
    assert got["sizeof.navhip_world"] == C.sizeof(navlib.World)
    assert got["sizeof.navhip_step_out"] == C.sizeof(navlib.StepOut)
    assert got["sizeof.navhip_circle"] == navlib.CIRCLE_DTYPE.itemsize == 24
    for f in ("radius", "faction_id", "delta"):
        assert got["navhip_circle." + f] == navlib.CIRCLE_DTYPE.fields[f][1], f
    for f in ("pos_xz", "vdes_xz", "field_pool", "map_pos_x", "grid_xmin", "work_begin"):
        assert got["navhip_world." + f] == getattr(navlib.World, f).offset, f
    assert got["navhip_step_out.status"] == navlib.StepOut.status.offset
    for name, cls in (("gate_in", navlib.GateIn), ("arrival_zone", navlib.ArrivalZone), ("settle_in", navlib.SettleIn),
                      ("settle_out", navlib.SettleOut), ("state_aux_in", navlib.StateAuxIn),
                      ("state_pass_in", navlib.StatePassIn), ("state_pass_out", navlib.StatePassOut)):
        assert got["sizeof.navhip_" + name] == C.sizeof(cls), name
    assert got["sizeof.navhip_arrival_zone"] == 48
    for f in ("radius", "key_end"):
        assert got["navhip_arrival_zone." + f] == getattr(navlib.ArrivalZone, f).offset, f
    for f in ("zones", "uid", "stuck"):
        assert got["navhip_settle_in." + f] == getattr(navlib.SettleIn, f).offset, f
    for f in ("state", "aux"):
        assert got["navhip_state_pass_in." + f] == getattr(navlib.StatePassIn, f).offset, f
    for f in ("ent_rot", "range_tiles", "n_range_rows"):
        assert got["navhip_state_aux_in." + f] == getattr(navlib.StateAuxIn, f).offset, f


def _ff_id_expected(r):
    """N_FlowFieldID restated from field.c:1952-1975 (independent of the C code under test)."""
    if r["type"] == 0:
        return ((int(r["layer"]) << 60) | (0 << 56) | ((int(r["next_iid"]) & 0xf) << 48)
                | ((int(r["port_iid"]) & 0xf) << 40) | (int(r["port_r0"]) << 34)
                | (int(r["port_c0"]) << 28) | (int(r["port_r1"]) << 22) | (int(r["port_c1"]) << 16)
                | (int(r["chunk_r"]) << 8) | int(r["chunk_c"]))
    return ((int(r["layer"]) << 60) | (1 << 56) | (int(r["tile_r"]) << 24) | (int(r["tile_c"]) << 16)
            | (int(r["chunk_r"]) << 8) | int(r["chunk_c"]))


def test_flow_field_id_bit_layout(navlib):
    rng = np.random.RandomState(0)
    reqs = navlib.make_reqs(64)
    reqs["layer"] = rng.randint(0, 12, 64)
    reqs["type"] = rng.randint(0, 2, 64)
    for f in ("tile_r", "tile_c", "port_r0", "port_c0", "port_r1", "port_c1", "chunk_r", "chunk_c"):
        reqs[f] = rng.randint(0, 64, 64)
    reqs["port_iid"] = rng.randint(0, 40, 64)
    reqs["next_iid"] = rng.randint(0, 40, 64)
    ids = [navlib.N_FlowFieldID(reqs[i]) for i in range(64)]
    assert ids == [_ff_id_expected(reqs[i]) for i in range(64)]
    # distinct requests of one chunk never collide on the cache key
    assert len(set(ids)) == 64


def test_flow_field_id_matches_the_reference(navlib):
    """navhip_flow_field_id / navhip_region_field_id against the reference's own N_FlowFieldID
    (field.c:1952) through oracle/_ref: the planner's real request stream (real portals, island ids
    above 15) plus every region target kind."""
    from oracle import pfref
    from tests import cases
    if not (pfref.available() or os.path.isdir("/root/reference")):
        pytest.skip("oracle/_ref not built and /root/reference absent")
    grid, nav = cases.ref_nav_for(4, 3, seed=21)
    reqs_t = cases.tile_requests(grid, 16, seed=5)
    reqs_p, _b, _a = cases.planner_requests(nav, grid, pairs=10, seed=9)
    ref_reqs = np.concatenate([reqs_t, reqs_p])
    assert (ref_reqs["type"] == 0).sum() > 8
    for layer in (0, 3, 11):
        ref_reqs["layer"] = 0                      # (the portals are those of layer 0)
        want = [nav.flow_field_id(r) for r in ref_reqs]
        mine = cases.reqs_from_ref(navlib, ref_reqs)
        got = [navlib.N_FlowFieldID(mine[i]) for i in range(len(mine))]
        assert got == want
        # the layer field only enters through the top nibble (field.c:1956,1969)
        mine["layer"] = layer
        got_l = [navlib.N_FlowFieldID(mine[i]) for i in range(len(mine))]
        assert got_l == [(w & ~(0xf << 60)) | (layer << 60) for w in want]
    rng = np.random.RandomState(2)
    for _ in range(200):
        layer, cr, cc = int(rng.randint(12)), int(rng.randint(64)), int(rng.randint(64))
        fac, uid = int(rng.randint(15)), int(rng.randint(1 << 20))
        ar, ac, rad = int(rng.randint(64 * 64)), int(rng.randint(64 * 64)), int(rng.randint(1, 200))
        for kind, a, b, c in ((navlib.FFID_ENEMIES, fac, 0, 0), (navlib.FFID_ENTITY, uid, 0, 0),
                              (navlib.FFID_ZONE, ar, ac, rad)):
            assert navlib.N_RegionFieldID(kind, layer, cr, cc, a, b, c) \
                == pfref.RefNav.region_field_id(kind, layer, cr, cc, a, b, c), (kind, a, b, c)
    assert navlib.N_RegionFieldID(3, 0, 1, 1, 0) == 0          # TARGET_PORTALMASK has no cache key
    nav.close()


def test_invalid_arguments_are_rejected(navlib):
    L = navlib.lib()
    h = C.c_void_p()
    assert L.navhip_ctx_create(C.byref(h), 0, 4, 0) == -1          # NAVHIP_ERR_INVALID
    assert L.navhip_ctx_create(C.byref(h), 65, 4, 0) == -1         # 6-bit chunk ids, nav.c:841
    assert L.navhip_ctx_create(None, 4, 4, 0) == -1
    assert L.navhip_sync(None) == -1
    assert L.navhip_build_fields(None, None, 1, None, None) == -1
    assert L.navhip_device(None) == -1
    assert L.navhip_plane_dev(None, 0, 0) is None
    assert L.navhip_last_error(None) == b""
    # the state pass (csrc/state_kernels.hip): no context, no answer
    w = navlib.World()
    assert L.navhip_heading_gate(None, C.byref(w), C.byref(navlib.GateIn()), None, None, None) == -1
    assert L.navhip_heading_gate_dev(None, C.byref(w), C.byref(navlib.GateIn()), None, None, None, None) == -1
    assert L.navhip_settled_count(None, C.byref(w), 1, None, None) == -1
    assert L.navhip_arrival_settle(None, C.byref(w), C.byref(navlib.SettleIn()), C.byref(navlib.SettleOut())) == -1
    assert L.navhip_arrival_settle_dev(None, C.byref(w), C.byref(navlib.SettleIn()), C.byref(navlib.SettleOut()), None) == -1
    assert L.navhip_state_update_aux(None, C.byref(w), C.byref(navlib.StateAuxIn()), None, None, None) == -1
    assert L.navhip_state_update_aux_dev(None, C.byref(w), C.byref(navlib.StateAuxIn()), None, None, None, None) == -1
    assert L.navhip_state_pass(None, C.byref(w), C.byref(navlib.StatePassIn()), C.byref(navlib.StatePassOut())) == -1


def _gpu_visible():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_gpu_visible(), reason="a GPU is present: the loud-failure path cannot be seen")
def test_no_gpu_means_loud_failure_not_cpu_fallback(navlib):
    """The product path must refuse to run without the device: no CPU fallback, no oracle."""
    with pytest.raises(navlib.NavHipError):
        navlib.NavContext(2, 2, device=0)
    src = open(os.path.join(ROOT, "permafrost-engine_amd", "navhip.py")).read() \
        + open(os.path.join(ROOT, "permafrost-engine_amd", "tick.py")).read() \
        + open(os.path.join(ROOT, "permafrost-engine_amd", "dist.py")).read()
    assert "oracle" not in src, "product code must not import the oracle"


def test_product_sources_do_not_reference_the_oracle():
    pkg = os.path.join(ROOT, "permafrost-engine_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".c", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "import oracle" not in txt and "from oracle" not in txt, f
                assert "pfref" not in txt and "navoracle" not in txt, f


def _device_kernels(lib_path, tmp_path):
    """{kernel name: (allocated VGPRs, LDS bytes)} of the gfx950 code objects embedded in the built library (the
    HSA metadata notes, read with the ROCm LLVM tools; None when they are not there)."""
    import re
    import shutil
    import subprocess
    objdump, readelf = (os.path.join("/opt/rocm/lib/llvm/bin", t) for t in ("llvm-objdump", "llvm-readelf"))
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        return None
    work = os.path.join(str(tmp_path), "co")
    os.makedirs(work, exist_ok=True)
    shutil.copy(lib_path, work)                      # (--offloading extracts next to its input)
    subprocess.run([objdump, "--offloading", os.path.basename(lib_path)], cwd=work, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL, check=False)
    out = {}
    for f in os.listdir(work):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([readelf, "--notes", f], cwd=work, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                               text=True).stdout
        for block in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", block)
            vgpr = re.search(r"\.vgpr_count:\s+(\d+)", block)
            lds = re.search(r"\.group_segment_fixed_size:\s+(\d+)", block)
            if name and vgpr and lds:
                out[name.group(1)] = (int(vgpr.group(1)), int(lds.group(1)))
    return out or None


def test_persistent_clearpath_kernels_share_their_allocation_granules(tmp_path):
    """Hole inheritance: k_cp_heavy's persistent workgroups move into the register and LDS ranges that
    k_cp_rows' workgroups leave behind and keep them for the whole launch.  A k_cp_rows wave with fewer allocated
    registers, or a k_cp_rows workgroup with less LDS, leaves holes k_cp_heavy cannot use: a quarter of its waves
    for the whole launch (4.9 -> 6.1 ms per tick in the crowded world).  The built code objects must keep the two
    matched -- whatever the compiler's register allocation did this time."""
    from permafrost_engine_amd import build as nb
    lib = os.environ.get("NAVHIP_LIB") or os.path.join(os.path.dirname(nb.__file__), "libnavhip.so")
    if not os.path.exists(lib):
        pytest.skip("libnavhip.so not built")
    ks = _device_kernels(lib, tmp_path)
    if ks is None:
        pytest.skip("ROCm LLVM tools not available")
    rows = [v for k, v in ks.items() if "k_cp_rows" in k]
    heavy = [v for k, v in ks.items() if "k_cp_heavy" in k]
    assert len(rows) == 1 and len(heavy) == 1, sorted(ks)[:8]
    alloc = lambda v: (v + 7) // 8 * 8                # (the hardware allocates registers in granules of 8 per lane)
    assert alloc(rows[0][0]) == alloc(heavy[0][0]) == 128, (rows, heavy)
    assert rows[0][1] >= heavy[0][1], (rows, heavy)
    assert 4 * rows[0][1] <= 160 * 1024 and 4 * heavy[0][1] <= 160 * 1024      # four workgroups per CU, by LDS


def test_profile_stamps_cover_the_files_they_list(tmp_path):
    """profiles/traffic.json and sq_counters.json are quoted by bench.py only for the kernel code they were
    measured on: the stamp lists the sources it covers.  A translation unit added later leaves it valid; a
    change to a covered file (or its removal) does not; comments and whitespace do not count."""
    import json
    import shutil
    sys.path.insert(0, ROOT)
    import bench
    csrc = os.path.join(ROOT, "permafrost-engine_amd", "csrc")
    for name in ("traffic.json", "sq_counters.json"):
        stamp = json.load(open(os.path.join(ROOT, "profiles", name)))
        assert stamp["files"] == sorted(stamp["files"]) and "agent_kernels.hip" in stamp["files"]
        # the sha is over the units that can change the measured kernels: every header, the units that define or
        # name one of them, the host units; a unit with other kernels only (the state pass) is not among them
        cov = stamp["covers"]
        assert set(cov) <= set(stamp["files"]) and "agent_kernels.hip" in cov and "agent_math.h" in cov
        assert cov == bench.stamp_units(stamp["files"], bench.stamp_kernels(stamp)) or not bench.stamp_is_current(stamp)
        if not bench.stamp_is_current(stamp):
            # not an error of the tree: bench.py then prints "traffic": null, "traffic_stale": true and no counters;
            # the next PMC session (scripts/gpu_job.sh pmc + summarize_prof.py) restamps
            import warnings
            warnings.warn(name + " was measured on other kernel code than this tree's: bench.py will not quote it")
    d = str(tmp_path / "csrc")
    os.makedirs(d)
    for f in bench.csrc_files(csrc):
        shutil.copy(os.path.join(csrc, f), d)
    files, sha = bench.csrc_files(d), bench.csrc_sha(d)
    assert sha == bench.csrc_sha(csrc)
    open(os.path.join(d, "later_unit.hip"), "w").write("__global__ void k_later() {}\n")
    assert bench.csrc_sha(d) != sha and bench.csrc_sha(d, files) == sha
    with open(os.path.join(d, "agent_math.h"), "a") as f:
        f.write("// a comment\n\n")
    assert bench.csrc_sha(d, files) == sha
    with open(os.path.join(d, "agent_math.h"), "a") as f:
        f.write("#define NH_SOMETHING_ELSE 1\n")
    assert bench.csrc_sha(d, files) != sha
    os.remove(os.path.join(d, "agent_math.h"))
    assert bench.csrc_sha(d, files).startswith("missing:")
    # a unit whose kernels were not measured and that names no measured kernel is outside a stamp's cover
    cov = bench.stamp_units(bench.csrc_files(csrc), ["k_agent_mid", "k_field_bfs"], csrc)
    assert "state_kernels.hip" not in cov and "agent_kernels.hip" in cov and "field_kernels.hip" in cov
    assert "tick_api.hip" in cov and "navhip_api.hip" in cov and all(f in cov for f in bench.csrc_files(csrc) if f.endswith(".h"))
    assert bench.stamp_units(bench.csrc_files(csrc), [], csrc) == bench.csrc_files(csrc)


def test_c_host_example_compiles_as_c99(tmp_path):
    """examples/c_host_tick.c -- the plain C host of navhip_tick_run -- against include/navhip.h and the HIP runtime's C
    API, as C99 with warnings as errors (not -pedantic: the HIP headers use unnamed unions; it RUNS in tests/test_c_host_gpu.py)."""
    import shutil
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    if shutil.which("gcc") is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("no gcc / HIP headers")
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
                        "-I" + os.path.join(rocm, "include"), "-c", os.path.join(ROOT, "examples", "c_host_tick.c"),
                        "-o", str(tmp_path / "c_host_tick.o")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout


def test_makefile_builds_what_build_py_builds():
    """csrc/Makefile (libnavhip.so for a C build system, no Python) lists the sources and the code-generation flags of
    permafrost_engine_amd/build.py: the arithmetic flags are part of the parity contract."""
    import re
    from permafrost_engine_amd import build as nb
    mk = open(os.path.join(os.path.dirname(nb.__file__), "csrc", "Makefile")).read().replace("\\\n", " ")
    srcs = re.search(r"^SOURCES\s*=\s*(.*)$", mk, re.M).group(1).split()
    assert srcs == nb.SOURCES
    flags = re.search(r"^FLAGS\s*=\s*(.*)$", mk, re.M).group(1).split()
    want = [f for f in nb.FLAGS if not f.startswith("-I")]
    assert [f for f in flags if not f.startswith("-I")] == want
