"""A plain C99 host (examples/c_host_tick.c: include/navhip.h + the HIP runtime's C API, gcc -std=c99) drives the
navigation tick through navhip_tick_run -- "host code stays in C" -- and must leave the world where the Python driver of
the same library leaves it: positions, velocities, status bytes and the baked field pool, bit for bit."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _build(tmp_path):
    exe = str(tmp_path / "c_host_tick")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-O1", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(rocm, "include"), os.path.join(ROOT, "examples", "c_host_tick.c"),
           "-L" + os.path.join(ROOT, "permafrost-engine_amd"), "-lnavhip", "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
           "-Wl,-rpath," + os.path.join(ROOT, "permafrost-engine_amd"), "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
@pytest.mark.parametrize("ahead", [1, 0], ids=["fields_ahead", "fields_in_front"])
def test_c_host_drives_the_tick(navlib, tmp_path, ahead):
    if os.path.basename(os.environ.get("NAVHIP_LIB", "")) == "_navhip_emu.so":
        pytest.skip("the C host links the product library and the HIP runtime")
    from permafrost_engine_amd import synth, tick
    ticks = 7
    kw = dict(chunk_w=4, fields_per_rank=3, agents_per_rank=2000, pipeline_fields=bool(ahead))
    T = tick.NavTick(driver="python", **kw)
    n, k, nreq = T.N, T.K, len(T.host["reqs"])
    world = str(tmp_path / "world.bin")
    with open(world, "wb") as f:
        f.write(np.array([T.Wt, T.H, n, k, nreq, T.hz, ahead, 0], np.int32).tobytes())
        f.write(synth.to_chunks(T.grid).astype(np.uint8).tobytes())
        f.write(np.zeros((T.H, T.Wt, 64, 64), np.uint16).tobytes())
        f.write(synth.to_chunks(T.host["liid"]).astype(np.uint16).tobytes())
        f.write(T.host["reqs"].tobytes())
        f.write(np.ascontiguousarray(T.host["slot_tbl"], np.int32).tobytes())
        f.write(T.t["pos_xz"].cpu().numpy().astype(np.float32).tobytes())
        f.write(T.t["vel_xz"].cpu().numpy().astype(np.float32).tobytes())
        for name in ("radius", "max_speed", "speed"):
            f.write(np.ascontiguousarray(T.host[name], np.float32).tobytes())
        f.write(np.full(n, navlib.ENTITY_FLAG_MOVABLE, np.uint32).tobytes())
        f.write(np.zeros(n, np.uint8).tobytes())                       # state: STATE_MOVING
        f.write(np.zeros(n, np.uint8).tobytes())                       # has_dest_los
        f.write(np.ascontiguousarray(T.host["flock"], np.int32).tobytes())
        f.write(np.ascontiguousarray(T.host["targets"], np.float32).tobytes())
        f.write(np.ascontiguousarray(T.host["flock_offsets"], np.int32).tobytes())
        f.write(np.ascontiguousarray(T.host["flock_members"], np.int32).tobytes())
    for _ in range(ticks):
        T.step()
    T.sync()
    exp = (T.t["pos_xz"].cpu().numpy(), T.t["vel_xz"].cpu().numpy(), T.status.cpu().numpy(), T.pool.cpu().numpy())
    T.close()
    out = str(tmp_path / "out.bin")
    r = subprocess.run([_build(tmp_path), world, out, str(ticks)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert "%d ticks" % ticks in r.stderr
    raw = open(out, "rb").read()
    pos = np.frombuffer(raw, np.float32, 2 * n).reshape(n, 2)
    vel = np.frombuffer(raw, np.float32, 2 * n, offset=8 * n).reshape(n, 2)
    status = np.frombuffer(raw, np.uint8, n, offset=16 * n)
    pool = np.frombuffer(raw, np.uint8, nreq * 4096, offset=17 * n).reshape(nreq, 4096)
    assert np.array_equal(pos.view(np.uint32), exp[0].view(np.uint32)) and np.array_equal(vel.view(np.uint32), exp[1].view(np.uint32))
    assert np.array_equal(status, exp[2]) and np.array_equal(pool, exp[3])
    assert (status & 1).any() and (pool != 0).any()
