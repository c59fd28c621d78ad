"""The parity tests of the GPU suite, run WITHOUT a GPU against libnavhip's own sources on a host emulator.

tests/hostsim/_navhip_emu.so is every translation unit of permafrost-engine_amd/csrc -- the API layer and all kernels,
unchanged but for two statements a host compiler cannot take -- compiled for the host on top of a lockstep emulator of
waves and workgroups (tests/hostsim/wave_emu.h: lanes are fibers, every ballot / shuffle / DPP move / barrier is a
rendezvous) and a stand-in for the HIP runtime (tests/hostsim/fakehip/: device memory is host memory, a launch runs its
grid block by block).  It exports the C ABI of include/navhip.h, so the very tests that pin the GPU build to the
reference build through that ABI run against it: the same kernel SOURCE is checked on a machine without a GPU, flow
fields, line of sight, blockers, the spatial index, the whole velocity step with its ClearPath kernels, the state
update, the exchange step of a multi-GPU tick.  Test infrastructure only: nothing loads this library unless NAVHIP_LIB names it, and libnavhip.so itself still
fails loudly without a device (test_abi_cpu.py).  Host arithmetic is IEEE where the device's native square root and
reciprocal square root are within an ulp: the kernels' own margins are what make both agree with the reference.

The selected tests are those that go through host buffers (no torch.cuda) and finish in seconds on the emulator; they run
in one pytest subprocess (the library is chosen when navhip.py is first imported)."""
import os
import subprocess
import sys

import pytest

from oracle import pfref
from tests import hostsim

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = [pytest.mark.skipif(not hostsim.group_available(), reason="no clang++ (ROCm LLVM) for the host build"),
              pytest.mark.skipif(not pfref.available(), reason="oracle/_ref (the reference build) is not present")]

SELECTION = [
    "tests/test_golden_gpu.py",
    "tests/test_fields_gpu.py",
    "tests/test_blockers_gpu.py",
    "tests/test_edge_gpu.py",
    "tests/test_agents_gpu.py",
    "tests/test_comm_gpu.py",          # the exchange step over the mailbox transport (device buffers = host buffers here)
    "tests/test_pool_gpu.py",          # the resident field pool and the asynchronous step
    "tests/test_state_gpu.py",         # the heading gate, the settled-neighbour count, the arrival overlay's settle rule
    "tests/test_tick_gpu.py",          # the whole tick behind one call (navhip_tick_run) against tick.py's schedule
]
# (agents: the tests that need torch.cuda, and the ones that take more than ~10 s each on the emulator)
DESELECT = ["test_prefetch_overlap_gives_identical_results", "test_shared_chunk_fields_give_identical_results",
            "test_pipelined_field_builds_give_identical_results", "test_lane_grouping_carried_between_ticks_is_only_a_hint",
            "test_device_los_lookup_matches_N_HasDestLOS", "test_slab_calls_share_one_output_buffer",
            "test_slab_lane_grouping_survives_membership_changes", "test_formation_arms_match_reference",
            "test_clearpath_retry_shortcut_matches_reference[12-32-32-4.5-team]",
            "test_clearpath_retry_shortcut_matches_reference[12-32-32-4.5-False]",
            "test_clearpath_retry_shortcut_matches_reference[11-24-24-3.0-team]",
            "test_clearpath_retry_shortcut_matches_reference[11-24-24-3.0-False]",
            "test_clearpath_retry_shortcut_matches_reference[15-16-16-2.6-team]",
            "test_velocity_step_matches_reference[True-1500-2-True]", "test_clearpath_matches_reference[2-32-32-9.5]"]


# device buffers ARE the test's host arrays in these (the mailbox transport between ranks of one process): they run
# without the strict pointer check below
NOT_STRICT = ["tests/test_comm_gpu.py", "tests/test_tick_gpu.py"]
# (the tick tests that take a minute and more each on the emulator)
DESELECT_TICK = ["test_c_tick_equals_the_python_schedule[crowded]", "test_drivers_can_take_turns",
                 "test_c_tick_equals_the_python_schedule[fields_in_front]",
                 "test_handovers_by_events_give_the_same_tick[crowded]", "test_tick_enters_a_jam_with_both_drivers",
                 "test_two_worlds_take_turns_on_the_same_streams",
                 # (a measurement of the hardware's queues: nothing for an emulator, and ten configs[2]-sized worlds)
                 "test_tick_time_does_not_depend_on_what_the_process_created_before"]


@pytest.mark.parametrize("strict", [True, False])
def test_gpu_parity_tests_pass_on_the_emulated_library(strict):
    """strict: with EMU_STRICT_POINTERS=1 the stand-in runtime aborts when a kernel is handed a pointer into a buffer
    the caller passed as HOST memory (instead of its staged device copy) -- on top of failing copies a GPU would
    reject: the two mistakes that work on an emulator, where host and device memory are one thing, and fault on a GPU."""
    lib = hostsim.build_navhip_emu()
    env = dict(os.environ, NAVHIP_LIB=lib)
    files = [f for f in SELECTION if (f not in NOT_STRICT) == strict]
    if strict:
        env["EMU_STRICT_POINTERS"] = "1"
    cmd = [sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider"] + files
    for name in DESELECT:
        cmd += ["--deselect", "tests/test_agents_gpu.py::" + name]
    cmd += ["--deselect", "tests/test_pool_gpu.py::test_step_joins_a_prefetch_issued_on_another_stream"]     # (torch.cuda streams)
    for mode in ("words", "events"):                                                                         # (torch.cuda streams)
        cmd += ["--deselect", "tests/test_pool_gpu.py::test_stage_waits_order_a_third_stream_behind_the_step[%s]" % mode]
    for name in DESELECT_TICK:
        cmd += ["--deselect", "tests/test_tick_gpu.py::" + name]
    try:
        import xdist  # noqa: F401
        cmd += ["-n", str(min(8, os.cpu_count() or 1))]
    except ImportError:
        pass
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = "\n".join(r.stdout.strip().splitlines()[-25:])
    assert r.returncode == 0, tail
    last = r.stdout.strip().splitlines()[-1]
    assert " passed" in last and "failed" not in last and "error" not in last, tail
    assert int(last.split(" passed")[0].split()[-1]) >= (70 if strict else 13), tail          # (the selection really ran)


def test_reference_binding_drives_the_emulated_library():
    """tests/test_binding_gpu.py and tests/test_state_binding_gpu.py -- the reference's own movement tick (both halves), field requests, field cache and blocker calls
    going through bindings/permafrost/*.c into the C ABI -- with the emulator build answering instead of libnavhip.so:
    the harness (oracle/_ref/libpfref.so) links the product library by name, so the emulator build is preloaded and its
    navhip_* symbols interpose."""
    lib = hostsim.build_navhip_emu()
    env = dict(os.environ, NAVHIP_LIB=lib, LD_PRELOAD=lib, EMU_STRICT_POINTERS="1")
    cmd = [sys.executable, "-m", "pytest", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "tests/test_binding_gpu.py",
           "tests/test_state_binding_gpu.py"]
    try:
        import xdist  # noqa: F401
        cmd += ["-n", str(min(8, os.cpu_count() or 1))]
    except ImportError:
        pass
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1500)
    tail = "\n".join(r.stdout.strip().splitlines()[-25:])
    assert r.returncode == 0, tail
    last = r.stdout.strip().splitlines()[-1]
    assert "17 passed" in last, tail


@pytest.mark.parametrize("extra", [["--pipeline-fields", "--shared", "--flow-velocities"], ["--straddle", "all"]])
def test_two_rank_gloo_tick_on_the_emulated_library(extra):
    """The multi-GPU tick with world_size 2 on CPU: two processes under torchrun, gloo, each with its own emulator build
    of the library (tick.NavTick sees host memory as its device then) -- field slices and agent slabs per rank, the slab
    exchange every tick, tiles travelling where flocks straddle ranks, the fields of the next tick built ahead, one
    shared world split over the ranks (bench.py --scaling strong) -- and after three ticks every rank's snapshot is
    bit-identical to ONE process that builds every field and steps every agent (scripts/check_multirank.py)."""
    lib = hostsim.build_navhip_emu()
    env = dict(os.environ, NAVHIP_LIB=lib, NAVHIP_DIST_BACKEND="gloo")
    port = 29540 + (len(extra) * 7 + sum(len(e) for e in extra)) % 40
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "scripts", "check_multirank.py"), "--small"] + extra
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    # (the two ranks print at the same moment: their lines may share one)
    assert r.returncode == 0 and r.stdout.count("IDENTICAL to solo") == 2 and "MISMATCH" not in r.stdout, r.stdout[-2000:]



def test_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus 2` -- the shape of the driver's N = 1 command, NOT wrapped in torchrun -- has to run two
    ranks by itself (torch.distributed.run, one rank per device; gloo and the emulator build here), split ONE world over
    them (BASELINE.json's metric: strong scaling), say how many ranks the communicator saw, run the weak-scaled job
    behind it, and print ONE line that fits the driver's record."""
    import json
    lib = hostsim.build_navhip_emu()
    env = dict(os.environ, NAVHIP_LIB=lib, NAVHIP_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--map", "4", "--fields", "2", "--agents", "300",
           "--steps", "2", "--warmup", "1", "--no-los"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and len(lines) == 1, (r.stdout + r.stderr)[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["steps"] == 2 and line["warmup"] == 1
    assert line["config"]["ranks"] == 2 and line["config"]["rccl_ranks"] == 2 and line["config"]["backend"] == "gloo"
    assert line["config"]["agents"] == 300 and line["weak_scaling"]["agents"] == 600        # one world split | a region per rank
    assert line["value"] > 0 and line["summary"]["ranks"] == 2
    assert len(lines[0]) < 6000
    assert all(len(v) <= 120 for v in line["config"].values() if isinstance(v, str))


def test_bench_refuses_more_ranks_than_devices():
    """Without the emulator: this container has no GPU, so `--gpus 2` must fail instead of faking ranks."""
    env = {k: v for k, v in os.environ.items() if k not in ("NAVHIP_LIB", "WORLD_SIZE", "RANK", "LOCAL_RANK")}
    import torch
    if torch.cuda.is_available() and torch.cuda.device_count() >= 2:
        pytest.skip("two devices present")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], cwd=ROOT,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 3 and "GPU(s) visible" in r.stderr and not r.stdout.strip()
