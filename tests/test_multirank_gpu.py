"""The multi-rank tick under torchrun on the GPU box: every rank's snapshot after K ticks must be
bit-identical to one process that builds every field and steps every agent (scripts/check_multirank.py).
With two or more devices the ranks run one per GPU over RCCL (backend "nccl"); on a single-GPU box they
share the device and gloo stages the exchanges through the host -- the same sharding plan, the same
exchange calls, the same stream ordering."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, backend, extra):
    env = dict(os.environ, NAVHIP_DIST_BACKEND=backend, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "scripts", "check_multirank.py")] + extra
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    out = r.stdout + r.stderr
    assert r.returncode == 0, out[-3000:]
    assert out.count("IDENTICAL to solo") == world, out[-3000:]
    return out


@pytest.mark.parametrize("extra", [[], ["all"], ["--pipeline-fields"], ["all", "--pipeline-fields"]])
def test_two_ranks_equal_one_process(extra):
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    out = _run(2, backend, extra)
    assert ("backend=%s" % backend) in out


@pytest.mark.parametrize("extra", [["--shared"], ["--shared", "--pipeline-fields", "--flow-velocities"]])
def test_two_ranks_split_one_world(extra):
    """Strong scaling (bench.py --scaling strong; BASELINE.json: "1024^2 map, 100k agents, 1/2/4/8 GPUs"): ONE
    map, its destinations split by rank, its agents split into uid slabs (movement.c:3759-3762) -- agents of
    different slabs are neighbours everywhere on the map.  Every rank's snapshot == the one-process result."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    _run(2, backend, extra)


@pytest.mark.parametrize("extra", [["--straddle"], ["--straddle", "--pipeline-fields"]])
def test_straddling_flocks_exchange_only_their_tiles(extra):
    """tile_exchange="auto" with flocks that straddle the ranks: a quarter of every rank's agents sample
    fields the other rank builds; only those destinations' tiles travel (one contiguous run per rank), and
    the result is still the one-process result."""
    import torch
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    out = _run(2, backend, extra)
    import re
    m = re.search(r"tile_exchange=auto \(tiles travelling (\d+) of (\d+)\)", out)
    assert m and 0 < int(m.group(1)) <= int(m.group(2)) // 2 + 8, out[-2000:]    # half of the flocks travel


def test_four_ranks_over_rccl():
    import torch
    if torch.cuda.device_count() < 4:
        pytest.skip("needs four devices")
    _run(4, "nccl", ["all", "--pipeline-fields"])


def test_c_abi_exchange_single_rank(navlib):
    """navhip_comm_*: librccl called directly from the library (what a C host uses).  One rank on the one
    GPU of this box: the communicator comes up, both all-gather forms run and leave the rows as they are."""
    import numpy as np
    import torch
    ctx = navlib.NavContext(2, 2)
    try:
        uid = navlib.comm_unique_id()
        assert len(uid) == navlib.COMM_ID_BYTES and any(uid)
        assert ctx.comm_world() == 0
        ctx.comm_init(0, 1, uid)
        assert ctx.comm_world() == 1
        n = 1000
        pos = torch.rand((n, 2), device="cuda")
        vel = torch.rand((n, 2), device="cuda")
        p0, v0 = pos.clone(), vel.clone()
        s = torch.cuda.Stream()
        ctx.comm_allgather_step_dev(pos, vel, [0, n], stream=s.cuda_stream)
        tiles = torch.randint(0, 9, (16, 4096), dtype=torch.uint8, device="cuda")
        t0 = tiles.clone()
        ctx.comm_allgather_rows_dev(tiles, 4096, [0, 16], stream=s.cuda_stream)
        s.synchronize()
        assert torch.equal(pos, p0) and torch.equal(vel, v0) and torch.equal(tiles, t0)
        with pytest.raises(navlib.NavHipError):
            ctx.comm_allgather_step_dev(pos, vel, [5, n])        # bounds[0] must be 0
        ctx.comm_destroy()
        assert ctx.comm_world() == 0
    finally:
        ctx.close()


@pytest.mark.parametrize("extra", [["--exchange-navhip"], ["--exchange-navhip", "--pipeline-fields"]])
def test_two_ranks_over_the_c_abi_exchange(extra):
    """The same two-rank job with the slab all-gather through navhip_comm_allgather_step_dev (RCCL from C)
    instead of torch.distributed.  RCCL refuses two ranks on one device: needs two GPUs."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two devices (RCCL does not run two ranks on one GPU)")
    out = _run(2, "nccl", extra)
    assert "exchange=navhip" in out
