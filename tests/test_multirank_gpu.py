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


def test_four_ranks_over_rccl():
    import torch
    if torch.cuda.device_count() < 4:
        pytest.skip("needs four devices")
    _run(4, "nccl", ["all", "--pipeline-fields"])
