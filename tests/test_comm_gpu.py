"""The exchange step of a multi-GPU tick (csrc/comm_api.hip) on ONE device.

RCCL refuses two ranks on one device, so on the one-GPU test box the collective itself cannot run with
world > 1.  Everything around it can: the exchange goes through a transport with two implementations -- RCCL
and a device-buffer MAILBOX (navhip_comm_init_mailbox) -- and packing, slab bounds, the ragged grouping and
unpacking are shared.  Here every rank of 2-, 3- and 4-rank jobs runs its side of the exchange against a
mailbox that holds what the other ranks would have sent (movement.c:3759-3762: the uid slabs of a ceil split)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _torch():
    import torch
    return torch


def _dev():
    """The device the buffers of these tests live on: the GPU -- or host memory when the library under test is the host
    emulator build (tests/test_emulated_cpu.py: "device" pointers are host pointers there)."""
    import os
    torch = _torch()
    emulated = os.path.basename(os.environ.get("NAVHIP_LIB", "")) == "_navhip_emu.so"
    return torch.device("cpu") if emulated else torch.device("cuda", 0)


def _sync():
    torch = _torch()
    if _dev().type == "cuda":
        torch.cuda.synchronize()


@pytest.mark.parametrize("bounds", [[0, 800, 1600], [0, 400, 1100, 1500], [0, 400, 800, 1200, 1600],
                                    [0, 0, 700, 1500], [0, 500, 500, 1500]])
def test_step_exchange_through_the_mailbox(navlib, bounds):
    torch = _torch()
    dev = _dev()
    world, n = len(bounds) - 1, bounds[-1]
    rng = np.random.RandomState(world * 7 + n)
    pos = rng.uniform(-500, 500, (n, 2)).astype(np.float32)
    vel = rng.normal(0, 1, (n, 2)).astype(np.float32)
    packed = np.concatenate([pos, vel], 1)
    for rank in range(world):
        b, e = bounds[rank], bounds[rank + 1]
        ctx = navlib.NavContext(1, 1)
        # the network: everybody's packed rows -- except this rank's own, which the call has to deposit
        mail = packed.copy()
        mail[b:e] = np.nan
        d_mail = torch.from_numpy(mail).to(dev)
        ctx.comm_init_mailbox(rank, world, d_mail)
        assert ctx.comm_world() == world and navlib.lib().navhip_comm_rank(ctx._h) == rank
        # this rank knows its own slab only
        p = np.full((n, 2), np.nan, np.float32)
        v = np.full((n, 2), np.nan, np.float32)
        p[b:e], v[b:e] = pos[b:e], vel[b:e]
        d_p, d_v = torch.from_numpy(p).to(dev), torch.from_numpy(v).to(dev)
        ctx.comm_allgather_step_dev(d_p, d_v, np.array(bounds, np.int32))
        _sync()
        assert np.array_equal(d_p.cpu().numpy(), pos), (rank, "positions")
        assert np.array_equal(d_v.cpu().numpy(), vel), (rank, "velocities")
        assert np.array_equal(d_mail.cpu().numpy(), packed), (rank, "deposit")
        ctx.comm_destroy()
        ctx.close()


@pytest.mark.parametrize("bounds", [[0, 6, 12], [0, 3, 4, 11]])
def test_tile_exchange_through_the_mailbox(navlib, bounds):
    """navhip_comm_allgather_rows_dev with 4 KB rows (baked flow tiles over the request stream)."""
    torch = _torch()
    dev = _dev()
    world, n = len(bounds) - 1, bounds[-1]
    tiles = np.random.RandomState(5).randint(0, 9, (n, 4096)).astype(np.uint8)
    for rank in range(world):
        b, e = bounds[rank], bounds[rank + 1]
        ctx = navlib.NavContext(1, 1)
        mail = tiles.copy()
        mail[b:e] = 0xEE
        d_mail = torch.from_numpy(mail).to(dev)
        ctx.comm_init_mailbox(rank, world, d_mail)
        mine = np.full((n, 4096), 0xDD, np.uint8)
        mine[b:e] = tiles[b:e]
        d_rows = torch.from_numpy(mine).to(dev)
        ctx.comm_allgather_rows_dev(d_rows, 4096, np.array(bounds, np.int32))
        _sync()
        assert np.array_equal(d_rows.cpu().numpy(), tiles), rank
        assert np.array_equal(d_mail.cpu().numpy(), tiles), rank
        ctx.close()                 # (destroys the communicator with the context)


def test_exchange_rejects_bad_bounds_and_short_mailboxes(navlib):
    torch = _torch()
    dev = _dev()
    ctx = navlib.NavContext(1, 1)
    n = 256
    d_p = torch.zeros((n, 2), dtype=torch.float32, device=dev)
    d_v = torch.zeros((n, 2), dtype=torch.float32, device=dev)
    L = navlib.lib()

    def call(bounds):
        b = np.array(bounds, np.int32)
        return L.navhip_comm_allgather_step_dev(ctx._h, C.c_void_p(d_p.data_ptr()), C.c_void_p(d_v.data_ptr()),
                                                b.ctypes.data_as(C.c_void_p), None)

    assert call([0, 128, 256]) == navlib.ERR_INVALID          # no communicator yet
    d_mail = torch.zeros((n, 4), dtype=torch.float32, device=dev)
    ctx.comm_init_mailbox(1, 2, d_mail)
    assert call([0, 128, 256]) == 0
    assert call([1, 128, 256]) == navlib.ERR_INVALID          # bounds[0] != 0
    assert call([0, 200, 100]) == navlib.ERR_INVALID          # decreasing
    short = torch.zeros((n // 2, 4), dtype=torch.float32, device=dev)
    ctx.comm_init_mailbox(1, 2, short)                        # (replaces the communicator)
    assert call([0, 128, 256]) == navlib.ERR_INVALID and "mailbox" in ctx.last_error()
    assert call([0, 100, 256]) == navlib.ERR_INVALID          # ragged path: the same check
    _sync()
    ctx.close()
