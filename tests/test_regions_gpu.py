"""Multi-rank world layout (tick.NavTick regions) on ONE GPU: two emulated ranks exchanging their
slab results by hand must stay bit-identical to one process that builds every field and steps every
agent (solo) -- covers the rank-local field pools, the slab filter of the spatial hash and agents
that see neighbours across the region border."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_emulated_ranks_match_solo():
    import torch
    from permafrost_engine_amd import tick
    kw = dict(chunk_w=2, fields_per_rank=3, agents_per_rank=3000, world=2)
    T0 = tick.NavTick(rank=0, **kw)
    T1 = tick.NavTick(rank=1, **kw)
    S = tick.NavTick(rank=0, solo=True, **kw)
    assert T0.tile_exchange == "none" and S.tile_exchange == "all"
    assert T0.n_req_local + T1.n_req_local == S.n_req_local
    (a0, a1), (b0, b1) = T0.agent_bounds
    # the two regions really interact: some agents sit within a query radius of the border
    x = S.t["pos_xz"][:, 0].cpu().numpy()
    assert (np.abs(x) < 30.0).sum() > 10
    for _ in range(4):
        T0.compute(); T1.compute()
        T0.sync(); T1.sync()
        for dst, src, (lo, hi) in ((T0, T1, (b0, b1)), (T1, T0, (a0, a1))):
            dst.new_pos[lo:hi] = src.new_pos[lo:hi]
            dst.new_vel[lo:hi] = src.new_vel[lo:hi]
        torch.cuda.synchronize()               # (the copies ran on the default stream)
        T0.advance(); T1.advance()
        S.step(); S.sync()
        for T in (T0, T1):
            assert torch.equal(T.t["pos_xz"], S.t["pos_xz"])
            assert torch.equal(T.t["vel_xz"], S.t["vel_xz"])
    moved = (S.t["vel_xz"].abs().sum(1) > 0).float().mean().item()
    assert moved > 0.5
    for T in (T0, T1, S):
        T.close()
