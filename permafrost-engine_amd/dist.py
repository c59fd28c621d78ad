"""One-process-per-GPU sharding of the navigation tick.

The path shards two ways, both along boundaries the reference already has:
  * chunk-field requests are independent given the replicated cost/blockers planes
    (the <=256 independent field_task jobs of nav.c:2049) -> contiguous request slices per rank;
  * the velocity step is fork-joined over contiguous uid slabs (move_submit_cpu_work,
    movement.c:3756-3762) -> one slab per rank, every rank holding the full position snapshot.
Exchange steps per tick, all-gathers (no reductions):
  * the slab results (new position + velocity, 16 B per agent), so every rank starts the next
    tick with the full snapshot -- always;
  * the baked 4 KB flow tiles -- only when agents sample fields another rank built (tick.NavTick
    keeps flocks rank aligned, so by default no tile travels; `tile_exchange="all"` gathers all).
`backend="nccl"` is RCCL over xGMI on the MI355X node; the CPU test-suite runs the same code on
`gloo` with world_size 2.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), \
        int(os.environ.get("LOCAL_RANK", "0"))


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (no-op for a single process)."""
    rank, world, local = env_world()
    if torch.cuda.is_available():
        # (more ranks than GPUs only happens in the single-GPU plumbing check, with gloo)
        local = local % torch.cuda.device_count()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("NAVHIP_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def slab(n_items, rank, world):
    """Contiguous [begin, end) share of n_items for `rank` (ceil split like movement.c:3759)."""
    per = -(-n_items // world)
    b = min(rank * per, n_items)
    return b, min(b + per, n_items)


def request_slices(dest_of_req, n_dests, world):
    """Per-rank [begin, end) slices of a destination-major chunk-field request stream: rank r
    builds every chunk field of the destinations slab(n_dests, r, world)."""
    import numpy as np
    dest_of_req = np.asarray(dest_of_req)
    out = []
    for r in range(world):
        d0, d1 = slab(n_dests, r, world)
        sel = np.flatnonzero((dest_of_req >= d0) & (dest_of_req < d1))
        out.append((int(sel[0]), int(sel[-1]) + 1) if len(sel) else (0, 0))
    return out



def travelling_destinations(flock, agents_per_rank, fields_per_rank, n_dests):
    """tile_exchange="auto": destination d is built by rank d // fields_per_rank; its baked tiles have to
    travel when some agent of its flock sits in another rank's uid slab (uid // agents_per_rank).
    Returns a bool array [n_dests]."""
    import numpy as np
    flock = np.asarray(flock)
    travels = np.zeros(n_dests, bool)
    ok = flock >= 0
    stepped_by = np.arange(len(flock)) // agents_per_rank
    travels[np.unique(flock[ok & (stepped_by != flock // fields_per_rank)])] = True
    return travels


def travel_first(dest_of_req, travels):
    """Stable order of one rank's requests with those of the travelling destinations first -- one
    contiguous run per rank to exchange -- and the length of that run."""
    import numpy as np
    first = np.asarray(travels)[np.asarray(dest_of_req)]
    return np.argsort(~first, kind="stable"), int(first.sum())


def exchange_rows(full, bounds, rank, world):
    """The tick's exchange step: rank r has just produced rows bounds[r] = [begin, end) of `full`
    (baked 4 KB flow tiles, or a slab of agent results); afterwards every rank holds every row.
    Equal contiguous shares go through ONE all-gather (RCCL ring over xGMI on the GPUs); ragged
    shares fall back to one broadcast per rank."""
    if world == 1:
        return
    sizes = [e - b for b, e in bounds]
    equal = len(set(sizes)) == 1 and all(bounds[r][0] == r * sizes[0] for r in range(world)) \
        and sizes[0] * world == full.shape[0]
    if equal:
        all_gather_rows(full, rank, world, sizes[0])
    else:
        for r, (b, e) in enumerate(bounds):
            if e > b:
                dist.broadcast(full[b:e], src=r)


def all_gather_rows(full, rank, world, rows_per_rank):
    """In-place all-gather of equally sized row slabs of `full` ([world*rows_per_rank, ...]):
    rank r contributes full[r*rows_per_rank:(r+1)*rows_per_rank] and receives the others."""
    if world == 1:
        return
    if full.is_cuda and dist.get_backend() == "gloo":
        # gloo has no GPU all-gather: stage through the host (plumbing checks on a single GPU only)
        host = full.cpu()
        dist.all_gather_into_tensor(host, host[rank * rows_per_rank:(rank + 1) * rows_per_rank].contiguous())
        full.copy_(host)
        return
    mine = full[rank * rows_per_rank:(rank + 1) * rows_per_rank]
    dist.all_gather_into_tensor(full, mine.contiguous())


def comm_ranks():
    """Ranks of the communicator the exchange steps run on (backend nccl = RCCL); 1 for a single process."""
    return dist.get_world_size() if dist.is_initialized() else 1


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value, device):
    if not dist.is_initialized():
        return value
    if dist.get_backend() == "gloo":
        device = "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
