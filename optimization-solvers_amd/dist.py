"""Process-group plumbing for row-sharded runs (one process per GPU).

The data plane is RCCL inside libqn_hip.so; this module only distributes the ncclUniqueId over an existing
torch.distributed group and offers a host-staged all-gather for tests (gloo)."""
import numpy as np

from .solver import Context


def sharded_context(device, group=None, host_exchange=False):
    """Create this rank's Context for the given torch.distributed group (default: WORLD).

    host_exchange=False: RCCL all-gather over xGMI (production).
    host_exchange=True : slices are staged through host memory and gathered with the group's own
                         all_gather (works with gloo; lets several ranks share one GPU in tests)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if world == 1:
        return Context(device)
    if host_exchange:
        return Context(device, rank=rank, world=world, host_allgather=gloo_allgather(group))
    ids = [None]
    if rank == 0:
        try:
            ids = [Context.unique_id()]
        except Exception as e:  # noqa: BLE001 -- tell the other ranks instead of leaving them in the broadcast
            ids = [e]
    dist.broadcast_object_list(ids, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    if isinstance(ids[0], Exception):
        raise ids[0]
    return Context(device, rank=rank, world=world, unique_id=ids[0])


def gloo_allgather(group=None):
    """Returns fn(send, recv): recv[r*count:(r+1)*count] = rank r's send, via torch.distributed.all_gather."""
    import torch
    import torch.distributed as dist

    def fn(send, recv):
        world = dist.get_world_size(group)
        t = torch.from_numpy(np.ascontiguousarray(send).copy())
        outs = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(outs, t, group=group)
        count = t.numel()
        for r in range(world):
            recv[r * count:(r + 1) * count] = outs[r].numpy()

    return fn


def row_range(n, rank, world, rows_per_rank):
    """Global rows [lo, hi) owned by `rank` (hi clipped to n)."""
    lo = min(n, rank * rows_per_rank)
    return lo, min(n, lo + rows_per_rank)
