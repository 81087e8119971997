"""Work partitioning across the GPUs of one node, one process per GPU.

Every ray and every iso value is independent and the tree is read-only, so each rank holds a full
replica of the node array and renders its own share; nothing is exchanged on the data path.  The
only collective is the optional gather of finished framebuffers to rank 0 (RCCL over xGMI under
torch.distributed's "nccl" backend, gloo on CPU).

  iso sharding   the 0..255 sweep of the reference driver (src/main.zig:111): rank r renders a
                 contiguous block of iso values
  row sharding   one large frame (src/raytrace.zig:40 row loop): rank r renders a contiguous block
                 of image rows
"""


def split_range(begin, count, rank, world):
    """Contiguous near-equal split of [begin, begin+count): returns (my_begin, my_count)."""
    base, rem = divmod(int(count), int(world))
    my_count = base + (1 if rank < rem else 0)
    my_begin = int(begin) + rank * base + min(rank, rem)
    return my_begin, my_count


def iso_shard(rank, world, iso_begin=0, iso_count=256):
    return split_range(iso_begin, iso_count, rank, world)


def row_shard(rank, world, height):
    b, c = split_range(0, height, rank, world)
    return b, b + c


def gather_frames(local, counts, dst=0, group=None):
    """Gathers per-rank tensors whose leading dimension differs (counts[r] entries on rank r) to rank
    `dst` and concatenates them along dim 0.  Returns the full tensor on `dst`, None elsewhere.
    Uses point-to-point gather (no ring): rank dst's ingress links are the bound."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    if world == 1:
        return local
    maxc = max(counts)
    pad_shape = (maxc,) + tuple(local.shape[1:])
    buf = local
    if local.shape[0] != maxc:
        buf = torch.zeros(pad_shape, dtype=local.dtype, device=local.device)
        buf[: local.shape[0]] = local
    if rank == dst:
        parts = [torch.empty(pad_shape, dtype=local.dtype, device=local.device) for _ in range(world)]
        dist.gather(buf, gather_list=parts, dst=dst, group=group)
        return torch.cat([parts[r][: counts[r]] for r in range(world)], dim=0)
    dist.gather(buf, gather_list=None, dst=dst, group=group)
    return None
