"""Work partitioning across the GPUs of one node, one process per GPU.

Every ray and every iso value is independent and the tree is read-only, so each rank holds a full
replica of the node array and renders its own share; nothing is exchanged on the data path.  The
only collective is the optional gather of finished framebuffers to rank 0 (RCCL over xGMI under
torch.distributed's "nccl" backend, gloo on CPU).

  iso sharding   the 0..255 sweep of the reference driver (src/main.zig:111): rank r renders a
                 contiguous block of iso values
  row sharding   one large frame (src/raytrace.zig:40 row loop): rank r renders a contiguous block
                 of image rows
  band sharding  the rows in bands of 32 dealt out round-robin: the rows that never meet the volume lie
                 at the top and the bottom of the image, so contiguous blocks leave the outer ranks idle;
                 every rank keeps all iso values of its rays, and with them the sharing of traversals
                 between iso values
"""

BAND_ROWS = 32  # one super-tile of the pre-pass: 4 tiles of 8 rows


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


def band_shard(rank, world, height, band_rows=BAND_ROWS):
    """Bands of `band_rows` image rows dealt out round-robin: rank r gets bands r, r + world, ...
    Returns dict(row_begin, band_rows, band_pitch, n_bands, n_rows) -- the arguments of vr_render_device_bands and
    the number of rows the rank renders (the last band of the image may be short); n_bands = 0: nothing to do."""
    total_bands = (int(height) + band_rows - 1) // band_rows
    n_bands = (total_bands - rank + world - 1) // world if rank < total_bands else 0
    row_begin = rank * band_rows
    n_rows = 0
    for j in range(n_bands):
        first = row_begin + j * world * band_rows
        n_rows += min(band_rows, height - first)
    return {"row_begin": row_begin, "band_rows": band_rows, "band_pitch": world * band_rows, "n_bands": n_bands,
            "n_rows": n_rows}


def band_rows_of(rank, world, height, band_rows=BAND_ROWS):
    """Image rows of a rank's bands, in the order they are packed in its output."""
    sh = band_shard(rank, world, height, band_rows)
    rows = []
    for j in range(sh["n_bands"]):
        first = sh["row_begin"] + j * sh["band_pitch"]
        rows.extend(range(first, min(first + band_rows, height)))
    return rows


def gather_rows(local, row_lists, height, dst=0, group=None):
    """Gathers per-rank frames [iso, rows_r, W, 3] whose rows are row_lists[r] of the image (contiguous blocks or
    bands) to rank `dst` and puts every row in its place: returns [iso, height, W, 3] on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    counts = [len(r) for r in row_lists]
    rows_first = local.transpose(0, 1).contiguous()  # [rows_r, iso, W, 3]: the gathered dimension leads
    got = gather_frames(rows_first, counts, dst=dst, group=group) if world > 1 else rows_first
    if rank != dst:
        return None
    order = torch.tensor([r for rows in row_lists for r in rows], dtype=torch.long, device=got.device)
    full = torch.empty((height,) + tuple(got.shape[1:]), dtype=got.dtype, device=got.device)
    full[order] = got
    return full.transpose(0, 1).contiguous()


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
