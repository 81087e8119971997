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


class FrameGather:
    """The one exchange step of a sharded render: every rank's part of the framebuffers goes to rank `dst`, point to point
    (grouped isend / irecv: ncclSend / ncclRecv under the "nccl" backend, no ring -- rank dst's ingress links are the bound).
    Built once per job; calling it gathers one step's parts and allocates nothing:

      frames   parts[r] = counts[r] leading entries (iso shards: whole frames): received straight into their place in the
               result, no staging, no copy besides rank dst's own part
      rows     parts[r] = [iso, len(row_lists[r]), W, 3] with the image rows row_lists[r] (contiguous blocks or bands): a part
               is contiguous on the wire but strided in the result, so it is received into a staging buffer of its own and
               put in place by one indexed copy -- one pass over the framebuffer in all

    The result lives in the object ([sum(counts), ...] or [iso, height, W, 3]) and is overwritten by the next call."""

    def __init__(self, like, counts=None, row_lists=None, height=None, dst=0, group=None):
        import torch
        import torch.distributed as dist
        self.dist, self.torch = dist, torch
        self.dst, self.group = dst, group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rows = row_lists is not None
        tail = tuple(like.shape[1:]) if not self.rows else None
        self.out, self.staging, self.index = None, {}, {}
        if self.rows:
            self.counts = [len(r) for r in row_lists]
            if self.rank == dst:
                n_iso = like.shape[0]
                self.out = torch.empty((n_iso, height) + tuple(like.shape[2:]), dtype=like.dtype, device=like.device)
                for r in range(self.world):
                    if self.counts[r] == 0:
                        continue
                    self.index[r] = torch.tensor(row_lists[r], dtype=torch.long, device=like.device)
                    if r != dst:
                        self.staging[r] = torch.empty((n_iso, self.counts[r]) + tuple(like.shape[2:]), dtype=like.dtype, device=like.device)
        else:
            self.counts = [int(c) for c in counts]
            if self.rank == dst:
                self.out = torch.empty((sum(self.counts),) + tail, dtype=like.dtype, device=like.device)

    def __call__(self, local):
        dist = self.dist
        if self.world == 1:
            return local
        if self.rank != self.dst:
            if self.counts[self.rank] > 0:
                for req in dist.batch_isend_irecv([dist.P2POp(dist.isend, local.contiguous(), self.dst, self.group)]):
                    req.wait()
            return None
        ops, off = [], 0
        for r in range(self.world):
            n = self.counts[r]
            if n and r != self.dst:
                into = self.staging[r] if self.rows else self.out[off:off + n]
                ops.append(dist.P2POp(dist.irecv, into, r, self.group))
            off += n
        reqs = dist.batch_isend_irecv(ops) if ops else []
        # rank dst's own part goes in place while the others are on the wire
        if self.counts[self.dst]:
            if self.rows:
                self.out.index_copy_(1, self.index[self.dst], local)
            else:
                own = sum(self.counts[:self.dst])
                self.out[own:own + self.counts[self.dst]].copy_(local)
        for req in reqs:
            req.wait()
        if self.rows:
            for r, part in self.staging.items():
                self.out.index_copy_(1, self.index[r], part)
        return self.out


def gather_rows(local, row_lists, height, dst=0, group=None):
    """Gathers per-rank frames [iso, rows_r, W, 3] whose rows are row_lists[r] of the image (contiguous blocks or
    bands) to rank `dst` and puts every row in its place: returns [iso, height, W, 3] on `dst`, None elsewhere.
    One-shot form of FrameGather (which a job that gathers every step builds once)."""
    return FrameGather(local, row_lists=row_lists, height=height, dst=dst, group=group)(local)


def gather_frames(local, counts, dst=0, group=None):
    """Gathers per-rank tensors whose leading dimension differs (counts[r] entries on rank r) to rank `dst`, one behind
    the other along dim 0.  Returns the full tensor on `dst`, None elsewhere.  One-shot form of FrameGather."""
    return FrameGather(local, counts=counts, dst=dst, group=group)(local)
