"""Multi-GPU use of the decoder: codewords are independent, so a batch is split into contiguous row
blocks, one per rank (one process per GPU), every rank decodes its block with no data-path collective,
and only the decoded bytes (+ one verdict byte per codeword) are exchanged at the end
(RCCL all-gather over xGMI; ``gloo`` on CPU in the tests).  SURVEY section 8(e).
"""
import torch
import torch.distributed as dist


def shard_bounds(rows, rank, world):
    """contiguous block of rank `rank`: rows [lo, hi); the first rows % world ranks get one extra row"""
    base, extra = divmod(rows, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_result(out_bytes, ok):
    """[B, nb] uint8 + [B] uint8 -> [B, nb + 1] uint8 (one message per codeword)"""
    return torch.cat([out_bytes, ok.reshape(-1, 1).to(torch.uint8)], dim=1).contiguous()


def gather_decoded(packed_local, rows_total=None, group=None, equal_rows=False, out=None):
    """All-gather the per-rank packed results into the full [rows_total, nb + 1] matrix (every rank gets
    it).  Ranks may hold different row counts (ragged tail): shorter blocks are padded for the collective.
    equal_rows=True (every rank holds the same number of rows, e.g. the weak-scaling benchmark) skips the
    row-count exchange: exactly one collective, no host synchronisation; `out` may be a preallocated buffer."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if world == 1:
        return packed_local
    if equal_rows:
        if out is None:
            out = torch.empty((world * packed_local.shape[0], packed_local.shape[1]), dtype=packed_local.dtype,
                              device=packed_local.device)
        dist.all_gather_into_tensor(out, packed_local, group=group)
        return out
    n_local = torch.tensor([packed_local.shape[0]], dtype=torch.int64, device=packed_local.device)
    counts = [torch.zeros_like(n_local) for _ in range(world)]
    dist.all_gather(counts, n_local, group=group)
    counts = [int(c.item()) for c in counts]
    mx = max(counts)
    if all(c == mx for c in counts):
        full = torch.empty((world * mx, packed_local.shape[1]), dtype=packed_local.dtype, device=packed_local.device)
        dist.all_gather_into_tensor(full, packed_local, group=group)
        return full
    pad = torch.zeros((mx, packed_local.shape[1]), dtype=packed_local.dtype, device=packed_local.device)
    pad[: packed_local.shape[0]] = packed_local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    full = torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
    if rows_total is not None:
        assert full.shape[0] == rows_total
    return full
