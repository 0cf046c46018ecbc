"""Multi-GPU sharding of independent units: frames / members are split into
contiguous per-rank ranges, every rank decodes its own range into its own HBM, and the
only exchange is a few bytes of per-rank summary (torch.distributed: RCCL over xGMI with
the "nccl" backend on the GPU node, gloo in the CPU tests).  No decoded byte moves unless
a caller explicitly gathers ranges."""


def shard_range(n_units: int, world: int, rank: int):
    """Contiguous, balanced split: returns (first, count) of this rank's units."""
    base, extra = divmod(n_units, world)
    first = rank * base + min(rank, extra)
    return first, base + (1 if rank < extra else 0)


def exchange_summaries(dist, device, elapsed_s: float, decoded_bytes: int, compressed_bytes: int, ok: bool):
    """Returns (max elapsed over ranks, total decoded, total compressed, all ok).

    One all_reduce(MAX) for the clock and one all_gather of a 3-word record per rank."""
    import torch
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return elapsed_s, float(decoded_bytes), float(compressed_bytes), bool(ok)
    t = torch.tensor([elapsed_s], dtype=torch.float64, device=device)
    rec = torch.tensor([float(decoded_bytes), float(compressed_bytes), 1.0 if ok else 0.0],
                       dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    allr = [torch.zeros_like(rec) for _ in range(world)]
    dist.all_gather(allr, rec)
    return (float(t[0]), sum(float(r[0]) for r in allr), sum(float(r[1]) for r in allr),
            all(float(r[2]) == 1.0 for r in allr))


def gather_ranges(dist, device, local_bytes, root: int = 0):
    """Explicit gather of decoded ranges to one rank (north star: "only when a single stream
    is split").  local_bytes: 1-D uint8 tensor on `device`.  Returns the concatenation on
    `root` (None elsewhere).  Sizes are exchanged first; payloads use gather-to-root."""
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    n = torch.tensor([local_bytes.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s[0]) for s in sizes]
    mx = max(sizes) if sizes else 0
    pad = torch.zeros(mx, dtype=torch.uint8, device=device)
    pad[:local_bytes.numel()] = local_bytes
    bufs = [torch.zeros(mx, dtype=torch.uint8, device=device) for _ in range(world)] if rank == root else None
    dist.gather(pad, bufs, dst=root)
    if rank != root:
        return None
    return torch.cat([b[:s] for b, s in zip(bufs, sizes)])
