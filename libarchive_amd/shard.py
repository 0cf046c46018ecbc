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


def frame_weights(index, stream_bytes: int):
    """Per-frame cost C + U of ONE indexed .lz4 stream: C = the frame's bytes in the
    stream (magic word up to the next frame's magic word, trailing bytes go to the last frame),
    U = its decoded size as the block table bounds it (sum of the blocks' dst_cap)."""
    import numpy as np
    frames, blocks = index.frames, index.blocks
    nf = len(frames)
    if nf == 0:
        return np.zeros(0, dtype=np.uint64), np.zeros(1, dtype=np.uint64)
    starts = frames["desc_off"].astype(np.uint64) - np.uint64(4)     # the magic word sits in front of FLG
    bounds = np.concatenate([starts, np.array([stream_bytes], dtype=np.uint64)])
    c = bounds[1:] - bounds[:-1]
    capsum = np.concatenate([[0], np.cumsum(blocks["dst_cap"].astype(np.uint64))])
    fb = frames["first_block"].astype(np.int64)
    u = capsum[fb + frames["n_blocks"].astype(np.int64)] - capsum[fb]
    return (c + u).astype(np.uint64), bounds


def split_stream(index, stream_bytes: int, world: int):
    """Cut ONE stream into `world` contiguous frame ranges balanced by C + U, cuts on frame
    boundaries only (a frame's content checksum then stays on one GPU; the reference resets its
    state per block / per frame: lz4.c:557-561, :615-668).  Returns [(frame_lo, frame_hi)] * world."""
    import numpy as np
    w, _ = frame_weights(index, stream_bytes)
    nf = len(w)
    cum = np.concatenate([[0], np.cumsum(w.astype(np.float64))])
    total = cum[-1]
    cuts = [0]
    for r in range(1, world):
        # first frame boundary at or past r/world of the total weight, never before the previous cut
        k = int(np.searchsorted(cum, total * r / world, side="left"))
        cuts.append(min(max(k, cuts[-1]), nf))
    cuts.append(nf)
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def slice_index(index, stream_bytes: int, frame_lo: int, frame_hi: int):
    """The part of a stream's index a rank owns: (rebased Lz4Index, byte_lo, byte_hi).  The rank
    uploads stream[byte_lo:byte_hi] only; offsets in the returned tables are relative to byte_lo."""
    import numpy as np
    from . import _native as N
    _, bounds = frame_weights(index, stream_bytes)
    lo, hi = int(bounds[frame_lo]), int(bounds[frame_hi])
    frames = index.frames[frame_lo:frame_hi].copy()
    if len(frames) == 0:
        return N.Lz4Index(index.blocks[:0].copy(), frames, N.LA_END_EOF, 0, 0), lo, lo
    b_lo = int(frames["first_block"][0])
    b_hi = int(frames["first_block"][-1]) + int(frames["n_blocks"][-1])
    blocks = index.blocks[b_lo:b_hi].copy()
    blocks["src_off"] -= np.uint64(lo)
    frames["desc_off"] -= np.uint64(lo)
    frames["first_block"] -= np.uint32(b_lo)
    max_out = int(blocks["dst_cap"].astype(np.uint64).sum())
    end_kind = index.end_kind if frame_hi == len(index.frames) else N.LA_END_EOF
    return N.Lz4Index(blocks, frames, end_kind, hi - lo, max_out), lo, hi


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


def gather_ranges_into(dist, device, local_bytes, root: int = 0):
    """The same gather without the final concatenation and without padding: the root posts one receive per peer
    straight into that peer's slot of ONE preallocated buffer (slots of max-size bytes, what a 100+ GiB gather can
    afford), every other rank sends exactly its own bytes -- no rank pads or copies its range (round 2 padded every
    range to the largest one and went through dist.gather).  Point-to-point on purpose: xGMI is point-to-point
    (seven links per GPU), the root's seven receives run on seven links.  Returns (buffer, slot_bytes, sizes) on the
    root, (None, slot_bytes, sizes) elsewhere."""
    import torch
    world = dist.get_world_size()
    rank = dist.get_rank()
    n = torch.tensor([local_bytes.numel()], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s[0]) for s in sizes]
    mx = max(sizes) if sizes else 0
    if rank == root:
        buf = torch.empty(world * mx, dtype=torch.uint8, device=device)
        reqs = [dist.irecv(buf[r * mx:r * mx + sizes[r]], src=r) for r in range(world) if r != root and sizes[r]]
        buf[root * mx:root * mx + sizes[root]] = local_bytes      # the root's own range: one device copy
        for q in reqs:
            q.wait()
        return buf, mx, sizes
    if local_bytes.numel():
        dist.send(local_bytes.contiguous(), dst=root)
    return None, mx, sizes


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
