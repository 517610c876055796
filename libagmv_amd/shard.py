"""Multi-GPU sharding of the AGMV hot path.

The path shards naturally by GOP (4 consecutive encoded frames, aligned to frame_count % 4 == 0): a GOP's
P-frames need only the entries of its own I-frame, so ranks never exchange anything while encoding or
decoding.  One process per GPU; the only collective is the FINAL GATHER of the per-frame bitstreams to
the rank that runs the host LZ stage and writes the container in frame order (RCCL over xGMI with the
"nccl" backend, gloo in the CPU tests).  Each peer sends over its own direct link to the root, so this is
a plain gather (sizes first, then the variable-length payloads), not a ring.
"""
import torch


def gop_ranges(n_frames, world, first_frame_count=0):
    """contiguous [lo, hi) frame ranges, one per rank, cut at GOP boundaries and balanced in GOPs.
    Frame f has frame_count = first_frame_count + f; a range may only start where that is 0 mod 4
    (except the very first, which continues the caller's GOP)."""
    phase = first_frame_count & 3
    n_gops = (n_frames + phase + 3) // 4
    out = []
    for r in range(world):
        g_lo = (n_gops * r) // world
        g_hi = (n_gops * (r + 1)) // world
        lo = max(0, 4 * g_lo - phase)
        hi = min(n_frames, 4 * g_hi - phase)
        out.append((lo, max(lo, hi)))
    return out


def pack_frames(out, sizes):
    """[n, stride] uint8 slab + [n] sizes -> one contiguous uint8 tensor of the used bytes"""
    sz = [int(s) for s in sizes.tolist()]
    if not sz:
        return out.new_empty(0)
    return torch.cat([out[f, :sz[f]] for f in range(len(sz))])


def gather_bitstreams(dist, sizes, packed, dst=0):
    """final gather.  sizes: int32 [n_local] per-frame usize, packed: uint8 [sum(sizes)].
    Returns on `dst` a list over ranks of (sizes, packed) in rank (= frame) order, elsewhere None."""
    world, rank = dist.get_world_size(), dist.get_rank()
    meta = torch.tensor([sizes.numel(), packed.numel()], dtype=torch.int64, device=sizes.device)
    metas = [torch.empty_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta)
    if rank == dst:
        res = []
        reqs = []
        for r in range(world):
            if r == dst:
                res.append((sizes, packed))
                continue
            n, nb = int(metas[r][0]), int(metas[r][1])
            s = torch.empty(n, dtype=sizes.dtype, device=sizes.device)
            p = torch.empty(nb, dtype=torch.uint8, device=packed.device)
            if n:
                reqs.append(dist.irecv(s, src=r))
            if nb:
                reqs.append(dist.irecv(p, src=r))
            res.append((s, p))
        for q in reqs:
            q.wait()
        return res
    if sizes.numel():
        dist.send(sizes, dst=dst)
    if packed.numel():
        dist.send(packed, dst=dst)
    return None


def split_packed(sizes, packed):
    """inverse of pack_frames: list of per-frame uint8 tensors"""
    out, off = [], 0
    for s in sizes.tolist():
        out.append(packed[off:off + int(s)])
        off += int(s)
    return out
