"""Multi-GPU sharding of the AGMV hot path.

The path shards naturally by GOP (4 consecutive encoded frames, aligned to frame_count % 4 == 0): a GOP's
P-frames need only the entries of its own I-frame, so ranks never exchange anything while encoding or
decoding.  One process per GPU; the only collective is the FINAL GATHER of the per-frame bitstreams to
the rank that runs the host LZ stage and writes the container in frame order (RCCL over xGMI with the
"nccl" backend, gloo in the CPU tests).  Each peer sends over its own direct link to the root, so this is
a plain gather (sizes first, then the variable-length payloads), not a ring.

Decode shards the same way.  A GOP is self-contained for every stream the encoder can emit; only a stream that
raises the reference's `escape` (blocks the bitstream does not reach keep the previous frame's pixels, reference
src/agmv_decode.c:229-232) or carries a COPY block in an I-frame (reads the snapshot of the previous GOP, :277-285)
makes a range depend on the decoder state before it.  decode_sharded() decodes every range in parallel from the fresh
state, finds the dependent ranges from the parser's own outputs, and repairs exactly those with a serial hand-off of
(last frame, last I-frame snapshot) from the rank before -- a point-to-point send per dependent boundary, nothing
otherwise.
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


def pack_frames(out, sizes, hip=None):
    """[n, stride] uint8 slab + [n] sizes -> one contiguous uint8 tensor of the used bytes.  With the rank's AgmvHip context and
    device tensors this is agmv_hip_pack_frames_dev (two kernels); the torch form below (one slice copy per frame) is what the
    CPU tests of the protocol use"""
    if hip is not None and out.is_cuda:
        return hip.pack_frames_dev(out, sizes)[0]
    sz = [int(s) for s in sizes.tolist()]
    if not sz:
        return out.new_empty(0)
    return torch.cat([out[f, :sz[f]] for f in range(len(sz))])


def _stages(dist, t):
    """gloo moves host memory only: a device tensor goes through a host copy there (tests, and bench.py --backend gloo, which
    rehearse the multi-rank path on one card).  With "nccl" (= RCCL) device tensors travel as they are, over xGMI."""
    return t.is_cuda and dist.get_backend() == "gloo"


def _isend(dist, t, dst):
    return dist.isend(t.cpu() if _stages(dist, t) else t, dst=dst)


class _Recv:
    """irecv into `t` (through a host buffer where the backend needs one); wait() completes the copy"""

    def __init__(self, dist, t, src):
        self.t = t
        self.buf = torch.empty(t.shape, dtype=t.dtype) if _stages(dist, t) else t
        self.req = dist.irecv(self.buf, src=src)

    def wait(self):
        self.req.wait()
        if self.buf is not self.t:
            self.t.copy_(self.buf)


def _all_gather(dist, t):
    world = dist.get_world_size()
    if _stages(dist, t):
        h = t.cpu()
        outs = [torch.empty_like(h) for _ in range(world)]
        dist.all_gather(outs, h)
        return outs
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    return outs


def gather_bitstreams(dist, sizes, packed, dst=0):
    """final gather.  sizes: int32 [n_local] per-frame usize, packed: uint8 [sum(sizes)] -- device tensors on the GPU path (they
    stay on the device end to end with RCCL).  Returns on `dst` a list over ranks of (sizes, packed) in rank (= frame) order,
    on the caller's device; elsewhere None."""
    world, rank = dist.get_world_size(), dist.get_rank()
    meta = torch.tensor([sizes.numel(), packed.numel()], dtype=torch.int64, device=sizes.device)
    metas = _all_gather(dist, meta)
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
                reqs.append(_Recv(dist, s, r))
            if nb:
                reqs.append(_Recv(dist, p, r))
            res.append((s, p))
        for q in reqs:
            q.wait()
        return res
    reqs = []                                                  # both messages in flight at once; the root has posted every receive
    if sizes.numel():
        reqs.append(_isend(dist, sizes, dst))
    if packed.numel():
        reqs.append(_isend(dist, packed, dst))
    for q in reqs:
        q.wait()
    return None


def split_packed(sizes, packed):
    """inverse of pack_frames: list of per-frame uint8 tensors"""
    out, off = [], 0
    for s in sizes.tolist():
        out.append(packed[off:off + int(s)])
        off += int(s)
    return out


COPY_FLAG = 0x5E            # include/agmv_defines.h:51
NORMAL_FLAG = 0x2F          # :50
FILL_FLAG = 0x4E            # :49


def _block_end(row, start, bpos, mode512):
    """one block of the reference's loop (src/agmv_decode.c:234-319 / :335-396) entered at byte `start` of the frame `row`
    (numpy uint8): returns (flag, end) with the flag the block is decoded under -- the first flag-valued byte at or after the
    entry position (the resync of :236-243) -- and the position behind its last byte; flag None when the resync runs off the
    stream (the block is then not written)."""
    n = len(row)
    p = start
    while p < n and row[p] not in (FILL_FLAG, NORMAL_FLAG, COPY_FLAG):
        p += 1
        if p >= bpos:                                         # the byte read at p makes bitpos = p + 1 > bpos: escape
            return None, p
    if p >= n:
        return None, p
    flag = int(row[p])
    pos = p + 1
    if flag == FILL_FLAG:
        pos += 2 if (mode512 and pos < n and (row[pos] & 0x7f) == 127) else 1
    elif flag == NORMAL_FLAG:
        for _ in range(16):
            pos += 2 if (mode512 and pos < n and (row[pos] & 0x7f) == 127) else 1
    return flag, pos


def range_depends_on_prior_state(bits, bpos, offsets, nentered, nblk, mode512=True, first_is_iframe=True):
    """True when the pixels of a range may depend on img_data / iframe_data from before its first frame.  (Host-side form for
    callers that hold the parser's outputs; a GPU decode reports the same fact itself, AgmvHip.decode_depends_on_prior().)
    bits [n, stride] uint8, bpos [n], offsets [n, nblk] (byte position at which each block is ENTERED, i.e. before the
    flag resync), nentered [n].
      (a) a frame that raises `escape` leaves blocks >= nentered at the previous frame's values;
      (b) a COPY block in the range's first frame reads the snapshot taken before the range.  A block is decoded under the
          first flag-valued byte at or after its entry position, so that byte is what is classified;
      (c) a block that runs past bpos is not (FILL, :268-271) or not completely (NORMAL, per-pixel check :310-314 /
          :386-392) stored.  Inside a frame that ends the frame (case a); as the LAST block it has to be looked at itself.
    Everything else a frame writes is a function of its own bitstream and of frames inside the range."""
    n = int(nentered.numel())
    if n == 0:
        return False
    if bool((nentered < nblk).any()):
        return True
    if not first_is_iframe:
        return True                                           # the range continues a GOP of the caller's
    import numpy as np
    bp = bpos.cpu().numpy().astype("int64")
    off = offsets.cpu().numpy().astype("int64")
    rows = bits.cpu().numpy()                                  # one transfer for the whole range
    first = rows[0]
    # (b), and (c) for the resync of every block of the first frame.  Vectorised for the common case -- the entry byte IS a
    # flag: COPY decides at once, FILL / NORMAL need no look here; only entries that have to slide take the scalar walk
    o0 = off[0, :nblk]
    inside = o0 < len(first)
    eb = np.where(inside, first[np.minimum(o0, len(first) - 1)], 0)
    if bool(((eb == COPY_FLAG) & inside).any()):
        return True
    slide = ~inside | ~np.isin(eb, (FILL_FLAG, NORMAL_FLAG, COPY_FLAG))
    for k in np.nonzero(slide)[0]:
        flag, _ = _block_end(first, int(o0[k]), int(bp[0]), mode512)
        if flag is None or flag == COPY_FLAG:
            return True
    for f in range(n):                                        # (c): the last block of every frame
        flag, end = _block_end(rows[f], int(off[f, nblk - 1]), int(bp[f]), mode512)
        if flag is None or (flag != COPY_FLAG and end > int(bp[f])):
            return True
    return False


def decode_sharded(dist, decode_range, n_frames, first_frame_count=0, prev=None, prev_iframe=None):
    """Bit-exact GOP-sharded decode.  decode_range(lo, hi, prev, prev_iframe) decodes frames [lo, hi) of the clip from
    the given prior state (None = fresh decoder) and returns (pixels [hi-lo, H, W] int32, depends: bool) where
    `depends` is range_depends_on_prior_state() of that range.  Returns this rank's (lo, hi, pixels).
    Hand-off: after the parallel pass the `depends` flags are all-gathered; for each dependent rank r (in rank order)
    the nearest non-empty rank before it sends its final frame and its I-frame snapshot, and r decodes again."""
    world, rank = dist.get_world_size(), dist.get_rank()
    ranges = gop_ranges(n_frames, world, first_frame_count)
    lo, hi = ranges[rank]
    pix, dep = decode_range(lo, hi, prev if rank == 0 else None, prev_iframe if rank == 0 else None)
    dev = pix.device
    flag = torch.tensor([1 if (dep and rank > 0) else 0], dtype=torch.int64, device=dev)
    need = [bool(int(f.item())) for f in _all_gather(dist, flag)]

    def last_state(p, r_lo, r_hi, in_prev, in_iframe):
        """(img_data, iframe_data) after the last frame of a decoded non-empty range"""
        last = p[-1]
        snap = in_iframe
        for f in range(r_hi - 1, r_lo - 1, -1):               # D4: snapshot at frame_count % 4 == 0 (:401-405)
            if ((first_frame_count + f) & 3) == 0:
                snap = p[f - r_lo]
                break
        return last, snap

    # the state entering this rank, once known to be needed
    in_prev, in_iframe = (prev, prev_iframe) if rank == 0 else (None, None)
    for r in range(1, world):
        if not need[r] or ranges[r][1] <= ranges[r][0]:
            continue
        src = max((q for q in range(r) if ranges[q][1] > ranges[q][0]), default=None)
        if src is None:
            continue                                          # nothing before it: the caller's state is rank 0's
        if rank == src:
            last, snap = last_state(pix, lo, hi, in_prev, in_iframe)
            if snap is None:
                snap = torch.zeros_like(last)                  # fresh decoder: zeroed iframe (:532-560)
            _isend(dist, last.contiguous(), r).wait()
            _isend(dist, snap.contiguous(), r).wait()
        elif rank == r:
            shape = pix.shape[1:]
            in_prev = torch.empty(shape, dtype=pix.dtype, device=dev)
            in_iframe = torch.empty(shape, dtype=pix.dtype, device=dev)
            _Recv(dist, in_prev, src).wait()
            _Recv(dist, in_iframe, src).wait()
            pix, _ = decode_range(lo, hi, in_prev, in_iframe)
    return lo, hi, pix
