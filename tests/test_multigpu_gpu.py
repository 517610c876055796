"""N > 1 with the REAL hot path: ranks shard one clip by GOP range, every rank encodes / decodes its range on the GPU
through the C-ABI, the final gather (libagmv_amd.shard: DEVICE tensors, as with RCCL in bench.py; the gloo backend used
here stages them through the host inside shard.py, so the backend string is all that differs) puts the bitstreams together on
rank 0 -- which must hold exactly what ONE process makes of the whole clip, and what the oracle says.  Sharded decode uses
the decoder's own account of whether a range depends on the state before it (agmv_hip_decode_prior_dependent).
On a one-GPU box both ranks use device 0 (two processes on the card); on a multi-GPU node rank r takes device r."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, q, kind):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracles as O
    import synth as S
    from libagmv_amd import AgmvHip, shard
    dev = rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    W, H, T = 320, 240, 22
    frames = np.stack([S.synth_frame(W, H, t) for t in range(T)])
    p0, p1 = S.content_palettes(frames[:4])
    hip = AgmvHip(dev)
    hip.set_palette(p0, p1, True)
    lo, hi = shard.gop_ranges(T, world)[rank]
    d = torch.from_numpy(frames[lo:hi].view(np.int32)).cuda()
    out, sizes = hip.encode_dev(d, hi - lo, W, H, first_frame_count=lo)
    hip.check()
    packed = shard.pack_frames(out, sizes, hip=hip)            # agmv_hip_pack_frames_dev, as in bench.py
    assert torch.equal(packed, shard.pack_frames(out, sizes))
    got = shard.gather_bitstreams(dist, sizes, packed, dst=0)  # DEVICE tensors, as bench.py passes them to RCCL
    assert rank != 0 or all(s_.is_cuda and p_.is_cuda for s_, p_ in got)
    ok = True
    bits = None
    if rank == 0:
        bits = []
        for s_, p_ in got:
            bits += [x.cpu().numpy() for x in shard.split_packed(s_, p_)]
        enc = O.OracleEncoder(W, H, True, p0, p1)
        ok = len(bits) == T and all(len(bits[t]) == len(e) and (bits[t] == e).all() for t, e in ((t, enc.encode(frames[t])) for t in range(T)))
    # every rank needs the whole stream for the decode half: broadcast it from the root
    obj = [bits]
    dist.broadcast_object_list(obj, src=0)
    bits = obj[0]
    if kind == "escape":                                       # damaged frames.  Frame 12 opens rank 1's range: what it does not reach
        # keeps frame 11's pixels (rank 0's) -- a real dependency; 13 and 17 fall back on frames of their own range only
        bits = [b[:max(1, len(b) * (3 + t % 4) // 8)] if t in (7, 12, 13, 17) else b for t, b in enumerate(bits)]
    dec = O.OracleDecoder(W, H, True, p0, p1)
    ref, pads = [], []
    for b in bits:
        pix, padded, _, _ = dec.decode(b, want_tables=True)
        ref.append(pix)
        pads.append(padded[len(b):len(b) + 16])
    calls = []

    def decode_range(a, b, prev, prev_iframe):
        calls.append((a, b))
        stride = (max(len(x) for x in bits[a:b]) + 16 + 255) & ~255
        slab = np.zeros((b - a, stride), np.uint8)
        for i, f in enumerate(range(a, b)):
            slab[i, :len(bits[f])] = bits[f]
            slab[i, len(bits[f]):len(bits[f]) + 16] = pads[f]
        dbits = torch.from_numpy(slab).cuda()
        dbpos = torch.tensor([len(x) for x in bits[a:b]], dtype=torch.int32).cuda()
        offs, nent = hip.parse_dev(dbits, dbpos, b - a, W, H)
        pix = hip.decode_dev(dbits, dbpos, offs, nent, b - a, W, H, a, prev=prev, prev_iframe=prev_iframe)   # hand-off state arrives as device tensors
        torch.cuda.synchronize()
        return pix, hip.decode_depends_on_prior(W, H)

    a, b, pix = shard.decode_sharded(dist, decode_range, T)
    ok = ok and pix.is_cuda and all(bool((pix[f - a].cpu().numpy().view(np.uint32).ravel() == ref[f]).all()) for f in range(a, b))
    if kind == "clean":
        ok = ok and len(calls) == 1                            # nothing the encoder emits needs a hand-off
    elif rank > 0:
        ok = ok and len(calls) == 2
    q.put(bool(ok))
    dist.barrier()
    dist.destroy_process_group()
    hip.close()


@pytest.mark.parametrize("kind", ["clean", "escape"])
def test_two_ranks_encode_gather_decode_on_the_gpu(kind):
    import torch.multiprocessing as mp
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 33500 + (os.getpid() % 2000) + len(kind)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert all(q.get(timeout=5) is True for _ in range(world))
