"""N > 1 path on CPU: two gloo ranks shard a clip by GOP ranges, each produces its frames' bitstreams
(with the oracle encoder standing in for the GPU -- test infrastructure only), rank 0 gathers them with
libagmv_amd.shard and must hold exactly the single-process result, in frame order."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def test_gop_ranges_cover_and_align():
    from libagmv_amd.shard import gop_ranges
    for n in (1, 3, 4, 5, 156, 765, 1024, 1027):
        for world in (1, 2, 3, 4, 8):
            for fc in (0, 1, 2, 3, 8):
                r = gop_ranges(n, world, fc)
                assert r[0][0] == 0 and r[-1][1] == n
                for (a, b), (c, d) in zip(r, r[1:]):
                    assert b == c and a <= b
                for lo, hi in r[1:]:
                    if 0 < lo < n and hi > lo:
                        assert (fc + lo) % 4 == 0, (n, world, fc, r)
                if n >= 4 * world:
                    sizes = [hi - lo for lo, hi in r]
                    assert max(sizes) - min(sizes) <= 4 + 3


def _worker(rank, world, port, q):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracles as O
    import synth as S
    from libagmv_amd import shard
    W, H, T = 64, 48, 22
    frames = [S.synth_frame(W, H, t) for t in range(T)]
    p0, p1 = S.content_palettes(frames[:4])
    lo, hi = shard.gop_ranges(T, world)[rank]
    enc = O.OracleEncoder(W, H, True, p0, p1, first_frame_count=lo)      # shard starts on a GOP boundary
    bits = [enc.encode(frames[f]) for f in range(lo, hi)]
    stride = max(len(b) for b in bits) + 7
    slab = torch.zeros((len(bits), stride), dtype=torch.uint8)
    for i, b in enumerate(bits):
        slab[i, :len(b)] = torch.from_numpy(b)
    sizes = torch.tensor([len(b) for b in bits], dtype=torch.int32)
    got = shard.gather_bitstreams(dist, sizes, shard.pack_frames(slab, sizes), dst=0)
    if rank == 0:
        all_bits = []
        for s, p in got:
            all_bits += [x.numpy() for x in shard.split_packed(s, p)]
        ref = O.OracleEncoder(W, H, True, p0, p1)
        ok = len(all_bits) == T
        for f in range(T):
            e = ref.encode(frames[f])
            ok = ok and len(e) == len(all_bits[f]) and bool((e == all_bits[f]).all())
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_encode_gather_equals_single_process(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True


# ---------------------------------------------------------------------------------------------
# decode: ranges that depend on the decoder state before them (escape / COPY in an I-frame) are repaired by the
# serial hand-off of libagmv_amd.shard.decode_sharded; clean streams must not trigger it
# ---------------------------------------------------------------------------------------------
def _streams(kind):
    """the clip's decompressed bitstreams as the reference's persistent buffer presents them (16 bytes past bpos
    included), their bpos, and the serial oracle decode"""
    import ctypes as C
    import oracles as O
    import synth as S
    W, H, T = 64, 48, 22
    frames = [S.synth_frame(W, H, t) for t in range(T)]
    p0, p1 = S.content_palettes(frames[:4])
    enc = O.OracleEncoder(W, H, True, p0, p1)
    bits = [enc.encode(f) for f in frames]
    nblk = W * H // 16
    if kind == "escape":                                       # truncated frames around every shard boundary
        for f in (3, 7, 8, 11, 12, 15, 16, 19):
            bits[f] = bits[f][:max(1, len(bits[f]) * (3 + f % 4) // 8)]
    elif kind == "copy":                                       # I-frames made of COPY blocks: read the previous snapshot
        for f in (8, 12, 16):
            bits[f] = np.full(nblk, 0x5E, np.uint8)
    dec = O.OracleDecoder(W, H, True, p0, p1)
    padded, ref = [], []
    for b in bits:
        pix, pad, _, _ = dec.decode(b, want_tables=True)
        padded.append(pad)
        ref.append(pix)
    return W, H, T, p0, p1, padded, [len(b) for b in bits], ref


def _dec_worker(rank, world, port, q, kind):
    import ctypes as C
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracles as O
    from libagmv_amd import shard
    W, H, T, p0, p1, padded, bpos, ref = _streams(kind)
    nblk = W * H // 16
    calls = []

    def decode_range(lo, hi, prev, prev_iframe):               # the oracle stands in for parse_dev + decode_dev
        calls.append((lo, hi))
        d = O.OracleDecoder(W, H, True, p0, p1)
        d.s.frame_count = lo
        if prev is not None:
            np.ctypeslib.as_array(d.s.img, (W * H,))[:] = prev.numpy().view(np.uint32).ravel()
        if prev_iframe is not None:
            np.ctypeslib.as_array(d.s.iframe, (W * H,))[:] = prev_iframe.numpy().view(np.uint32).ravel()
        pix, offs, nent = [], [], []
        for f in range(lo, hi):
            d.L.orc_decoder_set_bitstream(d.p, padded[f], len(padded[f]))
            d.s.bpos = bpos[f]
            o = np.zeros(nblk, np.uint32)
            n = C.c_uint32(0)
            d.L.orc_decoder_parse(d.p, o.ctypes.data_as(C.c_void_p), C.byref(n))
            pix.append(np.ctypeslib.as_array(d.s.img, (W * H,)).copy())
            offs.append(o)
            nent.append(n.value)
        stride = max(len(p) for p in padded)
        slab = torch.zeros((hi - lo, stride), dtype=torch.uint8)
        for i, f in enumerate(range(lo, hi)):
            slab[i, :len(padded[f])] = torch.from_numpy(padded[f])
        dep = shard.range_depends_on_prior_state(
            slab, torch.tensor(bpos[lo:hi], dtype=torch.int32), torch.from_numpy(np.stack(offs).astype(np.int32)),
            torch.tensor(nent, dtype=torch.int32), nblk, True, first_is_iframe=(lo % 4 == 0))
        return torch.from_numpy(np.stack(pix).view(np.int32)).reshape(hi - lo, H, W), dep

    lo, hi, pix = shard.decode_sharded(dist, decode_range, T)
    ok = all(bool((pix[f - lo].numpy().view(np.uint32).ravel() == ref[f]).all()) for f in range(lo, hi))
    if kind == "clean":
        ok = ok and len(calls) == 1                            # no hand-off for anything the encoder emits
    elif rank > 0:
        ok = ok and len(calls) == 2
    q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("kind", ["clean", "escape", "copy"])
def test_sharded_decode_equals_serial_decode(world, kind):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world * 7 + len(kind)
    procs = [ctx.Process(target=_dec_worker, args=(r, world, port, q, kind)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert all(q.get(timeout=5) is True for _ in range(world))
