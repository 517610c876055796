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
