"""quick device-side timing probe (not the bench contract): python tools/gpu_probe.py [W H T]"""
import sys
import os
import time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import synth as S
from libagmv_amd import AgmvHip

W, H, T = (int(x) for x in sys.argv[1:4]) if len(sys.argv) >= 4 else (1920, 1080, 256)
mode512 = True
hip = AgmvHip(0)
f_np = [S.synth_frame(W, H, t) for t in range(2)]
p0, p1 = S.content_palettes(f_np)


def timed(fn, n=5, warm=1):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2], ts


ms, _ = timed(lambda: hip.set_palette(p0, p1, mode512), n=3)
print("set_palette (LUT+matrix build): %.3f ms" % ms)
frames = hip.synth_dev(W, H, 0, T)
torch.cuda.synchronize()
stride = hip.max_usize(W, H)
out = torch.empty((T, stride), dtype=torch.uint8, device="cuda")
sizes = torch.empty(T, dtype=torch.int32, device="cuda")
ms, ts = timed(lambda: hip.encode_dev(frames, T, W, H, out=out, sizes=sizes))
hip.check()
usz = sizes.cpu().numpy().astype(np.int64)
alg = 4 * W * H * T + usz.sum()
print("encode: %.3f ms  (%s) -> %.1f frames/s, %.1f GB/s algorithmic (%.1f%% of 8 TB/s); mean usize %.0f B (%.3f B/px)"
      % (ms, ["%.2f" % t for t in ts], T / ms * 1e3, alg / ms / 1e6, alg / ms / 1e6 / 80, usz.mean(), usz.mean() / (W * H)))
nblk = W * H // 16
offs = torch.empty((T, nblk), dtype=torch.int32, device="cuda")
nent = torch.empty(T, dtype=torch.int32, device="cuda")
ms_p, ts = timed(lambda: hip.parse_dev(out, sizes, T, W, H, offsets=offs, nentered=nent), n=5, warm=1)
print("parse: %.3f ms" % ms_p)
dec = torch.empty((T, H, W), dtype=torch.int32, device="cuda")
ms_d, ts = timed(lambda: hip.decode_dev(out, sizes, offs, nent, T, W, H, out=dec))
print("decode: %.3f ms (%s) -> %.1f frames/s, %.1f GB/s algorithmic (%.1f%% of 8 TB/s)"
      % (ms_d, ["%.2f" % t for t in ts], T / ms_d * 1e3, alg / ms_d / 1e6, alg / ms_d / 1e6 / 80))
ms_f, _ = timed(lambda: dec.fill_(7))
print("torch fill of the clip (4 B/px write only): %.3f ms -> %.1f GB/s" % (ms_f, 4 * W * H * T / ms_f / 1e6))
ms_r, _ = timed(lambda: frames.sum())
print("torch sum of the clip (4 B/px read only): %.3f ms -> %.1f GB/s" % (ms_r, 4 * W * H * T / ms_r / 1e6))
# plain copy of the same pixel volume for reference
ms_c, _ = timed(lambda: dec.copy_(frames))
print("torch copy of the clip (4 B/px read + 4 B/px write): %.3f ms -> %.1f GB/s" % (ms_c, 8 * W * H * T / ms_c / 1e6))
