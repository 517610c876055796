# k_decode / parser timing on T x 1080p: the two-call form (offsets[]) against agmv_hip_decode_bitstreams_dev (entry bitmaps)
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import synth as S
from libagmv_amd import AgmvHip
W, H, T = int(os.environ.get("W", "1920")), int(os.environ.get("H", "1080")), int(os.environ.get("T", "256"))
hip = AgmvHip(0, lib=os.environ.get("PROBE_LIB"))
p0, p1 = S.content_palettes([S.synth_frame(W, H, t) for t in range(2)])
hip.set_palette(p0, p1, True); hip.enable_timing(True)
frames = hip.synth_dev(W, H, 0, T)
kind = sys.argv[1] if len(sys.argv) > 1 else "synth"
if kind == "noise3":
    frames = frames ^ torch.randint(0, 8, (T, H, W), dtype=torch.int32, device="cuda") ^ (torch.randint(0, 8, (T, H, W), dtype=torch.int32, device="cuda") << 8) ^ (torch.randint(0, 8, (T, H, W), dtype=torch.int32, device="cuda") << 16)
out, sizes = hip.encode_dev(frames, T, W, H)
nblk = W*H//16
offs, nent = hip.parse_dev(out, sizes, T, W, H)
dec = torch.empty((T, H, W), dtype=torch.int32, device="cuda")
dec2 = torch.empty((T, H, W), dtype=torch.int32, device="cuda")
ts = []; tp = []; ts2 = []; tp2 = []
LEG = bool(os.environ.get("LEGACY_ONLY"))
for _ in range(7):
    hip.parse_dev(out, sizes, T, W, H, offsets=offs, nentered=nent); tp.append(hip.last_kernel_ms(1))
    hip.decode_dev(out, sizes, offs, nent, T, W, H, out=dec); ts.append(hip.last_kernel_ms(2))
    if LEG: tp2.append(0.0); ts2.append(0.0); continue
    hip.decode_bitstreams_dev(out, sizes, T, W, H, out=dec2); tp2.append(hip.last_kernel_ms(1)); ts2.append(hip.last_kernel_ms(2))
torch.cuda.synchronize()
assert LEG or torch.equal(dec, dec2)
m = lambda a: sorted(a)[len(a) // 2]
print("%s  offsets[]: parse %.3f + decode %.3f = %.3f ms | bitmaps: parse %.3f + decode %.3f = %.3f ms | mean usize %.0f" %
      (kind, m(tp), m(ts), m(tp) + m(ts), m(tp2), m(ts2), m(tp2) + m(ts2), float(sizes.float().mean())))
