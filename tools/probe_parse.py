import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import synth as S
from libagmv_amd import AgmvHip
W, H, T = 1920, 1080, 256
hip = AgmvHip(0)
p0, p1 = S.content_palettes([S.synth_frame(W, H, t) for t in range(2)])
hip.set_palette(p0, p1, True); hip.enable_timing(True)
frames = hip.synth_dev(W, H, 0, T)
out, sizes = hip.encode_dev(frames, T, W, H)
nblk = W*H//16
offs = torch.empty((T, nblk), dtype=torch.int32, device="cuda"); nent = torch.empty(T, dtype=torch.int32, device="cuda")
for gx in sys.argv[1:]:
    os.environ["AGMV_PARSE_GX"] = gx
    ts = []
    for _ in range(5):
        hip.parse_dev(out, sizes, T, W, H, offsets=offs, nentered=nent); ts.append(hip.last_kernel_ms(1))
    print("gx", gx, "parse ms", sorted(ts)[2])
