import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import synth as S
from libagmv_amd import AgmvHip
W, H, T = 1920, 1080, int(os.environ.get("T", "256"))
hip = AgmvHip(0)
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
ts = []; tp = []
for _ in range(6):
    hip.parse_dev(out, sizes, T, W, H, offsets=offs, nentered=nent); tp.append(hip.last_kernel_ms(1))
    hip.decode_dev(out, sizes, offs, nent, T, W, H, out=dec); ts.append(hip.last_kernel_ms(2))
print("%s decode %.3f ms parse %.3f ms  mean usize %.0f" % (kind, sorted(ts)[3], sorted(tp)[3], float(sizes.float().mean())))
