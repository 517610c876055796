"""encode-only timing probe for ablation variants"""
import sys, os
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import synth as S
from libagmv_amd import AgmvHip
W, H, T = 1920, 1080, 256
hip = AgmvHip(0)
p0, p1 = S.content_palettes([S.synth_frame(W, H, t) for t in range(2)])
hip.set_palette(p0, p1, True)
frames = hip.synth_dev(W, H, 0, T)
kind = sys.argv[1] if len(sys.argv) > 1 else "synth"
if kind == "noise":
    frames = torch.randint(0, 1 << 24, (T, H, W), dtype=torch.int32, device="cuda")
elif kind == "flat":     # one colour everywhere: every look-up of a wave hits the same table line
    frames = torch.full((T, H, W), 0x336699, dtype=torch.int32, device="cuda")
elif kind == "hgrad":    # colour depends on x only, slowly: a 64-pixel span stays inside one or two 4x4x4 cubes
    x = (torch.arange(W, device="cuda", dtype=torch.int32) // 16) & 0xff
    frames = (x | (x << 8) | (x << 16)).view(1, 1, W).expand(T, H, W).contiguous()
elif kind == "noise3":   # synthetic with every pixel's low 3 bits randomised -> NORMAL blocks, local LUT access
    frames = frames ^ torch.randint(0, 8, (T, H, W), dtype=torch.int32, device="cuda") ^ (torch.randint(0, 8, (T, H, W), dtype=torch.int32, device="cuda") << 8) ^ (torch.randint(0, 8, (T, H, W), dtype=torch.int32, device="cuda") << 16)
out = torch.empty((T, hip.max_usize(W, H)), dtype=torch.uint8, device="cuda")
sizes = torch.empty(T, dtype=torch.int32, device="cuda")
for _ in range(2): hip.encode_dev(frames, T, W, H, out=out, sizes=sizes)
torch.cuda.synchronize()
ts = []
for _ in range(7):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); hip.encode_dev(frames, T, W, H, out=out, sizes=sizes); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b))
hip.check()
usz = sizes.cpu().numpy().astype(np.int64)
print("%-40s %-6s encode %.3f ms (min %.3f)  mean usize %.0f (%.2f B/px)" % (os.path.basename(os.environ.get("AGMV_HIP_LIB", "baseline")), kind, sorted(ts)[3], min(ts), usz.mean(), usz.mean() / (W * H)))
