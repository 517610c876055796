# parser timing (fast path + robust fallback) by content kind, with the number of frames that fell back
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import synth as S
from libagmv_amd import AgmvHip
W, H = 1920, 1080
T = int(os.environ.get("T", "256"))
hip = AgmvHip(0)
p0, p1 = S.content_palettes([S.synth_frame(W, H, t) for t in range(2)])
hip.set_palette(p0, p1, True); hip.enable_timing(True)
for kind in sys.argv[1:]:
    frames = hip.synth_dev(W, H, 0, T)
    if kind == "noise3":
        for sh in (0, 8, 16):
            frames ^= torch.randint(0, 8, (T, H, W), dtype=torch.int32, device="cuda") << sh
    elif kind == "flat":
        frames[:] = 0x336699
    out, sizes = hip.encode_dev(frames, T, W, H)
    del frames
    os.environ.pop("AGMV_HIP_PARSE", None)
    offs, nent = hip.parse_dev(out, sizes, T, W, H)
    ts = []
    for _ in range(5):
        hip.parse_dev(out, sizes, T, W, H, offsets=offs, nentered=nent); ts.append(hip.last_kernel_ms(1))
    fb = hip.parse_fallback_frames()
    os.environ["AGMV_HIP_PARSE"] = "robust"
    offs2, nent2 = hip.parse_dev(out, sizes, T, W, H)
    tr = []
    for _ in range(5):
        hip.parse_dev(out, sizes, T, W, H, offsets=offs2, nentered=nent2); tr.append(hip.last_kernel_ms(1))
    same = bool(torch.equal(nent, nent2))
    nb = W * H // 16
    idx = torch.arange(nb, device="cuda")[None, :] < nent[:, None]
    same = same and bool(torch.equal(offs[idx], offs2[idx]))
    print("%-7s fast %.3f ms  robust %.3f ms  fallback frames %d / %d  same=%s  mean usize %.0f" % (kind, sorted(ts)[2], sorted(tr)[2], fb, T, same, float(sizes.float().mean())), flush=True)
    del out, offs, offs2
