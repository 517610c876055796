"""determinism stress of k_encode: the same clip encoded many times must give the same bytes every time (a race in
the wave-to-wave protocol shows up as a run that differs).  Geometries with workgroup tiles that are mostly past the
end of the frame (empty waves) are the interesting ones.   usage: stress_enc.py [reps]"""
import hashlib, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import synth as S
from libagmv_amd import AgmvHip
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
hip = AgmvHip(0)
bad = 0
for (W, H, T) in ((1920, 1080, 128), (68, 4, 200), (2052, 4, 64), (260, 36, 90), (4, 4, 33), (516, 8, 120)):
    p0, p1 = S.content_palettes([S.synth_frame(max(W, 8), max(H, 8), t)[:H, :W] for t in range(2)])
    for mode512 in (True, False):
        hip.set_palette(p0, p1, mode512)
        frames = hip.synth_dev(W, H, 0, T)
        frames = frames ^ (torch.randint(0, 8, (T, H, W), dtype=torch.int32, device="cuda") * 0x010101 * (torch.rand((T, H, W), device="cuda") < 0.2))
        out = torch.empty((T, hip.max_usize(W, H)), dtype=torch.uint8, device="cuda")
        sizes = torch.empty(T, dtype=torch.int32, device="cuda")
        ref = None
        for r in range(reps):
            out.zero_()
            hip.encode_dev(frames, T, W, H, out=out, sizes=sizes)
            hip.check()
            sz = sizes.cpu().numpy()
            h = hashlib.sha256(sz.tobytes())
            o = out.cpu().numpy()
            for f in range(T):
                h.update(o[f, :sz[f]].tobytes())
            d = h.hexdigest()
            if ref is None:
                ref = d
            elif d != ref:
                bad += 1
                print("MISMATCH %dx%d x%d mode512=%s rep %d" % (W, H, T, mode512, r))
        print("%dx%d x%d mode512=%s: %d reps, %s" % (W, H, T, mode512, reps, ref[:16]), flush=True)
print("stress: %d mismatching runs" % bad)
sys.exit(1 if bad else 0)
