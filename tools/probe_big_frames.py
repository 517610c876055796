# both parser forms on large and odd geometries (4K, 8K, one block, one block row, one block column), both colour modes
import sys, os, torch, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import synth as S
from libagmv_amd import AgmvHip
hip = AgmvHip(0)
for (W, H, T) in ((3840, 2160, 8), (7680, 4320, 4), (4, 4, 9), (2052, 4, 5), (8, 1024, 6)):
    p0, p1 = S.content_palettes([S.synth_frame(min(W, 640), min(H, 480), 0)])
    for m512 in (True, False):
        hip.set_palette(p0, p1, m512)
        fr = hip.synth_dev(W, H, 0, T)
        out, sizes = hip.encode_dev(fr, T, W, H); hip.check()
        os.environ.pop("AGMV_HIP_PARSE", None)
        o1, n1 = hip.parse_dev(out, sizes, T, W, H); fb = hip.parse_fallback_frames()
        os.environ["AGMV_HIP_PARSE"] = "robust"
        o2, n2 = hip.parse_dev(out, sizes, T, W, H)
        os.environ.pop("AGMV_HIP_PARSE", None)
        nb = W * H // 16
        idx = torch.arange(nb, device="cuda")[None, :] < n1[:, None]
        ok = bool(torch.equal(n1, n2)) and bool(torch.equal(o1[idx], o2[idx])) and int(n1.min()) == nb
        dec, _, _ = hip.parse_decode_dev(out, sizes, T, W, H)
        ent = hip.quantise_dev(fr.reshape(-1))
        print(W, H, T, m512, "ok" if ok else "MISMATCH", "fallback", fb, "mean usize", int(sizes.float().mean()), flush=True)
        assert ok
