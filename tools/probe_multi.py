"""k_encode timing of several library builds in ONE process (256 x 1080p resident clips, median of 7).

usage: probe_multi.py KIND[,KIND...] NAME...    NAME -> tools/variants/libagmv_hip_NAME.so ("BASE" = the product build)
kinds: synth | noise3 | noise | flat | hgrad (see probe_enc.py) | synthpP (the first P frames of synth repeated)"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import synth as S
from libagmv_amd import AgmvHip
W, H, T = 1920, 1080, int(os.environ.get("PROBE_FRAMES", "256"))
kinds = sys.argv[1].split(",")
names = sys.argv[2:]


def lib_of(n):
    return None if n == "BASE" else os.path.join(R, "tools", "variants", "libagmv_hip_%s.so" % n)


base = AgmvHip(0)
p0, p1 = S.content_palettes([S.synth_frame(W, H, t) for t in range(2)])
synth = base.synth_dev(W, H, 0, T)
g = torch.Generator(device="cuda"); g.manual_seed(7)
clips = {}
for k in kinds:
    if k == "synth": clips[k] = synth
    elif k == "noise": clips[k] = torch.randint(0, 1 << 24, (T, H, W), dtype=torch.int32, device="cuda", generator=g)
    elif k == "flat": clips[k] = torch.full((T, H, W), 0x336699, dtype=torch.int32, device="cuda")
    elif k == "hgrad":
        x = (torch.arange(W, device="cuda", dtype=torch.int32) // 16) & 0xff
        clips[k] = (x | (x << 8) | (x << 16)).view(1, 1, W).expand(T, H, W).contiguous()
    elif k.startswith("synthp"):                               # the synthetic clip with period P frames (synthp8: two GOPs repeated): same motion
        P = int(k[6:]); clips[k] = synth[:P].repeat(T // P, 1, 1).contiguous()      # inside a GOP, but the clip's colour set is that of P frames
    elif k == "noise3":
        r = lambda: torch.randint(0, 8, (T, H, W), dtype=torch.int32, device="cuda", generator=g)
        clips[k] = synth ^ r() ^ (r() << 8) ^ (r() << 16)
    else: raise SystemExit("unknown kind " + k)
for n in names:
    hip = AgmvHip(0, lib=lib_of(n))
    hip.set_palette(p0, p1, True)
    out = torch.empty((T, hip.max_usize(W, H)), dtype=torch.uint8, device="cuda")
    sizes = torch.empty(T, dtype=torch.int32, device="cuda")
    line = "%-28s" % n
    for k in kinds:
        fr = clips[k]
        for _ in range(2): hip.encode_dev(fr, T, W, H, out=out, sizes=sizes)
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(); hip.encode_dev(fr, T, W, H, out=out, sizes=sizes); b.record(); torch.cuda.synchronize()
            ts.append(a.elapsed_time(b))
        try:
            hip.check(); err = ""
        except RuntimeError as e:
            err = " ERR(%s)" % str(e)[-40:]
        usz = sizes.cpu().numpy().astype(np.int64).mean()
        line += "  %s %.3f (min %.3f, usize %.0f)%s" % (k, sorted(ts)[3], min(ts), usz, err)
    print(line, flush=True)
    hip.close(); del out, sizes
