"""what the table look-ups cost k_encode: the same clips through the ENTRIES input mode (agmv_hip_encode_entries_dev: the
planes hold the entries, no look-up is issued; classification, emission, synchronisation and the 4 B/px stream are the same
and so are the bytes).  256 x 1080p, median of 7."""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import synth as S
from libagmv_amd import AgmvHip
W, H, T = 1920, 1080, 256
hip = AgmvHip(0)
p0, p1 = S.content_palettes([S.synth_frame(W, H, t) for t in range(2)])
hip.set_palette(p0, p1, True)
synth = hip.synth_dev(W, H, 0, T)
g = torch.Generator(device="cuda"); g.manual_seed(7)
r = lambda: torch.randint(0, 8, (T, H, W), dtype=torch.int32, device="cuda", generator=g)
clips = {"synth": synth, "noise3": synth ^ r() ^ (r() << 8) ^ (r() << 16), "flat": torch.full((T, H, W), 0x336699, dtype=torch.int32, device="cuda")}
out = torch.empty((T, hip.max_usize(W, H)), dtype=torch.uint8, device="cuda")
sizes = torch.empty(T, dtype=torch.int32, device="cuda")
def timed(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[3]
for k, fr in clips.items():
    t_pix = timed(lambda: hip.encode_dev(fr, T, W, H, out=out, sizes=sizes)); hip.check()
    ref = (sizes.clone(), out[0, :int(sizes[0])].clone())
    ent = (hip.quantise_dev(fr.reshape(-1)).to(torch.int32) & 0xFFFF).reshape(T, H, W).contiguous()
    def enc_entries():
        hip._ck(hip.L.agmv_hip_encode_entries_dev(hip.ctx, ent.data_ptr(), T, W, H, 0, out.data_ptr(), out.stride(0), sizes.data_ptr(), None, hip._stream()))
    t_ent = timed(enc_entries); hip.check()
    same = torch.equal(sizes, ref[0]) and torch.equal(out[0, :int(sizes[0])], ref[1])
    print("%-7s pixels %.3f ms | entries (no look-ups) %.3f ms | same bytes: %s" % (k, t_pix, t_ent, same), flush=True)
