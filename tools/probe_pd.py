# parse || decode overlap: agmv_hip_parse_decode_frames_dev against the two separate calls (same pixels), by slice count
import sys, os, numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import synth as S
from libagmv_amd import AgmvHip
W, H = 1920, 1080
T = int(os.environ.get("T", "512"))
hip = AgmvHip(0)
p0, p1 = S.content_palettes([S.synth_frame(W, H, t) for t in range(2)])
hip.set_palette(p0, p1, True); hip.enable_timing(True)
frames = hip.synth_dev(W, H, 0, T)
out, sizes = hip.encode_dev(frames, T, W, H)
del frames
offs, nent = hip.parse_dev(out, sizes, T, W, H)
dec = hip.decode_dev(out, sizes, offs, nent, T, W, H)
tp, td = [], []
for _ in range(5):
    hip.parse_dev(out, sizes, T, W, H, offsets=offs, nentered=nent); tp.append(hip.last_kernel_ms(1))
    hip.decode_dev(out, sizes, offs, nent, T, W, H, out=dec); td.append(hip.last_kernel_ms(2))
ref = dec.clone(); ref_off = offs.clone()
print("separate: parse %.3f + decode %.3f = %.3f ms (%d frames)" % (sorted(tp)[2], sorted(td)[2], sorted(tp)[2] + sorted(td)[2], T), flush=True)
for arg in sys.argv[1:]:                                      # N or N:GX (GX = parser workgroups per frame, AGMV_PARSE_GX)
    ns, _, gx = arg.partition(":")
    os.environ["AGMV_DEC_SLICES"] = ns
    if gx:
        os.environ["AGMV_PARSE_GX"] = gx
    else:
        os.environ.pop("AGMV_PARSE_GX", None)
    dec.zero_(); offs.zero_()
    ts = []
    for _ in range(5):
        hip.parse_decode_dev(out, sizes, T, W, H, out=dec, offsets=offs, nentered=nent); ts.append(hip.last_kernel_ms(3))
    ok = bool(torch.equal(dec, ref)) and bool(torch.equal(offs, ref_off))
    print("slices %s: %.3f ms  same=%s" % (arg, sorted(ts)[2], ok), flush=True)
