# debug: FOXLOGO.agmv through both decode forms, whole and in two batches, a few times; prints where nentered differs
import sys, os, numpy as np, torch, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import oracles as O
from libagmv_amd import AgmvHip
hip = AgmvHip(0, lib=os.environ.get("PROBE_LIB") or None)
data = open(os.path.join(R, "tests/golden/FOXLOGO.agmv"), "rb").read()
err, info, fr = O.oracle_decode_file(data, want_tables=True)
p0 = np.zeros(256, np.uint32); p1 = np.zeros(256, np.uint32)
finfo = O._FileInfo(); buf = np.frombuffer(data, np.uint8).copy()
O.oracle().orc_parse_header(buf, len(buf), C.byref(finfo), p0, p1)
hip.set_palette(p0, p1, True)
w, h = info.w, info.h
stride = hip.max_usize(w, h)
def run(lo, hi, tag):
    n = hi - lo
    bits = np.zeros((n, stride), np.uint8); bpos = np.zeros(n, np.int32)
    for i, f in enumerate(fr[lo:hi]):
        b = f["bitstream"][:f["bpos"]]; bits[i, :len(b)] = b
        pad = f["bitstream"][f["bpos"]:f["bpos"] + 16]; bits[i, len(b):len(b) + len(pad)] = pad
        bpos[i] = len(b)
    db = torch.from_numpy(bits).cuda(); dp = torch.from_numpy(bpos).cuda()
    exp = np.array([f["n_entered"] for f in fr[lo:hi]])
    for rep in range(4):
        offs, nent = hip.parse_dev(db, dp, n, w, h)
        fb = hip.parse_fallback_frames()
        nent2 = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        hip.decode_bitstreams_dev(db, dp, n, w, h, lo, nentered=nent2)
        torch.cuda.synchronize()
        a = nent.cpu().numpy(); b = nent2.cpu().numpy()
        print(tag, "rep", rep, "fallback", fb, "two-call != oracle:", np.nonzero(a != exp)[0].tolist(), " bitmap != oracle:", np.nonzero(b != exp)[0].tolist(), flush=True)
        for t in np.nonzero((b != exp) | (a != exp))[0][:3]:
            print("   frame", lo + t, "bpos", bpos[t], "oracle", exp[t], "two-call", a[t], "bitmap", b[t], "tail bytes", bits[t, bpos[t]-12:bpos[t]+8].tolist())
run(0, len(fr), "all")
run(0, 50, "a")
run(50, len(fr), "b")
