"""End-to-end wall time of the libagmv-compatible file API (BMP files -> AGMV_EncodeFullAGMV -> .agmv -> AGMV_DecodeAGMV -> BMP
files): disk + host LZ + PCIe + GPU.  Frames are written with the host library's own BMP writer and synthetic generator.
usage: e2e_time.py W H T [batch=0 (library default)] [devices=1]"""
import ctypes as C, hashlib, os, sys, tempfile, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import hostlib as H
W, Hh, T = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 0
devices = int(sys.argv[5]) if len(sys.argv) > 5 else 1
L = C.CDLL(H.SO)
L.CreateAGMV.restype = C.c_void_p; L.CreateAGMV.argtypes = [C.c_ulong] * 4
sig = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_ubyte] + [C.c_ulong] * 5 + [C.c_int] * 3
L.AGMV_EncodeFullAGMV.argtypes = sig
L.AGMV_DecodeAGMV.argtypes = [C.c_char_p, C.c_ubyte, C.c_int]
L.AGMV_SetBatchFrames.argtypes = [C.c_uint]
L.AGMV_SetDevices.argtypes = [C.c_uint]
L.AGMV_SynthFrame.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_ulonglong]
with tempfile.TemporaryDirectory(dir="/tmp") as td:
    os.chdir(td); os.mkdir("fr")
    buf = np.zeros(W * Hh, np.uint32)
    t0 = time.perf_counter()
    for t in range(1, T + 1):
        L.AGMV_SynthFrame(buf.ctypes.data, W, Hh, t, 0xA6D5)
        H.write_bmp("fr/f%d.bmp" % t, buf.reshape(Hh, W))
    t_gen = time.perf_counter() - t0
    L.AGMV_SetBatchFrames(batch)
    L.AGMV_SetDevices(devices)
    a = L.CreateAGMV(T, W, Hh, 24)
    t0 = time.perf_counter()
    L.AGMV_EncodeFullAGMV(a, b"out.agmv", b"fr", b"f", 1, 1, T, W, Hh, 24, 3, 1, 1)      # OPT_III, LOW quality, LZSS
    t_enc = time.perf_counter() - t0
    size = os.path.getsize("out.agmv")
    sha = hashlib.sha256(open("out.agmv", "rb").read()).hexdigest()[:16]
    t0 = time.perf_counter()
    rc = L.AGMV_DecodeAGMV(b"out.agmv", 1, 1)
    t_dec = time.perf_counter() - t0
    print("e2e %dx%d x %d frames (batch %d, %d GPU(s)): write inputs %.2f s | AGMV_EncodeFullAGMV %.2f s = %.1f frames/s (file %.1f MB, sha %s) | AGMV_DecodeAGMV rc=%d %.2f s = %.1f frames/s"
          % (W, Hh, T, batch, devices, t_gen, t_enc, T / t_enc, size / 1e6, sha, rc, t_dec, T / t_dec))
