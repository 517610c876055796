#!/usr/bin/env python3
"""Regenerates tests/golden/* from the COMPILED REFERENCE (oracle/_ref/libagmv_ref.so).

Run in the build container only (needs /root/reference for agmv_splash.agmv and to build
oracle/_ref):   python tests/golden/make_golden.py

What is committed is DATA: inputs are regenerated deterministically by tests/synth.py, the
expected outputs come from the reference's own compiled functions.
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracles as O  # noqa: E402
import synth as S  # noqa: E402

REFROOT = "/root/reference"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def encode_clip(W, H, T, mode512, first_frame_count=0):
    frames = [S.synth_frame(W, H, t) for t in range(T)]
    p0, p1 = S.content_palettes(frames[:4])
    enc = O.RefEncoder(W, H, mode512, p0, p1, first_frame_count)
    outs, ents = [], []
    for f in frames:
        b, e = enc.encode(f, True)
        outs.append(b)
        ents.append(e)
    enc.close()
    return frames, p0, p1, outs, ents


REF_DRIVER = """
import ctypes as C, sys
L = C.CDLL(%r)
L.CreateAGMV.restype = C.c_void_p; L.CreateAGMV.argtypes = [C.c_ulong] * 4
sig = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_ubyte] + [C.c_ulong] * 5 + [C.c_int] * 3
L.AGMV_EncodeAGMV.argtypes = sig; L.AGMV_EncodeFullAGMV.argtypes = sig
L.AGMV_EncodeVideo.argtypes = sig[1:]
L.AGMV_DecodeAGMV.argtypes = [C.c_char_p, C.c_ubyte, C.c_int]
drv, T, W, H, opt, q, comp = sys.argv[1], *[int(x) for x in sys.argv[2:]]
if drv == "video":
    L.AGMV_EncodeVideo(b"out.agmv", b"fr", b"f", 1, 1, T, W, H, 24, opt, q, comp)
else:
    a = L.CreateAGMV(T, W, H, 24)
    (L.AGMV_EncodeAGMV if drv == "agmv" else L.AGMV_EncodeFullAGMV)(a, b"out.agmv", b"fr", b"f", 1, 1, T, W, H, 24, opt, q, comp)
"""

# decode in a FRESH process: the reference's decoder mallocs img_data/iframe without clearing them
# (src/agmv_decode.c:551-557); only blocks >= the 128 KiB mmap threshold are guaranteed zero pages, and glibc raises
# that threshold once large blocks have been freed (which the encoder does)
REF_DECODE = """
import ctypes as C, sys
L = C.CDLL(%r)
L.AGMV_DecodeAGMV.argtypes = [C.c_char_p, C.c_ubyte, C.c_int]
import os
rc = L.AGMV_DecodeAGMV(b"out.agmv", 1, 1)
os._exit(rc)
"""

FILE_CASES = [
    # name, driver, T, W, H, opt, quality, compression
    ("agmv_opt3_low_lzss_160x128", "agmv", 26, 160, 128, 3, 3, 1),
    ("agmv_opt1_mid_lzss_160x128", "agmv", 26, 160, 128, 1, 2, 1),
    ("agmv_opt2_low_lz77_160x128", "agmv", 26, 160, 128, 2, 3, 2),
    ("full_opt3_high_lzss_160x128", "full", 10, 160, 128, 3, 1, 1),
    ("agmv_gba1_low_lzss_320x240", "agmv", 14, 320, 240, 5, 3, 1),
    ("agmv_nds_low_lzss_320x240", "agmv", 14, 320, 240, 8, 3, 1),
    ("video_opt3_low_lzss_160x128", "video", 26, 160, 128, 3, 3, 1),
    ("c2_agmv_opt3_low_lzss_320x240", "agmv", 212, 320, 240, 3, 3, 1),
]


def file_goldens():
    import subprocess
    import tempfile
    import hostlib as Hh
    out = {}
    for name, drv, T, W, H, opt, q, comp in FILE_CASES:
        with tempfile.TemporaryDirectory() as td:
            os.mkdir(os.path.join(td, "fr"))
            for t in range(1, T + 1):
                Hh.write_bmp(os.path.join(td, "fr", "f%d.bmp" % t), S.synth_frame(W, H, t))
            subprocess.run([sys.executable, "-c", REF_DRIVER % O.REF_SO, drv, str(T), str(W), str(H), str(opt), str(q), str(comp)],
                           cwd=td, stdout=subprocess.DEVNULL, check=True)
            r = subprocess.run([sys.executable, "-c", REF_DECODE % O.REF_SO], cwd=td, stdout=subprocess.DEVNULL)
            data = open(os.path.join(td, "out.agmv"), "rb").read()
            nfr = int.from_bytes(data[4:8], "little")
            fw, fh = int.from_bytes(data[8:12], "little"), int.from_bytes(data[12:16], "little")
            h = hashlib.sha256()
            for k in range(1, nfr + 1):
                h.update(open(os.path.join(td, "quick_export_%d.bmp" % k), "rb").read())
            out[name] = {"driver": drv, "T": T, "W": W, "H": H, "opt": opt, "quality": q, "compression": comp,
                         "decode_rc": r.returncode, "file_sha": hashlib.sha256(data).hexdigest(), "file_len": len(data),
                         "frames": nfr, "fps_field": int.from_bytes(data[18:22], "little"),
                         "decoded_bmps_sha": h.hexdigest(),
                         # frames below glibc's 128 KiB mmap threshold (8 B/px) live in recycled heap memory in the
                         # reference's decoder: pixels of blocks that are never written are garbage there
                         "decode_trusted": bool(fw * fh * 8 >= 131072 + 4096)}
            print(name, out[name]["file_len"], nfr, r.returncode)
    return out


def main():
    O.build_oracles()
    assert O.have_ref(), "reference build missing"
    meta = {}

    # ---- 1. nearest-entry known answers (E2/E3) ------------------------------------
    rng = np.random.default_rng(20241008)
    p0, p1 = S.random_palettes(11)
    p1[:16] = p0[:16]                       # cross-palette ties -> palette0 must win
    p0[40:44] = p0[40]                      # in-palette ties -> lowest index must win
    pix = rng.integers(0, 1 << 24, size=8192, dtype=np.uint32)
    pix[:256] = p0
    pix[256:512] = p1
    pix[512:520] |= np.uint32(0xAB000000)   # bits >= 24 are ignored
    e512 = np.zeros(len(pix), np.uint16)
    e256 = np.zeros(len(pix), np.uint16)
    O.ref().refshim_nearest_entries(p0, p1, 1, pix, len(pix), e512)
    O.ref().refshim_nearest_entries(p0, p1, 0, pix, len(pix), e256)
    np.savez_compressed(os.path.join(HERE, "nearest.npz"), p0=p0, p1=p1, pix=pix, e512=e512, e256=e256)

    # ---- 2. tiny clip with complete bitstreams (E4-E9), both colour modes ----------
    for mode512 in (1, 0):
        W, H, T = 64, 48, 9
        frames, p0, p1, outs, ents = encode_clip(W, H, T, mode512)
        np.savez_compressed(
            os.path.join(HERE, "clip64x48_m%d.npz" % (512 if mode512 else 256)),
            p0=p0, p1=p1, sizes=np.array([len(o) for o in outs], np.uint32),
            bytes=np.concatenate(outs), entries0=ents[0], entries1=ents[1])

    # ---- 3. larger clips: hashes only ------------------------------------------------
    for (W, H, T) in ((320, 240, 8), (1280, 720, 4)):
        for mode512 in (1, 0):
            frames, p0, p1, outs, ents = encode_clip(W, H, T, mode512)
            meta["clip%dx%d_m%d" % (W, H, 512 if mode512 else 256)] = {
                "W": W, "H": H, "T": T, "p0_sha": sha(p0), "p1_sha": sha(p1),
                "usize": [int(len(o)) for o in outs], "bytes_sha": [sha(o) for o in outs],
                "entries_sha": [sha(e) for e in ents]}

    # ---- 4. decode of the reference's own sample file (config 1) ----------------------
    src = os.path.join(REFROOT, "agmv_splash.agmv")
    shutil.copyfile(src, os.path.join(HERE, "agmv_splash.agmv"))
    os.chmod(os.path.join(HERE, "agmv_splash.agmv"), 0o644)
    err, info, fr = O.ref_decode_file(src)
    assert err == 0
    meta["agmv_splash"] = {
        "file_sha": hashlib.sha256(open(src, "rb").read()).hexdigest(),
        "w": info[0], "h": info[1], "n": info[2], "version": info[3],
        "usize": [f["usize"] for f in fr], "csize": [f["csize"] for f in fr],
        "bpos": [f["bpos"] for f in fr], "pix_sha": [sha(f["pix"]) for f in fr]}
    # header-rejection fixture: first 2 KiB of the old-layout file is enough for the header
    with open(os.path.join(REFROOT, "agmv_spash.agmv"), "rb") as f:
        open(os.path.join(HERE, "agmv_spash_header.bin"), "wb").write(f.read(2048))

    # ---- 5. LZ stage known answers (N1) ----------------------------------------------
    import ctypes as C
    frames, p0, p1, outs, ents = encode_clip(320, 240, 2, 1)
    lz = {}
    for k, o in enumerate(outs):
        xin = np.concatenate([o, np.zeros(8, np.uint8)])
        for comp, name in ((1, "lzss"), (2, "lz77")):
            out = np.zeros(4 * len(o) + 64, np.uint8)
            cs = C.c_uint32()
            n = O.ref().refshim_lz(xin, len(o), comp, out, len(out), C.byref(cs))
            lz["%s_%d" % (name, k)] = {"n_in": int(len(o)), "in_sha": sha(o), "n_out": int(n),
                                       "csize": int(cs.value), "out_sha": sha(out[:n])}
    meta["lz_320x240"] = lz

    # ---- 6. whole files through the reference's sequence drivers (N2): synthetic BMP clips -> .agmv -> BMPs
    if os.environ.get("GOLDEN_SKIP_FILES") and os.path.exists(os.path.join(HERE, "golden.json")):
        meta["files"] = json.load(open(os.path.join(HERE, "golden.json")))["files"]
    else:
        meta["files"] = file_goldens()

    # ---- 7. the per-frame FILE* API: AGMV_EncodeHeader + AGMV_EncodeFrame x n through the reference (refshim_*_file)
    import tempfile
    pf = {}
    for mode512, opt in ((1, 3), (0, 2)):
        for comp in (1, 2):
            W, H, T = 64, 48, 7
            frames = [S.synth_frame(W, H, t) for t in range(T)]
            p0, p1 = S.content_palettes(frames[:4])
            with tempfile.TemporaryDirectory() as td:
                path = os.path.join(td, "pf.agmv").encode()
                a = O.ref().refshim_create(W, H, opt, comp, p0, p1)
                O.ref().refshim_write_header(a, path)
                for f in frames:
                    O.ref().refshim_encode_frame_file(a, path, np.ascontiguousarray(f.reshape(-1)), 0)
                O.ref().refshim_destroy(a)
                data = open(path, "rb").read()
            pf["opt%d_comp%d" % (opt, comp)] = {"W": W, "H": H, "T": T, "file_sha": hashlib.sha256(data).hexdigest(), "file_len": len(data)}
    meta["per_frame_api"] = pf

    json.dump(meta, open(os.path.join(HERE, "golden.json"), "w"), indent=1, sort_keys=True)
    print("golden written:", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
