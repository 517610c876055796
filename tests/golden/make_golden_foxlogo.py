#!/usr/bin/env python3
"""Real-content fixtures from the reference's own sample assets (build container only; needs /root/reference and
oracle/_ref):   python tests/golden/make_golden_foxlogo.py

  foxlogo.npz            frames 1..24 of examples/simple_video/foxlogo/foxlogo<k>.bmp as the encoder sees them (u32
                         0x00RRGGBB, file row order) and the palettes the reference builds from all 212 frames
                         (AGMV_EncodeVideo, OPT_III / LOW / LZSS -- the file of SURVEY.md section 4, sha ad91dc37...)
  FOXLOGO.agmv           the reference's sample stream examples/simple_decoding/FOXLOGO.agmv (105 frames + audio chunks)
  golden_foxlogo.json    what the COMPILED REFERENCE makes of them: pre-LZ bitstreams / entry planes of frames 10..13 coded
                         I,P,P,P in both colour modes, the file AGMV_EncodeAGMV writes for frames 1..24,
                         and per-frame pixel hashes of the FOXLOGO.agmv decode

Everything stored is DATA (pixels, palettes, a sample stream, hashes); inputs of the tests are rebuilt from it."""
import hashlib
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import hostlib as Hh  # noqa: E402
import oracles as O  # noqa: E402
from make_golden import REF_DECODE, REF_DRIVER, sha  # noqa: E402

REFROOT = "/root/reference"
FOXDIR = os.path.join(REFROOT, "examples", "simple_video", "foxlogo")


def load_bmp24(path):
    """24-bit uncompressed BMP -> u32 0x00RRGGBB in FILE row order (AGIDL keeps the file's bottom-up order,
    extern/agidl/src/agidl_img_bmp.c:636-655)"""
    d = open(path, "rb").read()
    off = int.from_bytes(d[10:14], "little")
    w, h = int.from_bytes(d[18:22], "little", signed=True), int.from_bytes(d[22:26], "little", signed=True)
    assert int.from_bytes(d[28:30], "little") == 24 and h > 0
    row = (w * 3 + 3) & ~3
    a = np.frombuffer(d, np.uint8, row * h, off).reshape(h, row)[:, :w * 3].reshape(h, w, 3).astype(np.uint32)
    return a[:, :, 2] << 16 | a[:, :, 1] << 8 | a[:, :, 0]


def main():
    O.build_oracles()
    assert O.have_ref()
    W, H = 320, 240
    meta = {}
    frames = [load_bmp24(os.path.join(FOXDIR, "foxlogo%d.bmp" % k)) for k in range(1, 25)]
    assert frames[0].shape == (H, W)

    # ---- the palette of the reference's own foxlogo file: AGMV_EncodeVideo over the 212 original frames
    with tempfile.TemporaryDirectory() as td:
        os.symlink(FOXDIR, os.path.join(td, "fr"))                 # short path: the reference sprintf()s into char[60]
        drv = REF_DRIVER.replace('b"fr", b"f"', 'b"fr", b"foxlogo"')
        subprocess.run([sys.executable, "-c", drv % O.REF_SO, "video", "212", str(W), str(H), "3", "3", "1"], cwd=td,
                       stdout=subprocess.DEVNULL, check=True)
        data = open(os.path.join(td, "out.agmv"), "rb").read()
    meta["encodevideo_212"] = {"file_sha": hashlib.sha256(data).hexdigest(), "file_len": len(data),
                               "frames": int.from_bytes(data[4:8], "little"), "fps_field": int.from_bytes(data[18:22], "little")}
    pal = np.frombuffer(data, np.uint8, 1536, 38).reshape(512, 3).astype(np.uint32)
    pal = pal[:, 0] << 16 | pal[:, 1] << 8 | pal[:, 2]
    p0, p1 = pal[:256].copy(), pal[256:].copy()
    print("EncodeVideo 212:", meta["encodevideo_212"])

    # ---- frames 10..13 as I,P,P,P with that palette, both colour modes
    for mode512, name in ((1, "opt3"), (0, "opt2")):
        enc = O.RefEncoder(W, H, mode512, p0, p1)
        usz, bsha, esha = [], [], []
        for k in range(4):
            b, e = enc.encode(frames[9 + k], True)
            usz.append(int(len(b))); bsha.append(sha(b)); esha.append(sha(e))
        enc.close()
        meta["ippp_" + name] = {"usize": usz, "bytes_sha": bsha, "entries_sha": esha}
        print(name, usz, [s[:8] for s in bsha])

    # ---- frames 1..24 through AGMV_EncodeAGMV (README flow), then decoded, by the reference
    with tempfile.TemporaryDirectory() as td:
        os.mkdir(os.path.join(td, "fr"))
        for k, f in enumerate(frames):
            Hh.write_bmp(os.path.join(td, "fr", "f%d.bmp" % (k + 1)), f)
        subprocess.run([sys.executable, "-c", REF_DRIVER % O.REF_SO, "agmv", "24", str(W), str(H), "3", "3", "1"], cwd=td,
                       stdout=subprocess.DEVNULL, check=True)
        r = subprocess.run([sys.executable, "-c", REF_DECODE % O.REF_SO], cwd=td, stdout=subprocess.DEVNULL)
        data = open(os.path.join(td, "out.agmv"), "rb").read()
        nfr = int.from_bytes(data[4:8], "little")
        h = hashlib.sha256()
        for k in range(1, nfr + 1):
            h.update(open(os.path.join(td, "quick_export_%d.bmp" % k), "rb").read())
    meta["encodeagmv_24"] = {"T": 24, "W": W, "H": H, "opt": 3, "quality": 3, "compression": 1, "decode_rc": r.returncode,
                             "file_sha": hashlib.sha256(data).hexdigest(), "file_len": len(data), "frames": nfr,
                             "fps_field": int.from_bytes(data[18:22], "little"), "decoded_bmps_sha": h.hexdigest()}
    print("EncodeAGMV 24:", meta["encodeagmv_24"])

    # ---- the sample stream with audio chunks
    src = os.path.join(REFROOT, "examples", "simple_decoding", "FOXLOGO.agmv")
    shutil.copyfile(src, os.path.join(HERE, "FOXLOGO.agmv"))
    os.chmod(os.path.join(HERE, "FOXLOGO.agmv"), 0o644)
    err, info, fr = O.ref_decode_file(src)
    assert err == 0
    meta["FOXLOGO"] = {"file_sha": hashlib.sha256(open(src, "rb").read()).hexdigest(), "w": info[0], "h": info[1], "n": info[2],
                       "version": info[3], "usize": [f["usize"] for f in fr], "csize": [f["csize"] for f in fr],
                       "bpos": [f["bpos"] for f in fr], "pix_sha": [sha(f["pix"]) for f in fr],
                       "escape_frames": int(sum(1 for f in fr if f["bpos"] != f["usize"]))}
    print("FOXLOGO.agmv:", info, "frames with bpos != usize:", meta["FOXLOGO"]["escape_frames"])

    np.savez_compressed(os.path.join(HERE, "foxlogo.npz"), frames=np.stack(frames).astype(np.uint32), p0=p0, p1=p1)
    json.dump(meta, open(os.path.join(HERE, "golden_foxlogo.json"), "w"), indent=1, sort_keys=True)
    print(sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
