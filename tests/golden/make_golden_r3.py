#!/usr/bin/env python3
"""Round-3 fixtures (build container only: needs /root/reference and oracle/_ref):  python tests/golden/make_golden_r3.py

  golden_r3.json   lut_sha          sha256 of the COMPLETE 2^24-entry colour -> entry table (u16 little-endian, colour order)
                                    as the compiled reference's AGMV_FindNearestEntry / AGMV_FindNearestColor gives it, for the
                                    random_palettes(123) pair (with the tie block p1[:8] = p0[:8]) and for the palette of the
                                    reference's foxlogo file, 512- and 256-colour modes
                   encodevideo_212  decoded-BMP hash of the reference's own AGMV_EncodeVideo(foxlogo 1..212) file
                                    (ae2330f7...), next to the file hash golden_foxlogo.json already holds
  foxlogo212.npz   the 212 foxlogo frames as the encoder sees them (uint8 R,G,B, file row order): input DATA of that flow
  golden.json      + files.c4_256_agmv_gba1_low_lzss_1920x1080: config 4 at 256 source frames of 1920x1080 through
                     AGMV_EncodeAGMV / OPT_GBA_I (121x81 scaler quirk, heavy PDIFS -> 127 encoded 120x80 frames)
Hashes and pixels only."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)
import oracles as O  # noqa: E402
import synth as S  # noqa: E402
import make_golden as M  # noqa: E402
from make_golden_foxlogo import FOXDIR, load_bmp24  # noqa: E402


def full_lut_sha(p0, p1, mode512):
    L = O.ref()
    h = hashlib.sha256()
    step = 1 << 20
    out = np.zeros(step, np.uint16)
    for c0 in range(0, 1 << 24, step):
        pix = np.arange(c0, c0 + step, dtype=np.uint32)
        L.refshim_nearest_entries(np.ascontiguousarray(p0, np.uint32), np.ascontiguousarray(p1, np.uint32), int(mode512), pix, step, out)
        h.update(out.astype("<u2").tobytes())
    return h.hexdigest()


def main():
    O.build_oracles()
    assert O.have_ref()
    meta = {"lut_sha": {}}
    fox = np.load(os.path.join(HERE, "foxlogo.npz"))
    p0, p1 = S.random_palettes(123)
    p1[:8] = p0[:8]
    for name, (a, b) in (("random123_tie8", (p0, p1)), ("foxlogo", (fox["p0"], fox["p1"]))):
        meta["lut_sha"][name] = {"m512": full_lut_sha(a, b, 1), "m256": full_lut_sha(a, b, 0)}
        print(name, meta["lut_sha"][name], flush=True)

    # ---- the 212 foxlogo frames + the decode of the reference's own file of them
    frames = np.stack([load_bmp24(os.path.join(FOXDIR, "foxlogo%d.bmp" % k)) for k in range(1, 213)])
    rgb = np.stack([(frames >> 16) & 255, (frames >> 8) & 255, frames & 255], axis=-1).astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "foxlogo212.npz"), rgb=rgb)
    with tempfile.TemporaryDirectory() as td:
        os.symlink(FOXDIR, os.path.join(td, "fr"))
        drv = M.REF_DRIVER.replace('b"fr", b"f"', 'b"fr", b"foxlogo"')
        subprocess.run([sys.executable, "-c", drv % O.REF_SO, "video", "212", "320", "240", "3", "3", "1"], cwd=td,
                       stdout=subprocess.DEVNULL, check=True)
        r = subprocess.run([sys.executable, "-c", M.REF_DECODE % O.REF_SO], cwd=td, stdout=subprocess.DEVNULL)
        data = open(os.path.join(td, "out.agmv"), "rb").read()
        nfr = int.from_bytes(data[4:8], "little")
        h = hashlib.sha256()
        for k in range(1, nfr + 1):
            h.update(open(os.path.join(td, "quick_export_%d.bmp" % k), "rb").read())
    meta["encodevideo_212"] = {"file_sha": hashlib.sha256(data).hexdigest(), "file_len": len(data), "frames": nfr,
                               "decoded_bmps_sha": h.hexdigest(), "decode_rc": r.returncode}
    print("EncodeVideo 212:", meta["encodevideo_212"], flush=True)
    json.dump(meta, open(os.path.join(HERE, "golden_r3.json"), "w"), indent=1, sort_keys=True)

    # ---- config 4 at 256 source frames
    M.FILE_CASES = [("c4_256_agmv_gba1_low_lzss_1920x1080", "agmv", 256, 1920, 1080, 5, 3, 1)]
    new = M.file_goldens()
    path = os.path.join(HERE, "golden.json")
    g = json.load(open(path))
    g["files"].update(new)
    json.dump(g, open(path, "w"), indent=1, sort_keys=True)
    print("merged", list(new))


if __name__ == "__main__":
    main()
