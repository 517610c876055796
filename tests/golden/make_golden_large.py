"""File goldens of the LARGE configurations (BASELINE.json configs 4 and 5 in shape): the compiled reference
(oracle/_ref, built by oracle/Makefile from /root/reference) drives its own AGMV_EncodeAGMV over 1080p / 720p
agmv_synth_v1 BMPs -- config 4's 1920x1080 source through the GBA scaler (the 121x81-read-as-120x80 quirk,
src/agmv_encode.c:2707-2713) and a 1280x720 OPT_III clip -- and the result (file sha, decoded BMPs sha) is merged
into golden.json under "files".  Run here, in the container that holds the reference; only the JSON travels.
usage: python tests/golden/make_golden_large.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import make_golden as M

M.FILE_CASES = [
    # name, driver, T, W, H, opt, quality, compression
    ("c4_agmv_gba1_low_lzss_1920x1080", "agmv", 12, 1920, 1080, 5, 3, 1),
    ("c5_agmv_opt3_low_lzss_1280x720", "agmv", 9, 1280, 720, 3, 3, 1),
]

if __name__ == "__main__":
    M.O.build_oracles()
    assert M.O.have_ref(), "reference build missing"
    new = M.file_goldens()
    path = os.path.join(HERE, "golden.json")
    g = json.load(open(path))
    g["files"].update(new)
    json.dump(g, open(path, "w"), indent=1, sort_keys=True)
    print("merged", list(new))
