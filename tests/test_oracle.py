"""The oracle (oracle/agmv_oracle.c) pinned against the committed golden vectors, which were
produced by the compiled reference (tests/golden/make_golden.py), and -- where oracle/_ref is
present -- against the compiled reference directly on fresh seeded inputs. CPU only."""
import ctypes as C
import hashlib
import os

import numpy as np
import pytest

import oracles as O
import synth as S


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_nearest_entry_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "nearest.npz"))
    L = O.oracle()
    for mode, key in ((1, "e512"), (0, "e256")):
        out = np.zeros(len(g["pix"]), np.uint16)
        L.orc_quantise(g["p0"], g["p1"], mode, g["pix"], len(out), out)
        assert (out == g[key]).all()
    # tie rules: lowest index inside a palette, palette0 across palettes
    assert L.orc_find_nearest_color(g["p0"], int(g["p0"][42])) == 40
    assert L.orc_find_nearest_entry(g["p0"], g["p1"], int(g["p1"][3])) == 3


@pytest.mark.parametrize("mode", [512, 256])
def test_tiny_clip_bitstreams_golden(golden_dir, mode):
    g = np.load(os.path.join(golden_dir, "clip64x48_m%d.npz" % mode))
    W, H = 64, 48
    enc = O.OracleEncoder(W, H, mode == 512, g["p0"], g["p1"])
    off = 0
    for t, n in enumerate(g["sizes"]):
        b, e = enc.encode(S.synth_frame(W, H, t), True)
        assert len(b) == n
        assert (b == g["bytes"][off:off + n]).all(), "frame %d" % t
        if t < 2:
            assert (e == g["entries%d" % t]).all()
        off += n


@pytest.mark.parametrize("name", ["clip320x240_m512", "clip320x240_m256", "clip1280x720_m512"])
def test_clip_hashes_golden(golden, name):
    g = golden[name]
    W, H, T = g["W"], g["H"], g["T"]
    frames = [S.synth_frame(W, H, t) for t in range(T)]
    p0, p1 = S.content_palettes(frames[:4])
    assert sha(p0) == g["p0_sha"] and sha(p1) == g["p1_sha"]
    enc = O.OracleEncoder(W, H, name.endswith("512"), p0, p1)
    for t in range(T if W < 1000 else 2):
        b, e = enc.encode(frames[t], True)
        assert len(b) == g["usize"][t]
        assert sha(b) == g["bytes_sha"][t]
        assert sha(e) == g["entries_sha"][t]


def test_decode_splash_golden(golden, golden_dir):
    g = golden["agmv_splash"]
    data = open(os.path.join(golden_dir, "agmv_splash.agmv"), "rb").read()
    assert hashlib.sha256(data).hexdigest() == g["file_sha"]
    err, info, frames = O.oracle_decode_file(data)
    assert err == 0 and (info.w, info.h, info.num_frames, info.version) == (g["w"], g["h"], g["n"], g["version"])
    assert [f["usize"] for f in frames] == g["usize"]
    assert [f["bpos"] for f in frames] == g["bpos"]
    assert [sha(f["pix"]) for f in frames] == g["pix_sha"]
    # finding 3 of the survey: a good share of frames decompress to bpos != usize
    assert sum(1 for f in frames if f["bpos"] != f["usize"]) == 37


def test_foxlogo_frames_golden(golden_fox, foxlogo):
    """real content: frames 10..13 of the reference's foxlogo sample coded I,P,P,P with the palette of the reference's own
    foxlogo file, both colour modes -- the known answers of SURVEY.md Appendix C (b166e3c8..., 8b025027..., ccbfe015..., ...)"""
    fr, p0, p1 = foxlogo["frames"], foxlogo["p0"], foxlogo["p1"]
    for mode512, name in ((True, "opt3"), (False, "opt2")):
        g = golden_fox["ippp_" + name]
        enc = O.OracleEncoder(320, 240, mode512, p0, p1)
        for k in range(4):
            b, e = enc.encode(fr[9 + k], True)
            assert len(b) == g["usize"][k] and sha(b) == g["bytes_sha"][k] and sha(e) == g["entries_sha"][k], (name, k)
    assert golden_fox["ippp_opt3"]["bytes_sha"][0].startswith("b166e3c865c5fff5")
    assert golden_fox["ippp_opt2"]["bytes_sha"][3].startswith("a14138cb309803387")


def test_decode_foxlogo_sample_golden(golden_fox, golden_dir):
    """the reference's second sample stream (examples/simple_decoding/FOXLOGO.agmv: audio chunks between the frames, 64 of
    105 frames decompress to bpos != usize)"""
    g = golden_fox["FOXLOGO"]
    data = open(os.path.join(golden_dir, "FOXLOGO.agmv"), "rb").read()
    assert hashlib.sha256(data).hexdigest() == g["file_sha"]
    err, info, frames = O.oracle_decode_file(data)
    assert err == 0 and (info.w, info.h, info.num_frames, info.version) == (g["w"], g["h"], g["n"], g["version"])
    assert [f["usize"] for f in frames] == g["usize"] and [f["bpos"] for f in frames] == g["bpos"]
    assert [sha(f["pix"]) for f in frames] == g["pix_sha"]
    assert sum(1 for f in frames if f["bpos"] != f["usize"]) == g["escape_frames"] == 64


def test_old_header_rejected(golden_dir):
    data = open(os.path.join(golden_dir, "agmv_spash_header.bin"), "rb").read()
    err, _, _ = O.oracle_decode_file(data)
    assert err == 1            # INVALID_HEADER_FORMATTING_ERR, include/agmv_defines.h:39


def test_lz_golden(golden):
    L = O.oracle()
    frames = [S.synth_frame(320, 240, t) for t in range(2)]
    p0, p1 = S.content_palettes(frames)  # make_golden uses frames[:4] of a 2-frame clip
    enc = O.OracleEncoder(320, 240, True, p0, p1)
    for k, f in enumerate(frames):
        o = enc.encode(f)
        for name, fn in (("lzss", L.orc_lzss_compress), ("lz77", L.orc_lz77_compress)):
            g = golden["lz_320x240"]["%s_%d" % (name, k)]
            assert len(o) == g["n_in"] and sha(o) == g["in_sha"]
            if name == "lzss" and k == 0:
                continue   # brute force over a 28 KB I-frame takes a while; P-frame + LZ77 suffice
            xin = np.concatenate([o, np.zeros(8, np.uint8)])
            out = np.zeros(4 * len(o) + 64, np.uint8)
            cs = C.c_uint32()
            n = fn(xin, len(o), out, C.byref(cs))
            assert (n, cs.value) == (g["n_out"], g["csize"])
            assert sha(out[:n]) == g["out_sha"]


# ------------------------------------------------------------------ against the live reference
needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built here")


@needs_ref
@pytest.mark.parametrize("seed", [1, 2, 3])
def test_nearest_vs_ref(seed):
    rng = np.random.default_rng(seed)
    p0, p1 = S.random_palettes(seed, spread=(seed != 2))
    if seed == 3:
        p1[:] = p0
    pix = rng.integers(0, 1 << 24, size=30000, dtype=np.uint32)
    for mode in (1, 0):
        a = np.zeros(len(pix), np.uint16)
        b = np.zeros(len(pix), np.uint16)
        O.oracle().orc_quantise(p0, p1, mode, pix, len(pix), a)
        O.ref().refshim_nearest_entries(p0, p1, mode, pix, len(pix), b)
        assert (a == b).all()


@needs_ref
@pytest.mark.parametrize("mode512", [True, False])
@pytest.mark.parametrize("first_fc", [0, 4])
def test_encode_vs_ref(mode512, first_fc):
    W, H = 96, 64
    rng = np.random.default_rng(5)
    frames = [S.synth_frame(W, H, t) for t in range(5)]
    frames.append(rng.integers(0, 1 << 24, size=(H, W), dtype=np.uint32))   # pure noise
    frames.append(np.full((H, W), 0x123456, np.uint32))                       # flat
    p0, p1 = S.content_palettes(frames[:4])
    a = O.OracleEncoder(W, H, mode512, p0, p1, first_fc)
    b = O.RefEncoder(W, H, mode512, p0, p1, first_fc)
    for f in frames:
        x, y = a.encode(f), b.encode(f)
        assert len(x) == len(y) and (x == y).all()


@needs_ref
def test_decode_vs_ref(golden_dir):
    path = os.path.join(golden_dir, "agmv_splash.agmv")
    err, info, fr = O.oracle_decode_file(open(path, "rb").read(), want_tables=True)
    rerr, rinfo, rf = O.ref_decode_file(path, want_bitstream=True)
    assert err == rerr == 0 and len(fr) == len(rf)
    for a, b in zip(fr, rf):
        assert a["bpos"] == b["bpos"]
        assert (a["bitstream"] == b["bitstream"]).all()      # incl. the stale tail
        assert (a["pix"] == b["pix"]).all()


@needs_ref
def test_interp_vs_ref():
    rng = np.random.default_rng(9)
    a = rng.integers(0, 1 << 24, size=4096, dtype=np.uint32)
    b = rng.integers(0, 1 << 24, size=4096, dtype=np.uint32)
    x = np.zeros_like(a)
    y = np.zeros_like(a)
    O.oracle().orc_interp_frame(x, a, b, len(a))
    O.ref().refshim_interp(y, a, b, 64, 64)
    assert (x == y).all()


@needs_ref
@pytest.mark.parametrize("mode512", [True, False])
@pytest.mark.parametrize("shape", [(4, 4), (4, 24), (8, 8)])
def test_decode_narrow_frames_vs_ref(tmp_path, mode512, shape):
    """frames one block wide: the reference's last-block FILL quirk reads img_data[(x-1)+(y+1)*w] with 64-bit unsigned
    x == 0, which wraps to the block's own pixel (3,0) (src/agmv_decode.c:264-266).  File written by the reference's
    per-frame encoder, decoded by the reference and by the restatement."""
    W, H = shape
    rng = np.random.default_rng(W * 100 + H)
    frames = [np.full((H, W), int(c), np.uint32) for c in rng.integers(0, 1 << 24, 5)]
    frames += [rng.integers(0, 1 << 24, size=(H, W), dtype=np.uint32), frames[1].copy(), frames[2].copy()]
    p0, p1 = S.content_palettes(frames[:4])
    path = str(tmp_path / "narrow.agmv").encode()
    a = O.ref().refshim_create(W, H, 3 if mode512 else 2, 1, p0, p1)
    O.ref().refshim_write_header(a, path)
    for f in frames:
        O.ref().refshim_encode_frame_file(a, path, np.ascontiguousarray(f.reshape(-1)), 0)
    O.ref().refshim_destroy(a)
    data = bytearray(open(path, "rb").read())
    data[4:8] = len(frames).to_bytes(4, "little")           # num_of_frames, patched by the sequence drivers only
    open(path, "wb").write(data)
    err, info, fr = O.oracle_decode_file(bytes(data), want_tables=True)
    rerr, rinfo, rf = O.ref_decode_file(path.decode())
    assert err == rerr == 0 and len(fr) == len(rf) == len(frames)
    for t, (x, y) in enumerate(zip(fr, rf)):
        assert (x["pix"] == y["pix"]).all(), "frame %d" % t
