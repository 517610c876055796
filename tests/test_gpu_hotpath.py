"""Parity tests proper: the HIP path (through the C-ABI of include/agmv_hip.h) against the oracle
and the committed golden vectors.  Bit-exact: this is byte / index work.  Needs an MI355X."""
import hashlib
import os

import numpy as np
import pytest

import oracles as O
import synth as S

pytestmark = pytest.mark.gpu


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def torch():
    import torch
    assert torch.cuda.is_available(), "no GPU visible"
    return torch


@pytest.fixture(scope="module")
def hip(torch):
    from libagmv_amd import AgmvHip
    h = AgmvHip(0)
    yield h
    h.close()


def dev_u32(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a, np.uint32).view(np.int32)).cuda()


def to_u32(t):
    return t.cpu().numpy().view(np.uint32)


def to_u16(t):
    return t.cpu().numpy().view(np.uint16)


def gpu_encode(torch, hip, frames, first_fc=0, ientries=None):
    frames = np.ascontiguousarray(frames, np.uint32)
    n, h, w = frames.shape
    out, sizes = hip.encode_dev(dev_u32(torch, frames), n, w, h, first_fc, ientries=ientries)
    hip.check()
    sizes = sizes.cpu().numpy()
    out = out.cpu().numpy()
    return [out[i, :sizes[i]].copy() for i in range(n)]


def first_diff(a, b):
    n = min(len(a), len(b))
    d = np.nonzero(a[:n] != b[:n])[0]
    return (int(d[0]) if len(d) else n), len(a), len(b)


# ------------------------------------------------------------------------------- K0: the table
def test_lut_golden_and_random(torch, hip, golden_dir):
    g = np.load(os.path.join(golden_dir, "nearest.npz"))
    for mode, key in ((True, "e512"), (False, "e256")):
        hip.set_palette(g["p0"], g["p1"], mode)
        e = to_u16(hip.quantise_dev(dev_u32(torch, g["pix"])))
        assert (e == g[key]).all()


@pytest.mark.parametrize("which", ["random123_tie8", "foxlogo"])
def test_lut_exhaustive_all_colours(torch, hip, golden_dir, foxlogo, which):
    """The COMPLETE 2^24-entry colour -> entry table -- the one object every encoded byte depends on -- against the compiled
    reference's AGMV_FindNearestEntry / AGMV_FindNearestColor over all colours (sha256 of the table in colour order, made by
    tests/golden/make_golden_r3.py), both colour modes: a palette pair with the cross-palette tie block p1[:8] = p0[:8], and
    the palette the reference builds for its own foxlogo clip.  Plus a sample against the oracle restatement."""
    import hashlib
    import json
    g = json.load(open(os.path.join(golden_dir, "golden_r3.json")))["lut_sha"][which]
    if which == "foxlogo":
        p0, p1 = foxlogo["p0"], foxlogo["p1"]
    else:
        p0, p1 = S.random_palettes(123)
        p1[:8] = p0[:8]
    allc = np.arange(1 << 24, dtype=np.uint32)
    dall = dev_u32(torch, allc)
    for mode512, key in ((True, "m512"), (False, "m256")):
        hip.set_palette(p0, p1, mode512)
        e = to_u16(hip.quantise_dev(dall))
        assert hashlib.sha256(e.astype("<u2").tobytes()).hexdigest() == g[key], "%s %s: the table differs from the reference's" % (which, key)
        if mode512:
            sample = np.ascontiguousarray(np.concatenate([allc[5::16], allc[0x7F0000:0x800000]]))
            ref = np.zeros(len(sample), np.uint16)
            O.oracle().orc_quantise(p0, p1, 1, sample, len(sample), ref)
            assert (np.concatenate([e[5::16], e[0x7F0000:0x800000]]) == ref).all()


# ------------------------------------------------------------------------------- K1: encode
@pytest.mark.parametrize("mode", [512, 256])
def test_encode_tiny_clip_golden(torch, hip, golden_dir, mode):
    g = np.load(os.path.join(golden_dir, "clip64x48_m%d.npz" % mode))
    W, H = 64, 48
    T = len(g["sizes"])
    frames = np.stack([S.synth_frame(W, H, t) for t in range(T)])
    hip.set_palette(g["p0"], g["p1"], mode == 512)
    outs = gpu_encode(torch, hip, frames)
    off = 0
    for t, n in enumerate(g["sizes"]):
        exp = g["bytes"][off:off + n]
        assert len(outs[t]) == n and (outs[t] == exp).all(), "frame %d first diff %s" % (t, first_diff(outs[t], exp))
        off += n


@pytest.mark.parametrize("name", ["clip320x240_m512", "clip320x240_m256", "clip1280x720_m512", "clip1280x720_m256"])
def test_encode_clip_hashes_golden(torch, hip, golden, name):
    g = golden[name]
    W, H, T = g["W"], g["H"], g["T"]
    frames = np.stack([S.synth_frame(W, H, t) for t in range(T)])
    p0, p1 = S.content_palettes(frames[:4])
    hip.set_palette(p0, p1, name.endswith("512"))
    outs = gpu_encode(torch, hip, frames)
    assert [len(o) for o in outs] == g["usize"]
    assert [sha(o) for o in outs] == g["bytes_sha"]


@pytest.mark.parametrize("mode512", [True, False])
@pytest.mark.parametrize("shape", [(4, 4), (8, 8), (20, 12), (2052, 4), (68, 36), (512, 64), (280, 64), (300, 40)])
def test_encode_vs_oracle_shapes_and_content(torch, hip, mode512, shape):
    """ragged geometries (tiles straddling block rows, a single block, one block row) and
    adversarial content (noise -> all NORMAL with escapes, flat -> all FILL/COPY)."""
    W, H = shape
    rng = np.random.default_rng(W * 1000 + H)
    frames = [rng.integers(0, 1 << 24, size=(H, W), dtype=np.uint32) for _ in range(3)]
    frames += [np.full((H, W), 0x808080, np.uint32), np.full((H, W), 0x808181, np.uint32)]
    frames += [frames[0].copy(), frames[0] ^ np.uint32(0x010101)]
    frames += [S.synth_frame(max(W, 8), max(H, 8), t)[:H, :W] for t in range(4)]
    frames = np.stack(frames)
    p0, p1 = S.content_palettes(frames[:5])
    hip.set_palette(p0, p1, mode512)
    outs = gpu_encode(torch, hip, frames)
    enc = O.OracleEncoder(W, H, mode512, p0, p1)
    for t, f in enumerate(frames):
        exp = enc.encode(f)
        assert len(outs[t]) == len(exp) and (outs[t] == exp).all(), "frame %d first diff %s" % (t, first_diff(outs[t], exp))


@pytest.mark.parametrize("mode512", [True, False])
def test_encode_static_pictures(torch, hip, mode512):
    """Static pictures (every P-frame block is a COPY, src/agmv_encode.c:462-467): whole frames repeated inside a GOP and
    across a GOP boundary, a picture that is static in its left part only (waves of 64 blocks entirely inside it, entirely
    outside, straddling), pixels that differ only above bit 23, a batch that starts inside a GOP, against the oracle."""
    W, H = 1024, 32                                            # 256 blocks per block row: four waves per row
    rng = np.random.default_rng(77)
    base = rng.integers(0, 1 << 24, size=(H, W), dtype=np.uint32)
    base[:, 640:] = 0x203040                                   # flat right part
    frames = [base.copy() for _ in range(12)]
    frames[2] = base ^ np.uint32(0x01000000)                   # same colours, different top byte
    for t in (5, 6, 7, 9):                                     # GOP 4..7 / 8..11: left 5/8 static, the rest moves
        frames[t] = base.copy()
        frames[t][:, 640:] = 0x203040 + 0x010101 * t
    frames[10][:, 100:164] ^= np.uint32(0x3)                   # one wave's worth of change inside the static part
    frames = np.stack(frames)
    p0, p1 = S.content_palettes(frames[:4])
    hip.set_palette(p0, p1, mode512)
    enc = O.OracleEncoder(W, H, mode512, p0, p1)
    exp = [enc.encode(f) for f in frames]
    outs = gpu_encode(torch, hip, frames)
    for t in range(len(frames)):
        assert len(outs[t]) == len(exp[t]) and (outs[t] == exp[t]).all(), "frame %d first diff %s" % (t, first_diff(outs[t], exp[t]))
    # the same clip in two batches, the second one starting inside GOP 4..7 with the I-frame's entries handed over
    d = torch.from_numpy(frames.view(np.int32)).cuda()
    ient = torch.zeros(W * H, dtype=torch.int16, device="cuda")
    o1, s1 = hip.encode_dev(d[:6], 6, W, H, 0, ientries=ient)
    o2, s2 = hip.encode_dev(d[6:], 6, W, H, 6, ientries=ient)
    hip.check()
    got = [o1[t, :int(s1[t])].cpu().numpy() for t in range(6)] + [o2[t, :int(s2[t])].cpu().numpy() for t in range(6)]
    for t in range(12):
        assert len(got[t]) == len(exp[t]) and (got[t] == exp[t]).all(), "two batches, frame %d" % t


def test_encode_batches_continue_a_gop(torch, hip):
    """frame_count continuity: encoding 11 frames as batches of 1,2,5,3 (starting inside GOPs, with
    the I-frame entry plane carried between calls like agmv->iframe_entries) equals one batch."""
    W, H = 96, 64
    frames = np.stack([S.synth_frame(W, H, t) for t in range(11)])
    p0, p1 = S.content_palettes(frames[:4])
    hip.set_palette(p0, p1, True)
    whole = gpu_encode(torch, hip, frames)
    ient = torch.zeros(W * H, dtype=torch.int16, device="cuda")
    got, fc = [], 0
    for n in (1, 2, 5, 3):
        got += gpu_encode(torch, hip, frames[fc:fc + n], first_fc=fc, ientries=ient)
        fc += n
    for t in range(11):
        assert len(got[t]) == len(whole[t]) and (got[t] == whole[t]).all(), t
    enc = O.OracleEncoder(W, H, True, p0, p1)
    for t in range(11):
        assert (enc.encode(frames[t]) == whole[t]).all()


def test_encode_1080p_vs_oracle(torch, hip):
    W, H = 1920, 1080
    frames = np.stack([S.synth_frame(W, H, t) for t in range(5)])
    p0, p1 = S.content_palettes(frames[:4])
    hip.set_palette(p0, p1, True)
    outs = gpu_encode(torch, hip, frames)
    enc = O.OracleEncoder(W, H, True, p0, p1)
    for t in (0, 1):
        exp = enc.encode(frames[t])
        assert len(outs[t]) == len(exp) and (outs[t] == exp).all(), "frame %d first diff %s" % (t, first_diff(outs[t], exp))


def test_host_buffer_entry_points(torch, hip):
    W, H = 64, 48
    frames = np.stack([S.synth_frame(W, H, t) for t in range(6)])
    p0, p1 = S.content_palettes(frames[:4])
    hip.set_palette(p0, p1, True)
    outs = hip.encode_host(frames)
    enc = O.OracleEncoder(W, H, True, p0, p1)
    dec = O.OracleDecoder(W, H, True, p0, p1)
    exp_pix = []
    for t in range(6):
        exp = enc.encode(frames[t])
        assert (outs[t] == exp).all()
        exp_pix.append(dec.decode(exp))
    got = hip.decode_host(outs, W, H)
    for t in range(6):
        assert (got[t].reshape(-1) == exp_pix[t]).all(), t


# ------------------------------------------------------------------------------- K2-K4: decode
def gpu_decode(torch, hip, bits_list, pads, w, h, first_fc=0, prev=None, prev_iframe=None):
    n = len(bits_list)
    stride = (max(len(b) for b in bits_list) + 16 + 255) & ~255
    bits = np.zeros((n, stride), np.uint8)
    bpos = np.zeros(n, np.int32)
    for i, b in enumerate(bits_list):
        bits[i, :len(b)] = b
        if pads is not None:
            bits[i, len(b):len(b) + 16] = pads[i]
        bpos[i] = len(b)
    dbits = torch.from_numpy(bits).cuda()
    dbpos = torch.from_numpy(bpos).cuda()
    offs, nent = hip.parse_dev(dbits, dbpos, n, w, h)
    dprev = dev_u32(torch, prev) if prev is not None else None
    dprevi = dev_u32(torch, prev_iframe) if prev_iframe is not None else None
    out = hip.decode_dev(dbits, dbpos, offs, nent, n, w, h, first_fc, prev=dprev, prev_iframe=dprevi)
    same_without_offsets(torch, hip, dbits, dbpos, n, w, h, first_fc, dprev, dprevi, out, nent)
    torch.cuda.synchronize()
    return to_u32(out).reshape(n, -1), to_u32(offs), nent.cpu().numpy()


def same_without_offsets(torch, hip, dbits, dbpos, n, w, h, first_fc, dprev, dprevi, ref_out, ref_nent):
    """agmv_hip_decode_bitstreams_dev (entry bitmaps straight into k_decode, no offsets[]) must give the pixels and the
    nentered[] of the two-call form -- every decode test goes through here, damaged streams included"""
    nent2 = torch.full((n,), -1, dtype=torch.int32, device=dbits.device)
    out2 = hip.decode_bitstreams_dev(dbits, dbpos, n, w, h, first_fc, nentered=nent2, prev=dprev, prev_iframe=dprevi)
    torch.cuda.synchronize()
    assert torch.equal(nent2, ref_nent), "nentered: bitmap form differs"
    if not torch.equal(out2, ref_out):
        bad = (out2 != ref_out).reshape(n, -1).any(dim=1).nonzero().flatten().tolist()
        raise AssertionError("pixels: bitmap form differs in frames %s" % bad[:8])


@pytest.mark.parametrize("mode512", [True, False])
def test_roundtrip_vs_oracle(torch, hip, mode512):
    W, H = 320, 240
    frames = np.stack([S.synth_frame(W, H, t) for t in range(9)])
    p0, p1 = S.content_palettes(frames[:4])
    hip.set_palette(p0, p1, mode512)
    outs = gpu_encode(torch, hip, frames)
    dec = O.OracleDecoder(W, H, mode512, p0, p1)
    exp, exp_off = [], []
    for b in outs:
        pix, padded, offs, n_ent = dec.decode(b, want_tables=True)
        assert n_ent == W * H // 16
        exp.append(pix)
        exp_off.append(offs)
    got, offs, nent = gpu_decode(torch, hip, outs, None, W, H)
    assert (nent == W * H // 16).all()
    for t in range(9):
        assert (offs[t] == exp_off[t]).all(), "offsets frame %d" % t
        assert (got[t] == exp[t]).all(), "pixels frame %d" % t


@pytest.mark.parametrize("which", ["agmv_splash", "FOXLOGO"])
def test_decode_reference_sample_file(torch, hip, golden, golden_fox, golden_dir, which):
    """the reference's own sample streams (config 1's agmv_splash.agmv; examples/simple_decoding/FOXLOGO.agmv with audio
    chunks): the host oracle runs the LZ stage (host-side by design) and hands the decompressed buffers -- INCLUDING the
    stale bytes beyond bpos -- to the GPU; 37 of 119 / 64 of 105 frames raise `escape`, so stale tails, the fix-up pass and
    the last-block quirk are all hit."""
    g = golden["agmv_splash"] if which == "agmv_splash" else golden_fox["FOXLOGO"]
    data = open(os.path.join(golden_dir, which + ".agmv"), "rb").read()
    err, info, fr = O.oracle_decode_file(data, want_tables=True)
    assert err == 0
    p0 = np.zeros(256, np.uint32)
    p1 = np.zeros(256, np.uint32)
    import ctypes as C
    finfo = O._FileInfo()
    buf = np.frombuffer(data, np.uint8).copy()
    O.oracle().orc_parse_header(buf, len(buf), C.byref(finfo), p0, p1)
    hip.set_palette(p0, p1, True)
    bits = [f["bitstream"][:f["bpos"]] for f in fr]
    pads = [f["bitstream"][f["bpos"]:f["bpos"] + 16] for f in fr]
    got, offs, nent = gpu_decode(torch, hip, bits, pads, info.w, info.h)
    assert [int(x) for x in nent] == [f["n_entered"] for f in fr]
    for t, f in enumerate(fr):
        ne = f["n_entered"]
        assert (offs[t][:ne] == f["offsets"][:ne]).all(), "offsets frame %d" % t
    bad = [t for t in range(len(fr)) if sha(got[t]) != g["pix_sha"][t]]
    assert not bad, "frames differing from the reference: %s" % bad[:10]
    # same clip decoded in two batches, decoder state (img_data / iframe) carried across
    k = 50
    a, _, _ = gpu_decode(torch, hip, bits[:k], pads[:k], info.w, info.h)
    last_i = (k - 1) // 4 * 4
    b, _, _ = gpu_decode(torch, hip, bits[k:], pads[k:], info.w, info.h, first_fc=k, prev=a[k - 1], prev_iframe=a[last_i])
    assert [sha(x) for x in list(a) + list(b)] == g["pix_sha"]


def test_encode_foxlogo_frames_golden(torch, hip, golden_fox, foxlogo):
    """real content: foxlogo frames 10..13 as I,P,P,P with the reference-built foxlogo palette --
    duplicate zero slots in the palettes, FILL / COPY counts on the borderline, both colour modes; then all 24 stored frames
    in one batch against the oracle"""
    fr, p0, p1 = foxlogo["frames"], foxlogo["p0"], foxlogo["p1"]
    for mode512, name in ((True, "opt3"), (False, "opt2")):
        g = golden_fox["ippp_" + name]
        hip.set_palette(p0, p1, mode512)
        ient = torch.zeros(320 * 240, dtype=torch.int16, device="cuda")
        outs = gpu_encode(torch, hip, fr[9:13], ientries=ient)
        assert [len(o) for o in outs] == g["usize"]
        assert [sha(o) for o in outs] == g["bytes_sha"]
        ent = to_u16(hip.quantise_dev(dev_u32(torch, fr[9])))
        if not mode512:
            ent = ent & 0xFF
        assert sha(ent) == g["entries_sha"][0]
        whole = gpu_encode(torch, hip, fr)
        enc = O.OracleEncoder(320, 240, mode512, p0, p1)
        for t in range(len(fr)):
            exp = enc.encode(fr[t])
            assert len(whole[t]) == len(exp) and (whole[t] == exp).all(), (name, t)


def test_decode_truncated_and_garbage_streams(torch, hip):
    """fault tolerance of the parser: truncated tails, garbage prefixes that need the
    flag resync, stale bytes that look like flags."""
    W, H = 64, 48
    rng = np.random.default_rng(77)
    frames = np.stack([S.synth_frame(W, H, t) for t in range(8)])
    for mode512 in (True, False):
        p0, p1 = S.content_palettes(frames[:4])
        hip.set_palette(p0, p1, mode512)
        outs = gpu_encode(torch, hip, frames)
        bits = []
        for t, b in enumerate(outs):
            b = b.copy()
            if t % 4 == 1:
                b = b[:len(b) - int(rng.integers(1, 40))]
            elif t % 4 == 2:
                b = np.concatenate([rng.integers(0, 256, 5, dtype=np.uint8), b])[:len(b)]
            elif t % 4 == 3:
                b[int(rng.integers(0, len(b)))] = 0x5E
            bits.append(b)
        dec = O.OracleDecoder(W, H, mode512, p0, p1)
        exp, pads = [], []
        for b in bits:
            pix, padded, offs, n_ent = dec.decode(b, want_tables=True)
            exp.append(pix)
            pads.append(padded[len(b):len(b) + 16])
        got, _, _ = gpu_decode(torch, hip, bits, pads, W, H)
        for t in range(len(bits)):
            assert (got[t] == exp[t]).all(), "mode512=%s frame %d" % (mode512, t)


@pytest.mark.parametrize("seed", range(int(os.environ.get("AGMV_FUZZ_SEEDS", "10"))))
def test_fuzz_encode_decode_vs_oracle(torch, hip, seed):
    """seeded fuzz: random geometry, frame count, colour mode and content mix, encode against the oracle encoder; then
    the streams -- a third of them damaged (cut, bytes flipped to flag values, garbage spliced in) -- through the GPU
    parser + reconstruct against the oracle decoder (resync, escape, stale tails, fix-up, last-block quirk)."""
    rng = np.random.default_rng(1000 + seed)
    W, H = 4 * int(rng.integers(1, 90)), 4 * int(rng.integers(1, 40))
    n = int(rng.integers(1, 14))
    mode512 = bool(rng.integers(0, 2))
    frames, prevf = [], None
    for t in range(n):
        kind = int(rng.integers(0, 6))
        if kind == 0 or prevf is None and kind == 3:
            f = rng.integers(0, 1 << 24, size=(H, W), dtype=np.uint32)
        elif kind == 1:
            f = np.full((H, W), int(rng.integers(0, 1 << 24)), np.uint32)
        elif kind == 2:
            f = S.synth_frame(max(W, 8), max(H, 8), int(rng.integers(0, 50)))[:H, :W].copy()
        elif kind == 3:
            f = prevf.copy()
        elif kind == 4:
            f = (prevf if prevf is not None else np.zeros((H, W), np.uint32)) ^ rng.integers(0, 4, size=(H, W), dtype=np.uint32) * np.uint32(0x010101)
        else:
            f = np.repeat(np.repeat(rng.integers(0, 1 << 24, size=((H + 7) // 8, (W + 7) // 8), dtype=np.uint32), 8, 0), 8, 1)[:H, :W].copy()
        frames.append(f.astype(np.uint32))
        prevf = frames[-1]
    frames = np.stack(frames)
    p0, p1 = S.content_palettes(frames[:4]) if rng.integers(0, 2) else S.random_palettes(int(rng.integers(0, 1 << 30)))
    hip.set_palette(p0, p1, mode512)
    outs = gpu_encode(torch, hip, frames)
    enc = O.OracleEncoder(W, H, mode512, p0, p1)
    for t, f in enumerate(frames):
        exp = enc.encode(f)
        assert len(outs[t]) == len(exp) and (outs[t] == exp).all(), "encode frame %d first diff %s" % (t, first_diff(outs[t], exp))
    bits = []
    for b in outs:
        b = b.copy()
        hurt = int(rng.integers(0, 9))
        if hurt == 0 and len(b) > 2:
            b = b[:int(rng.integers(1, len(b)))]
        elif hurt == 1:
            for _ in range(int(rng.integers(1, 6))):
                b[int(rng.integers(0, len(b)))] = [0x4E, 0x2F, 0x5E, 0x7F, 0xFF][int(rng.integers(0, 5))]
        elif hurt == 2:
            room = W * H * 3 + 64 - 16 - len(b)                # the oracle decoder's persistent buffer (w*h*3+64 bytes)
            if room >= 1:
                at = int(rng.integers(0, len(b)))
                b = np.concatenate([b[:at], rng.integers(0, 256, int(rng.integers(1, min(70, room) + 1)), dtype=np.uint8), b[at:]])
        bits.append(b)
    dec = O.OracleDecoder(W, H, mode512, p0, p1)
    exp, pads = [], []
    for b in bits:
        pix, padded, offs, n_ent = dec.decode(b, want_tables=True)
        exp.append(pix)
        pads.append(padded[len(b):len(b) + 16])
    got, _, _ = gpu_decode(torch, hip, bits, pads, W, H)
    for t in range(n):
        assert (got[t] == exp[t]).all(), "decode frame %d of %d (%dx%d, mode512=%s)" % (t, n, W, H, mode512)
    # the same clip in random batches, state carried between calls like agmv->iframe_entries (encode) and
    # frame->img_data / iframe->img_data (decode): batches that start inside a GOP, single-frame batches
    cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n + 1, int(rng.integers(0, 4)))]))
    ient = torch.zeros(W * H, dtype=torch.int16, device="cuda")
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        part = gpu_encode(torch, hip, frames[lo:hi], first_fc=lo, ientries=ient)
        for t in range(lo, hi):
            assert len(part[t - lo]) == len(outs[t]) and (part[t - lo] == outs[t]).all(), "batched encode frame %d (batch %d..%d)" % (t, lo, hi)
        prev = exp[lo - 1].reshape(H, W) if lo else None
        previ = exp[(lo - 1) // 4 * 4].reshape(H, W) if lo else None
        gotb, _, _ = gpu_decode(torch, hip, bits[lo:hi], pads[lo:hi], W, H, first_fc=lo, prev=prev, prev_iframe=previ)
        for t in range(lo, hi):
            assert (gotb[t - lo] == exp[t]).all(), "batched decode frame %d (batch %d..%d)" % (t, lo, hi)


@pytest.mark.parametrize("mode512", [True, False])
@pytest.mark.parametrize("shape", [(4, 4), (4, 24), (4, 1028), (8, 8)])
def test_decode_narrow_frames_vs_oracle(torch, hip, mode512, shape):
    """one block per row (w == 4): the last-block FILL quirk falls on the block's OWN pixel (3,0) of the previous frame
    (the reference's 64-bit x-1 wraps); pinned against the compiled reference in tests/test_oracle.py."""
    W, H = shape
    rng = np.random.default_rng(W * 7 + H)
    frames = [np.full((H, W), int(c), np.uint32) for c in rng.integers(0, 1 << 24, 5)]
    noise = rng.integers(0, 1 << 24, size=(H, W), dtype=np.uint32)
    frames += [noise, frames[1].copy(), frames[2].copy(), noise ^ np.uint32(1)]
    frames = np.stack(frames)
    p0, p1 = S.content_palettes(frames[:4])
    hip.set_palette(p0, p1, mode512)
    outs = gpu_encode(torch, hip, frames)
    dec = O.OracleDecoder(W, H, mode512, p0, p1)
    exp = [dec.decode(b) for b in outs]
    got, _, _ = gpu_decode(torch, hip, outs, None, W, H)
    for t in range(len(frames)):
        assert (got[t] == exp[t]).all(), "frame %d" % t


def test_parallel_parser_matches_serial_walk(torch, hip, monkeypatch):
    """the chunked pointer-doubling parser (k_parse_chunks/stitch/emit) against the one-lane-per-frame
    walk (k_parse_serial) on streams the CPU oracle is too slow for: 1080p clips (COPY-heavy P-frames =
    1-byte nodes, FILL, NORMAL with escapes), pure noise bytes (constant flag resync), and truncations."""
    W, H, T = 1920, 1080, 8
    nblk = W * H // 16
    frames = hip.synth_dev(W, H, 0, T)
    noise = torch.randint(0, 1 << 24, (2, H, W), dtype=torch.int32, device="cuda")
    frames = torch.cat([frames, noise])
    T += 2
    for mode512 in (True, False):
        p0, p1 = S.content_palettes([S.synth_frame(W, H, 0)])
        hip.set_palette(p0, p1, mode512)
        out, sizes = hip.encode_dev(frames, T, W, H)
        hip.check()
        # add two garbage "streams" and truncate two real ones
        out = torch.cat([out, torch.randint(0, 256, (2, out.shape[1]), dtype=torch.uint8, device="cuda")])
        sizes = torch.cat([sizes, torch.tensor([300000, 5], dtype=torch.int32, device="cuda")])
        sizes[1] = sizes[1] - 7
        sizes[2] = 1000
        n = T + 2
        monkeypatch.setenv("AGMV_HIP_PARSE", "serial")
        o_ser, n_ser = hip.parse_dev(out, sizes, n, W, H)
        torch.cuda.synchronize()
        ne = n_ser.cpu().numpy()
        for how in (None, "robust"):                          # speculative walks + proof (default), the map/stitch/emit kernels alone
            if how:
                monkeypatch.setenv("AGMV_HIP_PARSE", how)
            else:
                monkeypatch.delenv("AGMV_HIP_PARSE")
            o_par, n_par = hip.parse_dev(out, sizes, n, W, H)
            torch.cuda.synchronize()
            assert torch.equal(n_ser, n_par), (how, n_ser.cpu().numpy(), n_par.cpu().numpy())
            for f in range(n):
                assert torch.equal(o_ser[f, :ne[f]], o_par[f, :ne[f]]), "%s mode512=%s frame %d (nentered %d of %d)" % (how, mode512, f, ne[f], nblk)
        monkeypatch.delenv("AGMV_HIP_PARSE", raising=False)


@pytest.mark.parametrize("seed", range(int(os.environ.get("AGMV_FUZZ_SEEDS", "6"))))
def test_fuzz_parser_random_bytes(torch, hip, monkeypatch, seed):
    """parallel parser against the one-lane serial walk on BYTE SOUP: every stream is random bytes drawn from a
    distribution rich in flag values, escape codes and near-misses, with random lengths (chunk boundaries, empty,
    one byte) and random stale tails -- the map/stitch/emit logic with no help from well-formed structure."""
    rng = np.random.default_rng(5000 + seed)
    W, H = 4 * int(rng.integers(1, 200)), 4 * int(rng.integers(1, 120))
    n = int(rng.integers(1, 9))
    mode512 = bool(rng.integers(0, 2))
    stride = int(rng.choice([256, 512, 768, 1024, 4096, 20480, 66048]))
    alphabet = np.array([0x4E, 0x2F, 0x5E, 0x7F, 0xFF, 0x4F, 0x2E, 0x5F, 0x00, 0x80] + list(rng.integers(0, 256, 6)), np.uint8)
    probs = rng.dirichlet(np.ones(len(alphabet)) * float(rng.choice([0.3, 1.0, 5.0])))
    bits = rng.choice(alphabet, size=(n, stride), p=probs).astype(np.uint8)
    mix = rng.random((n, stride)) < float(rng.choice([0.0, 0.1, 0.5]))
    bits[mix] = rng.integers(0, 256, int(mix.sum()), dtype=np.uint8)
    bpos = np.array([int(rng.choice([0, 1, 2, 511, 512, 513, 1023, 1024, stride - 16, int(rng.integers(0, stride - 15))])) for _ in range(n)], np.int32)
    bpos = np.clip(bpos, 0, stride - 16)
    p0, p1 = S.random_palettes(seed)
    hip.set_palette(p0, p1, mode512)
    exp = None
    if W * H <= 40000:                                        # small enough for the CPU oracle: pixels must match it too
        bpos = np.minimum(bpos, W * H * 3 + 64 - 16).astype(np.int32)       # its persistent buffer
        dec, exp = O.OracleDecoder(W, H, mode512, p0, p1), []
        for f in range(n):
            pix, padded, _, _ = dec.decode(bits[f, :bpos[f]], want_tables=True)
            bits[f, bpos[f]:bpos[f] + 16] = padded[bpos[f]:bpos[f] + 16]    # what the reference's buffer holds there
            exp.append(pix)
    dbits, dbpos = torch.from_numpy(bits).cuda(), torch.from_numpy(bpos).cuda()
    monkeypatch.setenv("AGMV_HIP_PARSE", "serial")
    o_ser, n_ser = hip.parse_dev(dbits, dbpos, n, W, H)
    torch.cuda.synchronize()
    ne = n_ser.cpu().numpy()
    for how in (None, "robust"):
        if how:
            monkeypatch.setenv("AGMV_HIP_PARSE", how)
        else:
            monkeypatch.delenv("AGMV_HIP_PARSE")
        o_par, n_par = hip.parse_dev(dbits, dbpos, n, W, H)
        torch.cuda.synchronize()
        assert torch.equal(n_ser, n_par), (how, ne, n_par.cpu().numpy(), bpos)
        for f in range(n):
            assert torch.equal(o_ser[f, :ne[f]], o_par[f, :ne[f]]), "%s frame %d bpos %d nentered %d (%dx%d mode512=%s stride %d)" % (how, f, bpos[f], ne[f], W, H, mode512, stride)
    monkeypatch.delenv("AGMV_HIP_PARSE", raising=False)
    # and the reconstruct step must agree with itself on both tables (fix-up path: almost every block is stale here)
    a = hip.decode_dev(dbits, dbpos, o_ser, n_ser, n, W, H, 0)
    b = hip.decode_dev(dbits, dbpos, o_par, n_par, n, W, H, 0)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    same_without_offsets(torch, hip, dbits, dbpos, n, W, H, 0, None, None, a, n_ser)
    monkeypatch.setenv("AGMV_HIP_PARSE", "robust")            # ... and with every frame's entry bits from the robust kernels
    same_without_offsets(torch, hip, dbits, dbpos, n, W, H, 0, None, None, a, n_ser)
    monkeypatch.delenv("AGMV_HIP_PARSE", raising=False)
    if exp is not None:
        got = to_u32(b).reshape(n, -1)
        for f in range(n):
            assert (got[f] == exp[f]).all(), "pixels frame %d bpos %d (%dx%d mode512=%s)" % (f, bpos[f], W, H, mode512)


@pytest.mark.parametrize("seed", range(int(os.environ.get("AGMV_FUZZ_SEEDS", "6"))))
def test_fuzz_parser_block_sequences(torch, hip, monkeypatch, seed):
    """the fast parser's STRETCH steps (a run of COPY / two- and three-byte FILL blocks taken at once) against the one-lane
    serial walk on streams that are sequences of blocks -- long clean runs, FILL bodies that hold flag values (0x4E 0x4E,
    0x4E 0x5E ...), escape codes, NORMAL blocks with flag-valued and escaped codes, stray bytes between blocks -- cut at
    every kind of bpos; both parser forms, both decode forms."""
    rng = np.random.default_rng(9000 + seed)
    W, H = 4 * int(rng.integers(8, 120)), 4 * int(rng.integers(8, 80))
    nblk = W * H // 16
    n = int(rng.integers(2, 7))
    mode512 = bool(rng.integers(0, 2)) if seed else True
    stride = hip.max_usize(W, H)
    p_copy, p_fill, p_norm, p_junk = rng.dirichlet([4, 4, 1, 0.3]) if seed % 3 else (0.45, 0.5, 0.04, 0.01)
    spicy = np.array([0x4E, 0x5E, 0x2F, 0x7F, 0xFF, 0x7E, 0x00], np.uint8)

    def code():
        c = int(rng.choice(spicy)) if rng.random() < 0.25 else int(rng.integers(0, 256))
        if mode512 and (c & 0x7F) == 127:
            return [c, int(rng.integers(127, 256))]
        return [c]

    bits = np.zeros((n, stride), np.uint8)
    bpos = np.zeros(n, np.int32)
    for f in range(n):
        buf = []
        blocks = int(nblk * rng.choice([0.3, 1.0, 1.0, 1.2]))
        for _ in range(blocks):
            r = rng.random()
            if r < p_copy:
                buf.append(0x5E)
            elif r < p_copy + p_fill:
                buf.append(0x4E); buf += code()
            elif r < p_copy + p_fill + p_norm:
                buf.append(0x2F)
                for _ in range(16):
                    buf += code()
            else:
                buf += [int(x) for x in rng.integers(0, 256, int(rng.integers(1, 4)))]
            if len(buf) > stride - 64:
                break
        buf = np.array(buf[:stride - 32], np.uint8)
        bits[f, :len(buf)] = buf
        bpos[f] = int(rng.choice([len(buf), len(buf), max(0, len(buf) - int(rng.integers(0, 70))), int(rng.integers(0, len(buf) + 1))]))
        bits[f, len(buf):len(buf) + 16] = rng.integers(0, 256, 16)          # stale bytes behind the stream
    p0, p1 = S.random_palettes(seed)
    hip.set_palette(p0, p1, mode512)
    dbits, dbpos = torch.from_numpy(bits).cuda(), torch.from_numpy(bpos).cuda()
    monkeypatch.setenv("AGMV_HIP_PARSE", "serial")
    o_ser, n_ser = hip.parse_dev(dbits, dbpos, n, W, H)
    torch.cuda.synchronize()
    ne = n_ser.cpu().numpy()
    for how in (None, "robust"):
        if how:
            monkeypatch.setenv("AGMV_HIP_PARSE", how)
        else:
            monkeypatch.delenv("AGMV_HIP_PARSE")
        o_par, n_par = hip.parse_dev(dbits, dbpos, n, W, H)
        torch.cuda.synchronize()
        assert torch.equal(n_ser, n_par), (how, ne, n_par.cpu().numpy(), bpos)
        for f in range(n):
            assert torch.equal(o_ser[f, :ne[f]], o_par[f, :ne[f]]), "%s frame %d bpos %d nentered %d (%dx%d mode512=%s)" % (how, f, bpos[f], ne[f], W, H, mode512)
    monkeypatch.delenv("AGMV_HIP_PARSE", raising=False)
    a = hip.decode_dev(dbits, dbpos, o_ser, n_ser, n, W, H, 0)
    torch.cuda.synchronize()
    same_without_offsets(torch, hip, dbits, dbpos, n, W, H, 0, None, None, a, n_ser)


@pytest.mark.parametrize("mode512", [True, False])
def test_parser_proof_repair_and_fallback(torch, hip, monkeypatch, mode512):
    """The speculative parser on streams built to defeat the speculation.  "4E 4E 4E 4E ..." is a run of two-byte FILLs
    whose index byte is itself a flag value: a walk that starts on the wrong byte parity follows a second chain through
    the whole run and never merges.  (a) a clean encoder frame: proven as walked; (b) a few such runs across region
    boundaries inside COPY filler: the regions behind them are walked again with a forced entry and the frame is proven;
    (c) a whole frame of it: given up, parsed by the map/stitch/emit kernels.  All against the one-lane serial walk."""
    W, H = 1920, 1080
    nblk = W * H // 16
    p0, p1 = S.content_palettes([S.synth_frame(W, H, 0)])
    hip.set_palette(p0, p1, mode512)
    out, sizes = hip.encode_dev(hip.synth_dev(W, H, 0, 1), 1, W, H)
    hip.check()
    stride = out.shape[1]
    a = out[0].cpu().numpy()
    region = 59 * 64
    b = np.full(stride, 0x5E, np.uint8)                       # COPY filler, one block per byte
    pos, blocks = 0, 0
    for k in (1, 2, 3, 5, 8, 13, 14, 15):                     # runs of 500 FILLs that start 775 bytes before a region boundary, on an odd byte
        start = k * region - 775
        assert start > pos and start % 2 == 1
        blocks += start - pos
        b[start:start + 1000] = 0x4E
        pos = start + 1000
        blocks += 500
    usize_b = pos + (nblk - blocks)
    assert usize_b < stride - 64
    c = np.full(stride, 0x4E, np.uint8)                       # one COPY, then FILL(entry 0x4E) for every other block: the true chain is on odd bytes
    c[0] = 0x5E
    usize_c = 1 + 2 * (nblk - 1)
    bits = torch.from_numpy(np.stack([a, b, c])).cuda()
    bpos = torch.tensor([int(sizes[0]), usize_b, usize_c], dtype=torch.int32, device="cuda")
    monkeypatch.setenv("AGMV_HIP_PARSE", "serial")
    o_ser, n_ser = hip.parse_dev(bits, bpos, 3, W, H)
    torch.cuda.synchronize()
    assert n_ser.cpu().tolist() == [nblk, nblk, nblk]
    monkeypatch.delenv("AGMV_HIP_PARSE")
    o_par, n_par = hip.parse_dev(bits, bpos, 3, W, H)
    torch.cuda.synchronize()
    assert hip.parse_fallback_frames() == 1                   # (c) only
    assert torch.equal(n_ser, n_par)
    assert torch.equal(o_ser, o_par)
    # the same three frames through the ranges-of-GOPs pipeline (parse || reconstruct) and through the two separate calls
    dec_a = hip.decode_dev(bits, bpos, o_par, n_par, 3, W, H)
    dec_b, o_b, n_b = hip.parse_decode_dev(bits, bpos, 3, W, H)
    torch.cuda.synchronize()
    assert torch.equal(dec_a, dec_b) and torch.equal(o_b, o_par) and torch.equal(n_b, n_par)
    same_without_offsets(torch, hip, bits, bpos, 3, W, H, 0, None, None, dec_a, n_par)       # (c) gets its entry bits from the robust kernels
    assert hip.parse_fallback_frames() == 1


@pytest.mark.parametrize("geom", [(3840, 2160, 4), (4, 4, 9), (2052, 4, 5), (8, 1024, 6)])
@pytest.mark.parametrize("mode512", [True, False])
def test_parser_forms_agree_on_large_and_degenerate_frames(torch, hip, monkeypatch, geom, mode512):
    """speculative walks + proof against the map / stitch / emit kernels on a 4K frame (1 100 regions of 59 pieces per
    frame), one block, one block row and one block column: same entry offsets, every block entered, nothing given up"""
    W, H, T = geom
    p0, p1 = S.content_palettes([S.synth_frame(min(W, 640), min(H, 480), 0)])
    hip.set_palette(p0, p1, mode512)
    out, sizes = hip.encode_dev(hip.synth_dev(W, H, 0, T), T, W, H)
    hip.check()
    monkeypatch.delenv("AGMV_HIP_PARSE", raising=False)
    o1, n1 = hip.parse_dev(out, sizes, T, W, H)
    assert hip.parse_fallback_frames() == 0
    monkeypatch.setenv("AGMV_HIP_PARSE", "robust")
    o2, n2 = hip.parse_dev(out, sizes, T, W, H)
    torch.cuda.synchronize()
    nblk = W * H // 16
    assert n1.cpu().tolist() == [nblk] * T and torch.equal(n1, n2)
    assert torch.equal(o1, o2)


@pytest.mark.parametrize("first_fc", [0, 2])
def test_parse_decode_in_ranges(torch, hip, monkeypatch, first_fc):
    """agmv_hip_parse_decode_frames_dev cut into ranges of GOPs (parser on its own stream, reconstruction of a range behind
    the parse of that range) gives the pixels, offsets and counts of the two separate calls -- also when the batch starts
    inside a GOP and carries decoder state in."""
    W, H, T = 320, 240, 41
    p0, p1 = S.content_palettes([S.synth_frame(W, H, 0)])
    hip.set_palette(p0, p1, True)
    frames = hip.synth_dev(W, H, 0, T)
    out, sizes = hip.encode_dev(frames, T, W, H, first_fc, ientries=torch.zeros(W * H, dtype=torch.int16, device="cuda"))
    hip.check()
    sizes[7] = sizes[7] // 2                                   # a truncated frame: stale blocks, k_fixup has work
    prev = torch.randint(0, 1 << 24, (H, W), dtype=torch.int32, device="cuda")
    previ = torch.randint(0, 1 << 24, (H, W), dtype=torch.int32, device="cuda")
    offs, nent = hip.parse_dev(out, sizes, T, W, H)
    ref = hip.decode_dev(out, sizes, offs, nent, T, W, H, first_fc, prev=prev, prev_iframe=previ)
    torch.cuda.synchronize()
    same_without_offsets(torch, hip, out, sizes, T, W, H, first_fc, prev, previ, ref, nent)
    for ns in ("1", "3", "5", "32"):
        monkeypatch.setenv("AGMV_DEC_SLICES", ns)
        dec, o2, n2 = hip.parse_decode_dev(out, sizes, T, W, H, first_fc, prev=prev, prev_iframe=previ)
        torch.cuda.synchronize()
        assert torch.equal(n2, nent), ns
        assert torch.equal(dec, ref), ns
        idx = torch.arange(W * H // 16, device="cuda")[None, :] < nent[:, None]
        assert torch.equal(o2[idx], offs[idx]), ns



# ------------------------------------------------------------------------------- helpers
def test_synth_interp_histogram(torch, hip):
    for (W, H) in ((320, 240), (68, 36)):
        d = hip.synth_dev(W, H, 3, 5)
        torch.cuda.synchronize()
        got = to_u32(d)
        for i in range(5):
            assert (got[i] == S.synth_frame(W, H, 3 + i)).all()
    a = S.synth_frame(320, 240, 1).reshape(-1)
    b = np.random.default_rng(4).integers(0, 1 << 24, size=a.size, dtype=np.uint32)
    exp = np.zeros_like(a)
    O.oracle().orc_interp_frame(exp, a, b, a.size)
    got = to_u32(hip.interp_dev(dev_u32(torch, a), dev_u32(torch, b)))
    assert (got == exp).all()
    for q, shifts in ((1, (2, 2, 1, 13, 7)), (2, (3, 2, 2, 12, 6)), (3, (3, 2, 3, 11, 5))):
        r, g_, bb = (b >> 16) & 255, (b >> 8) & 255, b & 255
        code = ((r >> shifts[0]) << shifts[3]) | ((g_ >> shifts[1]) << shifts[4]) | (bb >> shifts[2])
        exp_h = np.bincount(code, minlength=1 << 19).astype(np.uint32)
        got_h = to_u32(hip.histogram_dev(dev_u32(torch, b), q))
        assert (got_h == exp_h).all()


# ------------------------------------------------------------------------------- full size
def test_full_size_properties(torch, hip):
    """BASELINE-sized frames (1080p), sizes the CPU oracle cannot cover in seconds: check
    size-independent properties instead -- determinism under re-scheduling (two runs identical),
    batch-split invariance (GOP-aligned halves == whole), and encode->decode round trip against
    the quantised image (every decoded pixel of a NORMAL block is palette[nearest(pixel)])."""
    W, H, T = 1920, 1080, 32
    frames = hip.synth_dev(W, H, 0, T)
    f_np = [S.synth_frame(W, H, t) for t in range(2)]
    p0, p1 = S.content_palettes(f_np)
    hip.set_palette(p0, p1, True)
    out1, sz1 = hip.encode_dev(frames, T, W, H)
    out2, sz2 = hip.encode_dev(frames, T, W, H)
    hip.check()
    assert torch.equal(sz1, sz2)
    sz = sz1.cpu().numpy()
    for t in range(T):
        assert torch.equal(out1[t, :sz[t]], out2[t, :sz[t]])
    h1, s1 = hip.encode_dev(frames[:16], 16, W, H)
    h2, s2 = hip.encode_dev(frames[16:], 16, W, H, first_frame_count=16)
    hip.check()
    assert torch.equal(torch.cat([s1, s2]), sz1)
    for t in range(16):
        assert torch.equal(h1[t, :sz[t]], out1[t, :sz[t]]) and torch.equal(h2[t, :sz[16 + t]], out1[16 + t, :sz[16 + t]])
    offs, nent = hip.parse_dev(out1, sz1, T, W, H)
    dec = hip.decode_dev(out1, sz1, offs, nent, T, W, H)
    same_without_offsets(torch, hip, out1, sz1, T, W, H, 0, None, None, dec, nent)
    torch.cuda.synchronize()
    assert (nent.cpu().numpy() == W * H // 16).all()
    pal = torch.from_numpy(np.concatenate([p0, p1]).view(np.int32)).cuda()
    ent = hip.quantise_dev(frames.reshape(-1)).to(torch.int64) & 0xFFFF
    quant = pal[ent].reshape(T, H, W)
    # I-frames: a block is either NORMAL (== quantised) or FILL (== quantised top-left pixel)
    for t in (0, 4, 28):
        d = dec[t].reshape(H // 4, 4, W // 4, 4)
        q = quant[t].reshape(H // 4, 4, W // 4, 4)
        normal = (d == q).all(dim=3).all(dim=1)
        fill = (d == q[:, :1, :, :1]).all(dim=3).all(dim=1)
        ok = normal | fill
        ok[-1, -1] = True        # last block: FILL takes its colour from the left neighbour (quirk)
        assert bool(ok.all()), "frame %d: %d blocks neither NORMAL nor FILL" % (t, int((~ok).sum()))
    # P-frames: additionally COPY (== decoded I-frame block)
    for t in (1, 7, 30):
        d = dec[t].reshape(H // 4, 4, W // 4, 4)
        q = quant[t].reshape(H // 4, 4, W // 4, 4)
        i = dec[t // 4 * 4].reshape(H // 4, 4, W // 4, 4)
        ok = (d == q).all(dim=3).all(dim=1) | (d == q[:, :1, :, :1]).all(dim=3).all(dim=1) | (d == i).all(dim=3).all(dim=1)
        ok[-1, -1] = True
        assert bool(ok.all()), "frame %d" % t
    # oracle spot check of two full frames of this very clip
    enc = O.OracleEncoder(W, H, True, p0, p1)
    for t in range(2):
        exp = enc.encode(f_np[t])
        got = out1[t, :sz[t]].cpu().numpy()
        assert len(got) == len(exp) and (got == exp).all()


@pytest.mark.parametrize("shape", [(1920, 1080), (1280, 720)])
def test_decode_full_size_vs_oracle(torch, hip, shape):
    """configs 3 and 5 at their real frame sizes: parse + reconstruct of two GOPs against the oracle decoder, pixel for pixel"""
    W, H = shape
    frames = np.stack([S.synth_frame(W, H, t) for t in range(8)])
    p0, p1 = S.content_palettes(frames[:4])
    hip.set_palette(p0, p1, True)
    bits = gpu_encode(torch, hip, frames)
    got, _, nent = gpu_decode(torch, hip, bits, None, W, H)
    assert (nent == W * H // 16).all()
    dec = O.OracleDecoder(W, H, True, p0, p1)
    for t in range(8):
        exp = dec.decode(bits[t])
        assert (got[t] == exp).all(), "frame %d: %d pixels differ" % (t, int((got[t] != exp).sum()))


def test_c3_full_batch_vs_oracle(torch, hip):
    """config 3 at its real batch size -- 1024 resident 1080p frames in ONE encode / parse / decode (254 tiles x 1024 status
    words, 260 096 tickets): the first, a middle and the last GOP against the oracle, byte for byte and pixel for pixel"""
    W, H, T = 1920, 1080, 1024
    free, _ = torch.cuda.mem_get_info()
    if free < 30 * (1 << 30):
        pytest.skip("needs ~25 GB of device memory")
    frames = hip.synth_dev(W, H, 0, T)
    # the palette bench.py encodes with: AGMV_HIGH_QUALITY histogram over the whole clip (GPU pass 1) + the reference's pick
    hist = hip.histogram_dev(frames.reshape(-1), 1).cpu().numpy().view(np.uint32)
    import hostlib as HL
    q0, q1 = np.zeros(256, np.uint64), np.zeros(256, np.uint64)
    HL.lib().AGMV_BuildPalette(hist, 1, 3, q0, q1)
    p0, p1 = q0.astype(np.uint32), q1.astype(np.uint32)
    hip.set_palette(p0, p1, True)
    out, sizes = hip.encode_dev(frames, T, W, H)
    hip.check()
    offs, nent = hip.parse_dev(out, sizes, T, W, H)
    dec = hip.decode_dev(out, sizes, offs, nent, T, W, H)
    torch.cuda.synchronize()
    assert hip.decode_depends_on_prior(W, H) is False
    del offs
    dec2 = hip.decode_bitstreams_dev(out, sizes, T, W, H)     # the form bench.py times: no offsets[]
    torch.cuda.synchronize()
    assert torch.equal(dec, dec2) and hip.parse_fallback_frames() == 0 and hip.decode_depends_on_prior(W, H) is False
    del dec2
    sz = sizes.cpu().numpy()
    assert (nent.cpu().numpy() == W * H // 16).all() and (sz > 0).all() and (sz < hip.max_usize(W, H)).all()
    for g0 in (0, 508, 1020):
        enc = O.OracleEncoder(W, H, True, p0, p1, first_frame_count=g0)
        odec = O.OracleDecoder(W, H, True, p0, p1)
        odec.s.frame_count = g0
        for t in range(g0, g0 + 4):
            exp = enc.encode(S.synth_frame(W, H, t))
            got = out[t, :sz[t]].cpu().numpy()
            assert len(got) == len(exp) and (got == exp).all(), "frame %d first diff %s" % (t, first_diff(got, exp))
            pix = odec.decode(exp)
            assert (to_u32(dec[t]).reshape(-1) == pix).all(), "decoded frame %d" % t


def test_decode_heavily_truncated_1080p(torch, hip):
    """thousands of stale blocks per frame: every frame of a 64-frame 1080p clip is cut at about half its length, so the lower
    half of each picture keeps earlier pixels across GOPs (reference src/agmv_decode.c:229-232) -- the repair pass (k_fixup,
    one wave per 64 block positions) against the oracle decoder, and a clip whose only damaged frame is the very last block"""
    W, H, T = 1920, 1080, 64
    frames = hip.synth_dev(W, H, 0, T)
    p0, p1 = S.content_palettes([S.synth_frame(W, H, t) for t in range(2)])
    hip.set_palette(p0, p1, True)
    out, sizes = hip.encode_dev(frames, T, W, H)
    hip.check()
    sz = sizes.cpu().numpy()
    full = [out[t, :sz[t]].cpu().numpy() for t in range(T)]
    rng = np.random.default_rng(4)
    cut = [b[:int(len(b) * (0.35 + 0.3 * rng.random()))] if t else b for t, b in enumerate(full)]   # frame 0 whole: a known picture underneath
    dec = O.OracleDecoder(W, H, True, p0, p1)
    exp, pads = [], []
    for b in cut:
        pix, padded, _, _ = dec.decode(b, want_tables=True)
        exp.append(pix)
        pads.append(padded[len(b):len(b) + 16])
    hip.enable_timing(True)
    got, _, nent = gpu_decode(torch, hip, cut, pads, W, H)
    ms = hip.last_kernel_ms(2)
    hip.enable_timing(False)
    assert (nent[1:] < W * H // 16).all() and hip.decode_depends_on_prior(W, H) is False    # stale, but nothing from before the batch
    for t in range(T):
        assert (got[t] == exp[t]).all(), "frame %d: %d pixels differ" % (t, int((got[t] != exp[t]).sum()))
    print("k_decode + k_fixup, 64 x 1080p, every frame truncated: %.3f ms" % ms)
    # only the last block is damaged (its FILL is cut off), in a P-frame: the repair involves the left neighbour only
    one = list(full[:8])
    nblk = W * H // 16
    one[5] = np.concatenate([np.tile(np.array([0x4E, 3], np.uint8), nblk - 1), np.array([0x4E], np.uint8)])
    dec = O.OracleDecoder(W, H, True, p0, p1)
    exp1 = [dec.decode(b) for b in one]
    got1, _, _ = gpu_decode(torch, hip, one, None, W, H)
    for t in range(8):
        assert (got1[t] == exp1[t]).all(), "frame %d" % t


@pytest.mark.parametrize("kind", ["clean", "escape", "copy", "overrun", "fill_cut", "garbage_copy"])
def test_gop_range_decode_and_dependency_predicate(torch, hip, kind):
    """what libagmv_amd.shard.decode_sharded relies on, with the real parser and k_decode: a GOP range decoded on its
    own (fresh decoder state) equals the serial decode unless shard.range_depends_on_prior_state says it depends on
    earlier frames -- and then it does once the hand-off state (last frame, I-frame snapshot) is supplied."""
    from libagmv_amd import shard
    W, H, T, cut = 64, 48, 16, 8
    nblk = W * H // 16
    frames = np.stack([S.synth_frame(W, H, t) for t in range(T)])
    p0, p1 = S.content_palettes(frames[:4])
    hip.set_palette(p0, p1, True)
    bits = [b.copy() for b in gpu_encode(torch, hip, frames)]
    if kind == "escape":
        for f in (7, 8, 11):
            bits[f] = bits[f][:len(bits[f]) * 5 // 8]
    elif kind == "copy":
        bits[8] = np.full(nblk, 0x5E, np.uint8)
    elif kind == "overrun":                                    # the range's first frame: FILL blocks, then a last NORMAL block the stream ends inside
        bits[8] = np.concatenate([np.tile(np.array([0x4E, 7], np.uint8), nblk - 1), np.array([0x2F, 1, 2, 3, 4, 5], np.uint8)])
    elif kind == "fill_cut":                                   # every block entered, but the last FILL's index byte lies behind bpos:
        bits[8] = np.tile(np.array([0x4E, 7], np.uint8), nblk)[:-1]   # the reference does not store it (src/agmv_decode.c:268-271)
    elif kind == "garbage_copy":                               # block 0 is entered on a non-flag byte and slides to a COPY flag (:236-243)
        bits[8] = np.concatenate([np.array([0x00, 0x5E], np.uint8), np.tile(np.array([0x4E, 9], np.uint8), nblk - 1)])
    dec = O.OracleDecoder(W, H, True, p0, p1)
    exp, pads = [], []
    for b in bits:
        pix, padded, _, _ = dec.decode(b, want_tables=True)
        exp.append(pix)
        pads.append(padded[len(b):len(b) + 16])
    # second range on its own
    stride = (max(len(b) for b in bits) + 16 + 255) & ~255
    slab = np.zeros((T - cut, stride), np.uint8)
    for i, f in enumerate(range(cut, T)):
        slab[i, :len(bits[f])] = bits[f]
        slab[i, len(bits[f]):len(bits[f]) + 16] = pads[f]
    alone, offs, nent = gpu_decode(torch, hip, bits[cut:], pads[cut:], W, H, first_fc=cut)
    dep = shard.range_depends_on_prior_state(torch.from_numpy(slab), torch.tensor([len(b) for b in bits[cut:]], dtype=torch.int32),
                                             torch.from_numpy(offs.astype(np.int64)), torch.from_numpy(nent.astype(np.int32)), nblk, True)
    same = all((alone[i] == exp[cut + i]).all() for i in range(T - cut))
    assert dep == (kind != "clean")
    # the decoder's own account of the same fact (what a sharded GPU decode uses): the last decode on `hip` was `alone`
    assert hip.decode_depends_on_prior(W, H) == dep
    if not dep:
        assert same
    else:
        assert not same, "the damaged clip was meant to make the range depend on earlier frames"
        first, _, _ = gpu_decode(torch, hip, bits[:cut], pads[:cut], W, H)
        fixed, _, _ = gpu_decode(torch, hip, bits[cut:], pads[cut:], W, H, first_fc=cut, prev=first[cut - 1], prev_iframe=first[cut - 4])
        for i in range(T - cut):
            assert (fixed[i] == exp[cut + i]).all(), "frame %d after the hand-off" % (cut + i)


def test_encode_batch_reads_and_writes_entry_plane(torch, hip):
    """a batch that starts inside a GOP reads the caller's I-frame entry plane while tiles of its own next I-frame
    produce the new one (fuzz seed 361 caught the two meeting in one buffer): P-frames that equal the NEXT I-frame turn
    into COPY blocks if they see its entries.  Repeated, since the overlap is a matter of timing."""
    W, H = 320, 72
    rng = np.random.default_rng(5)
    blocks = lambda: np.repeat(np.repeat(rng.integers(0, 1 << 24, size=(H // 8, W // 8), dtype=np.uint32), 8, 0), 8, 1)
    a, b = S.synth_frame(W, H, 3), blocks()
    frames = np.stack([a, blocks(), b, b, b, b, b, b, b])     # frames 2..3 equal the I-frames 4 and 8 that follow them
    p0, p1 = S.content_palettes(frames[:4])
    for mode512 in (False, True):
        hip.set_palette(p0, p1, mode512)
        whole = gpu_encode(torch, hip, frames)
        for rep in range(25):
            ient = torch.zeros(W * H, dtype=torch.int16, device="cuda")
            first = gpu_encode(torch, hip, frames[:1], first_fc=0, ientries=ient)
            rest = gpu_encode(torch, hip, frames[1:], first_fc=1, ientries=ient)
            for t, got in enumerate(first + rest):
                assert len(got) == len(whole[t]) and (got == whole[t]).all(), "rep %d frame %d mode512=%s" % (rep, t, mode512)


def test_pack_unpack_frames(torch, hip):
    """agmv_hip_pack_frames_dev / agmv_hip_unpack_frames_dev (the message of the GOP-sharded encoder's final gather): ragged
    sizes with empty frames, every alignment of the message offsets, rows longer than one chunk; against plain slicing"""
    rng = np.random.default_rng(11)
    for n, stride, big in ((1, 256, 0), (7, 512, 0), (301, 1280, 0), (40, 70400, 60000)):
        slab = torch.from_numpy(rng.integers(0, 256, (n, stride), dtype=np.uint8)).cuda()
        sz = rng.integers(0, min(stride, 700) + 1, n)
        sz[rng.integers(0, n, max(1, n // 5))] = 0
        if big:
            sz[::3] = rng.integers(big - 5, big + 5, len(sz[::3]))
        sizes = torch.from_numpy(sz.astype(np.int32)).cuda()
        packed, offs = hip.pack_frames_dev(slab, sizes)
        exp = torch.cat([slab[f, :int(sz[f])] for f in range(n)]) if sz.sum() else slab.new_empty(0)
        assert torch.equal(packed, exp)
        assert offs.cpu().tolist() == [0] + np.cumsum(sz).tolist()
        back = hip.unpack_frames_dev(packed, sizes, stride)
        keep = torch.arange(stride, device="cuda")[None, :] < sizes[:, None]
        assert torch.equal(back, slab * keep)
