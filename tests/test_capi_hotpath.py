"""The exported hot-path helpers and the playback helpers of libagmv_amd/libagmv.so against the compiled
reference (oracle/_ref/libagmv_ref.so), called the way a C consumer calls them:

  AGMV_FindNearestColor / AGMV_FindNearestEntry            reference src/agmv_utils.c:785-816, :851-895
  AGMV_CompareIFrameBlock / AGMV_ComparePFrameBlock        src/agmv_encode.c:302-352, :240-300
  AGMV_AssembleIFrameBitstream / AGMV_AssemblePFrameBitstream   src/agmv_encode.c:354-436, :438-527
  AGMV_ParseAGMV / ResetVideo / SkipForwards / SkipBackwards / SkipTo / PlayAGMV    src/agmv_playback.c:18-115

The palettes used here contain DUPLICATE colours and the entry planes use the later duplicates: an
implementation that turns entries into colours and quantises them again returns the first duplicate.
Both libraries have the same AGMV layout (tests/test_abi.py), so the shim's field accessors read either
library's objects.  The seek / parse helpers do no arithmetic on pixels and run without a GPU; everything
that decodes or compares on the GPU is marked gpu."""
import ctypes as C
import os

import numpy as np
import pytest

import hostlib as H
import oracles as O

needs_ref = pytest.mark.skipif(not O.have_ref(), reason="oracle/_ref not built (needs /root/reference at build time)")
vp = C.c_void_p
SPLASH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "agmv_splash.agmv")


class Entry(C.Structure):                       # AGMV_ENTRY, include/agmv.h (reference include/agmv_defines.h:122-126)
    _fields_ = [("pal_num", C.c_ubyte), ("index", C.c_ubyte), ("occurence", C.c_ulong)]


def _libc():
    libc = C.CDLL(None)
    libc.fopen.restype = vp
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [vp]
    libc.ftell.restype = C.c_long
    libc.ftell.argtypes = [vp]
    libc.calloc.restype = vp
    libc.calloc.argtypes = [C.c_size_t, C.c_size_t]
    return libc


def _bind_playback(L):
    L.CreateAGMV.restype = vp
    L.CreateAGMV.argtypes = [C.c_ulong] * 4
    L.DestroyAGMV.argtypes = [vp]
    L.AGMV_DecodeHeader.restype = C.c_int
    L.AGMV_DecodeHeader.argtypes = [vp, vp]
    for f in ("AGMV_ParseAGMV", "AGMV_ResetVideo", "AGMV_PlayAGMV"):
        getattr(L, f).restype = None
        getattr(L, f).argtypes = [vp, vp]
    for f in ("AGMV_SkipForwards", "AGMV_SkipBackwards", "AGMV_SkipTo", "AGMV_SkipForwardsAndDecodeAudio"):
        getattr(L, f).restype = None
        getattr(L, f).argtypes = [vp, vp, C.c_int]
    L.AGMV_IsVideoDone.restype = C.c_int
    L.AGMV_IsVideoDone.argtypes = [vp]
    return L


def _ref_accessors():
    R = O.ref()
    R.refshim_get_frame_count.restype = C.c_uint32
    R.refshim_get_frame_count.argtypes = [vp]
    R.refshim_offset_table.restype = C.c_uint32
    R.refshim_offset_table.argtypes = [vp, C.c_uint32]
    R.refshim_frame_pixels.argtypes = [vp, O.u32p]
    return R


class Player:
    """one library's view of the splash file: an AGMV object + an open FILE*, driven step by step"""

    def __init__(self, L, R, libc):
        self.L, self.R, self.libc = _bind_playback(L), R, libc
        self.a = self.L.CreateAGMV(119, 320, 240, 12)
        self.f = libc.fopen(SPLASH.encode(), b"rb")
        assert self.L.AGMV_DecodeHeader(self.f, self.a) == 0
        # the file carries audio: give the object the sample buffer a player allocates (reference src/agmv_decode.c:575), the
        # reference's chunk walk writes the decoded samples there; DestroyAGMV frees it
        self.L.AGMV_GetAudioSize.restype = C.c_ulong
        self.L.AGMV_GetAudioSize.argtypes = [vp]
        track = C.c_void_p.from_address(self.a + 4216).value                  # agmv->audio_track
        C.c_void_p.from_address(track + 16).value = libc.calloc(2 * self.L.AGMV_GetAudioSize(self.a) + 64, 2)   # ->pcm (room for the chunks the script decodes twice)
        C.c_void_p.from_address(track + 24).value = None                      # ->pcm8
        C.c_ulong.from_address(track + 8).value = 0                           # ->start_point
        # the reference's CreateAGMV mallocs img_data / iframe / the bitstream buffer without clearing them (src/agmv_utils.c:332-369);
        # blocks a frame does not reach keep what is there, so in a process whose heap is no longer fresh the reference
        # would start from garbage.  Both objects start from zeroes (ours callocs; SURVEY 8c freezes the UB to zero pages).
        npx = 320 * 240
        for off in (4200, 4208):                                              # agmv->frame, agmv->iframe
            fr = C.c_void_p.from_address(self.a + off).value
            C.memset(C.c_void_p.from_address(fr + 16).value, 0, npx * C.sizeof(C.c_ulong))      # ->img_data
        bs = C.c_void_p.from_address(self.a + 4192).value                     # agmv->bitstream
        C.memset(C.c_void_p.from_address(bs).value, 0, C.c_ulong.from_address(bs + 8).value)    # ->data[0 .. len)
        chunk = C.c_void_p.from_address(self.a + 4184).value                  # agmv->audio_chunk: the reference's CreateAGMV leaves
        C.c_void_p.from_address(chunk + 16).value = None                      # its pointers uninitialised and DestroyAGMV frees them
        C.c_void_p.from_address(chunk + 24).value = None

    def state(self):
        return (int(self.libc.ftell(self.f)), int(self.R.refshim_get_frame_count(self.a)))

    def table(self, n):
        return [int(self.R.refshim_offset_table(self.a, i)) for i in range(n)]

    def pixels(self):
        out = np.zeros(320 * 240, np.uint32)
        self.R.refshim_frame_pixels(self.a, out)
        return out

    def close(self):
        self.libc.fclose(self.f)
        self.L.DestroyAGMV(self.a)


@needs_ref
def test_parse_and_seek_helpers_match_reference():
    """no decode involved: chunk walk, offset_table bookkeeping and the seek arithmetic (also the version-1-only reset)"""
    libc, R = _libc(), _ref_accessors()
    ours, ref = Player(H.lib(), R, libc), Player(O.ref(), R, libc)
    steps = [("AGMV_ParseAGMV", None), ("AGMV_ResetVideo", None), ("AGMV_SkipForwards", 1), ("AGMV_SkipForwards", 6),
             ("AGMV_SkipBackwards", 3), ("AGMV_SkipForwards", 4), ("AGMV_SkipTo", 37), ("AGMV_SkipBackwards", 8),
             ("AGMV_SkipTo", 118), ("AGMV_SkipTo", 400), ("AGMV_SkipForwardsAndDecodeAudio", 2), ("AGMV_SkipBackwards", 1),
             ("AGMV_ResetVideo", None), ("AGMV_SkipForwards", 0), ("AGMV_SkipForwards", 9)]
    for name, n in steps:
        for p in (ours, ref):
            fn = getattr(p.L, name)
            fn(p.f, p.a) if n is None else fn(p.f, p.a, n)
        assert ours.state() == ref.state(), (name, n, ours.state(), ref.state())
        assert ours.L.AGMV_IsVideoDone(ours.a) == ref.L.AGMV_IsVideoDone(ref.a)
    assert ours.table(119) == ref.table(119)
    # a fresh object that is never parsed: the skips fill offset_table as they go (reference src/agmv_playback.c:35-60)
    ours.close(); ref.close()
    ours, ref = Player(H.lib(), R, libc), Player(O.ref(), R, libc)
    for n in (2, 5, 1):
        for p in (ours, ref):
            p.L.AGMV_SkipForwards(p.f, p.a, n)
        assert ours.state() == ref.state()
    fc = ours.state()[1]
    assert ours.table(fc) == ref.table(fc) and fc >= 8
    for p in (ours, ref):
        p.L.AGMV_SkipBackwards(p.f, p.a, 4)
    assert ours.state() == ref.state()
    ours.close(); ref.close()


@needs_ref
@pytest.mark.gpu
def test_play_and_seek_like_a_player():
    """AGMV_PlayAGMV + skips in a player loop WITHOUT AGMV_ParseAGMV (tools/agmvp's pattern): file position, frame_count,
    offset_table and the decoded frame after every step equal the reference's"""
    libc, R = _libc(), _ref_accessors()
    ours, ref = Player(H.lib(), R, libc), Player(O.ref(), R, libc)
    script = [("play", 6), ("fwd", 3), ("play", 3), ("back", 4), ("play", 5), ("fwd", 1), ("play", 2), ("reset", 0), ("play", 2)]
    for what, n in script:
        for k in range(n if what == "play" else 1):
            for p in (ours, ref):
                if what == "play":
                    p.L.AGMV_PlayAGMV(p.f, p.a)
                elif what == "fwd":
                    p.L.AGMV_SkipForwards(p.f, p.a, n)
                elif what == "back":
                    p.L.AGMV_SkipBackwards(p.f, p.a, n)
                else:
                    p.L.AGMV_ResetVideo(p.f, p.a)
            assert ours.state() == ref.state(), (what, n, k)
            if what == "play":
                assert (ours.pixels() == ref.pixels()).all(), (what, k, ours.state())
    fc = max(ours.state()[1], 16)
    assert ours.table(fc) == ref.table(fc)
    ours.close(); ref.close()


def _dup_palettes(seed=5):
    """512 colours with many duplicates: zeros in unused slots (like palettes the reference builds), repeated colours
    inside a palette and the same colours in both palettes"""
    rng = np.random.default_rng(seed)
    base = rng.integers(0, 1 << 24, 96, dtype=np.uint32)
    p0 = base[rng.integers(0, 96, 256)].astype(np.uint32)
    p1 = base[rng.integers(0, 96, 256)].astype(np.uint32)
    p0[100:130] = 0
    p1[200:256] = 0
    p0[126] = 0
    return p0, p1


@needs_ref
@pytest.mark.gpu
def test_exported_find_nearest_matches_reference():
    L = H.lib()
    L.AGMV_FindNearestColor.restype = C.c_ubyte
    L.AGMV_FindNearestColor.argtypes = [H.u64p, C.c_ulong]
    L.AGMV_FindNearestEntry.restype = Entry
    L.AGMV_FindNearestEntry.argtypes = [H.u64p, H.u64p, C.c_ulong]
    rng = np.random.default_rng(11)
    for seed in (5, 6):
        p0, p1 = _dup_palettes(seed)
        pix = np.concatenate([rng.integers(0, 1 << 24, 300, dtype=np.uint32), p0[:20], p1[-20:], p0[40:44] ^ 1,
                              np.array([0, 0xFFFFFF, 0xAB123456], np.uint32)]).astype(np.uint32)
        exp512 = np.zeros(pix.size, np.uint16)
        exp256 = np.zeros(pix.size, np.uint16)
        O.ref().refshim_nearest_entries(p0, p1, 1, pix, pix.size, exp512)
        O.ref().refshim_nearest_entries(p0, p1, 0, pix, pix.size, exp256)
        P0, P1 = p0.astype(np.uint64), p1.astype(np.uint64)
        for k, c in enumerate(pix):                        # alternating palettes / modes call after call, like a foreign caller may
            e = L.AGMV_FindNearestEntry(P0, P1, int(c))
            assert (e.pal_num << 8 | e.index) == exp512[k], (seed, k, hex(int(c)))
            assert L.AGMV_FindNearestColor(P0, int(c)) == exp256[k], (seed, k)


def _entry_plane(rng, W, Hh, p0, p1):
    """an entry plane that prefers the LATER of duplicate colours, with flat, near-flat and noisy blocks"""
    ent = rng.integers(0, 512, (Hh, W)).astype(np.uint16)
    for by in range(0, Hh, 4):
        for bx in range(0, W, 4):
            kind = rng.integers(0, 4)
            e0 = int(rng.integers(0, 512))
            if kind == 0:
                ent[by:by + 4, bx:bx + 4] = e0
            elif kind == 1:                                 # same COLOUR through different entries (duplicates)
                pal = np.concatenate([p0, p1])
                same = np.nonzero(pal == pal[e0])[0]
                ent[by:by + 4, bx:bx + 4] = same[rng.integers(0, same.size, (4, 4))]
    return ent


def _bind_assemble(L):
    L.CreateAGMV.restype = vp
    L.CreateAGMV.argtypes = [C.c_ulong] * 4
    L.DestroyAGMV.argtypes = [vp]
    L.AGMV_SetOPT.argtypes = [vp, C.c_int]
    L.AGMV_SetICP0.argtypes = [vp, H.u64p]
    L.AGMV_SetICP1.argtypes = [vp, H.u64p]
    for f in ("AGMV_AssembleIFrameBitstream", "AGMV_AssemblePFrameBitstream"):
        getattr(L, f).restype = None
        getattr(L, f).argtypes = [vp, C.POINTER(Entry)]
    L.AGMV_CompareIFrameBlock.restype = C.c_ubyte
    L.AGMV_CompareIFrameBlock.argtypes = [vp, C.c_ulong, C.c_ulong, C.c_ulong, C.POINTER(Entry)]
    L.AGMV_ComparePFrameBlock.restype = C.c_ubyte
    L.AGMV_ComparePFrameBlock.argtypes = [vp, C.c_ulong, C.c_ulong, C.POINTER(Entry)]


def _entries_c(ent):
    arr = (Entry * ent.size)()
    flat = ent.reshape(-1)
    for i in range(ent.size):
        arr[i].pal_num = int(flat[i]) >> 8
        arr[i].index = int(flat[i]) & 0xFF
    return arr


@needs_ref
@pytest.mark.gpu
@pytest.mark.parametrize("opt", [O.OPT_III, O.OPT_II])
def test_exported_assemble_and_compare_match_reference(opt):
    """the reference's own AGMV_Assemble{I,P}FrameBitstream / AGMV_Compare{I,P}FrameBlock on the same AGMV_ENTRY planes"""
    L, R = H.lib(), _ref_accessors()
    _bind_assemble(L)
    R.refshim_assemble.restype = C.c_size_t
    R.refshim_assemble.argtypes = [vp, O.u16p, C.c_int, O.u8p]
    R.refshim_set_iframe_entries.argtypes = [vp, O.u16p]
    R.refshim_compare_i.restype = C.c_uint
    R.refshim_compare_i.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_uint32, O.u16p]
    R.refshim_compare_p.restype = C.c_uint
    R.refshim_compare_p.argtypes = [vp, C.c_uint32, C.c_uint32, O.u16p]
    W, Hh = 64, 48
    rng = np.random.default_rng(3 + opt)
    p0, p1 = _dup_palettes(9)
    ient = _entry_plane(rng, W, Hh, p0, p1)
    ent = ient.copy()
    chg = rng.random((Hh // 4, W // 4)) < 0.5                 # half of the blocks differ from the I-frame's
    new = _entry_plane(rng, W, Hh, p0, p1)
    for by in range(Hh // 4):
        for bx in range(W // 4):
            if chg[by, bx]:
                ent[by * 4:by * 4 + 4, bx * 4:bx * 4 + 4] = new[by * 4:by * 4 + 4, bx * 4:bx * 4 + 4]
    if opt == O.OPT_II:
        ient &= 0xFF
        ent &= 0xFF
    ra = R.refshim_create(W, Hh, opt, O.LZSS, p0, p1)
    R.refshim_set_iframe_entries(ra, np.ascontiguousarray(ient.reshape(-1)))
    a = L.CreateAGMV(1, W, Hh, 24)
    L.AGMV_SetOPT(a, opt)
    L.AGMV_SetICP0(a, p0.astype(np.uint64))
    L.AGMV_SetICP1(a, p1.astype(np.uint64))
    # our object's iframe_entries: assemble the I plane first (AGMV_EncodeFrame does the same before P-frames, :626-630)
    ie_c, e_c = _entries_c(ient), _entries_c(ent)
    buf = np.zeros(W * Hh * 3 + 64, np.uint8)
    # I-frame assembly of the I plane
    n_ref = R.refshim_assemble(ra, np.ascontiguousarray(ient.reshape(-1)), 1, buf)
    exp_i = buf[:n_ref].copy()
    got_i = _assemble(L, a, ie_c, True)
    assert got_i.size == exp_i.size and (got_i == exp_i).all(), "AGMV_AssembleIFrameBitstream differs from the reference"
    # P-frame assembly against that I plane: install the I-frame entries in our object, then assemble
    _set_iframe_entries(a, ient)
    n_ref = R.refshim_assemble(ra, np.ascontiguousarray(ent.reshape(-1)), 0, buf)
    exp_p = buf[:n_ref].copy()
    got_p = _assemble(L, a, e_c, False)
    assert got_p.size == exp_p.size and (got_p == exp_p).all(), "AGMV_AssemblePFrameBitstream differs from the reference"
    assert (exp_p == 0x5E).any() and (exp_p == 0x2F).any()   # COPY and NORMAL blocks are present
    # the block predicates on a sample of blocks
    pal = np.concatenate([p0, p1])
    for k in range(24):
        x, y = int(rng.integers(0, W // 4)) * 4, int(rng.integers(0, Hh // 4)) * 4
        color = int(pal[int(ent[y, x])]) if k % 3 else int(rng.integers(0, 1 << 24))
        assert L.AGMV_CompareIFrameBlock(a, x, y, color, e_c) == R.refshim_compare_i(ra, x, y, color, np.ascontiguousarray(ent.reshape(-1)))
        assert L.AGMV_ComparePFrameBlock(a, x, y, e_c) == R.refshim_compare_p(ra, x, y, np.ascontiguousarray(ent.reshape(-1)))
    R.refshim_destroy(ra)
    L.DestroyAGMV(a)


def _agmv_field(a, name):
    """addresses of AGMV fields through the layout the ABI test pins (include/agmv.h == reference include/agmv_defines.h:140-162)"""
    R = O.ref()
    R.refshim_offsetof_frame_count.restype = C.c_size_t
    off_fc = R.refshim_offsetof_frame_count()
    # header | frame_chunk* | audio_chunk* | bitstream* | frame* | iframe* | audio_track* | iframe_entries* | ... | frame_count
    base = 4200 - 3 * 8                                       # &frame_chunk (agmv->frame sits at 4200, tests/test_gpu_files.py)
    order = ["frame_chunk", "audio_chunk", "bitstream", "frame", "iframe", "audio_track", "iframe_entries"]
    assert base + 8 * len(order) <= off_fc
    return a + base + 8 * order.index(name)


def _assemble(L, a, entries_c, iframe):
    bs = C.c_void_p.from_address(_agmv_field(a, "bitstream")).value      # AGMV_BITSTREAM { u8* data; u32 len; u32 pos; }
    C.c_ulong.from_address(bs + 16).value = 0                            # pos = 0
    (L.AGMV_AssembleIFrameBitstream if iframe else L.AGMV_AssemblePFrameBitstream)(a, entries_c)
    pos = C.c_ulong.from_address(bs + 16).value
    data = C.c_void_p.from_address(bs).value
    return np.ctypeslib.as_array(C.cast(data, C.POINTER(C.c_uint8)), (pos,)).copy()


def _set_iframe_entries(a, ient):
    p = C.c_void_p.from_address(_agmv_field(a, "iframe_entries")).value
    arr = C.cast(p, C.POINTER(Entry))
    flat = ient.reshape(-1)
    for i in range(flat.size):
        arr[i].pal_num = int(flat[i]) >> 8
        arr[i].index = int(flat[i]) & 0xFF
