"""ctypes bindings for the two CHECKERS (test infrastructure only):

* ``oracle/liboracle.so``        -- our plain-C restatement (always available after
  ``__graft_entry__.build()`` / ``make -C oracle``)
* ``oracle/_ref/libagmv_ref.so`` -- the unmodified reference compiled from
  /root/reference (present where it was built; travels to the GPU box as a built
  artefact; tests that need it skip when it is absent)

Nothing in ``libagmv_amd`` imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libagmv_ref.so")

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u16p = np.ctypeslib.ndpointer(np.uint16, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")


def build_oracles():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all"], check=True,
                   stdout=subprocess.DEVNULL)


_oracle = None
_ref = None


class _FileInfo(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "num_frames", "w", "h", "fps", "version", "fmt", "total_audio_duration", "sample_rate",
        "audio_size", "channels", "bits_per_sample")] + [("first_chunk", C.c_size_t)]


def oracle():
    global _oracle
    if _oracle is None:
        if not os.path.exists(ORACLE_SO):
            build_oracles()
        L = C.CDLL(ORACLE_SO)
        L.orc_find_nearest_color.restype = C.c_uint8
        L.orc_find_nearest_color.argtypes = [u32p, C.c_uint32]
        L.orc_find_nearest_entry.restype = C.c_uint16
        L.orc_find_nearest_entry.argtypes = [u32p, u32p, C.c_uint32]
        L.orc_quantise.restype = None
        L.orc_quantise.argtypes = [u32p, u32p, C.c_int, u32p, C.c_size_t, u16p]
        L.orc_encoder_new.restype = C.c_void_p
        L.orc_encoder_new.argtypes = [C.c_uint32, C.c_uint32, C.c_int, u32p, u32p, C.c_uint32]
        L.orc_encoder_free.argtypes = [C.c_void_p]
        L.orc_encode_frame.restype = C.c_size_t
        L.orc_encode_frame.argtypes = [C.c_void_p, u32p, u8p, C.c_void_p]
        for f in (L.orc_lzss_compress, L.orc_lz77_compress):
            f.restype = C.c_size_t
            f.argtypes = [u8p, C.c_size_t, u8p, C.POINTER(C.c_uint32)]
        L.orc_decoder_new.restype = C.c_void_p
        L.orc_decoder_new.argtypes = [C.c_uint32, C.c_uint32, C.c_int, u32p, u32p]
        L.orc_decoder_free.argtypes = [C.c_void_p]
        L.orc_decoder_lz.restype = C.c_size_t
        L.orc_decoder_lz.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint32]
        L.orc_decoder_set_bitstream.restype = None
        L.orc_decoder_set_bitstream.argtypes = [C.c_void_p, u8p, C.c_uint32]
        L.orc_decoder_parse.restype = None
        L.orc_decoder_parse.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32)]
        L.orc_parse_header.restype = C.c_int
        L.orc_parse_header.argtypes = [u8p, C.c_size_t, C.POINTER(_FileInfo), u32p, u32p]
        L.orc_find_next_frame_chunk.restype = C.c_size_t
        L.orc_find_next_frame_chunk.argtypes = [u8p, C.c_size_t, C.c_size_t]
        L.orc_fnv1a64.restype = C.c_uint64
        L.orc_fnv1a64.argtypes = [C.c_void_p, C.c_size_t, C.c_uint64]
        L.orc_interp_frame.restype = None
        L.orc_interp_frame.argtypes = [u32p, u32p, u32p, C.c_size_t]
        _oracle = L
    return _oracle


def have_ref():
    return os.path.exists(REF_SO)


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.refshim_create.restype = C.c_void_p
        L.refshim_create.argtypes = [C.c_uint32, C.c_uint32, C.c_int, C.c_int, u32p, u32p]
        L.refshim_destroy.argtypes = [C.c_void_p]
        L.refshim_set_frame_count.argtypes = [C.c_void_p, C.c_uint32]
        L.refshim_nearest_entries.restype = None
        L.refshim_nearest_entries.argtypes = [u32p, u32p, C.c_int, u32p, C.c_size_t, u16p]
        L.refshim_encode_frame_hot.restype = C.c_size_t
        L.refshim_encode_frame_hot.argtypes = [C.c_void_p, u32p, u8p, C.c_void_p]
        L.refshim_quantise_only.argtypes = [C.c_void_p, u32p, C.c_size_t]
        L.refshim_encode_frame_file.argtypes = [C.c_void_p, C.c_char_p, u32p, C.c_int]
        L.refshim_write_header.argtypes = [C.c_void_p, C.c_char_p]
        L.refshim_lz.restype = C.c_size_t
        L.refshim_lz.argtypes = [u8p, C.c_size_t, C.c_int, u8p, C.c_size_t, C.POINTER(C.c_uint32)]
        L.refshim_decoder_open.restype = C.c_void_p
        L.refshim_decoder_open.argtypes = [C.c_char_p, C.POINTER(C.c_int)] + [C.POINTER(C.c_uint32)] * 4
        L.refshim_decoder_palettes.argtypes = [C.c_void_p, u32p, u32p]
        L.refshim_decoder_next.restype = C.c_int
        L.refshim_decoder_next.argtypes = [C.c_void_p, C.c_void_p] + [C.POINTER(C.c_uint32)] * 3
        L.refshim_decoder_bitstream.argtypes = [C.c_void_p, u8p, C.c_size_t]
        L.refshim_decoder_close.argtypes = [C.c_void_p]
        L.refshim_interp.argtypes = [u32p, u32p, u32p, C.c_uint32, C.c_uint32]
        _ref = L
    return _ref


# ---------------------------------------------------------------- convenience wrappers

OPT_III, OPT_II = 3, 2
LZSS, LZ77 = 1, 2


def max_usize(w, h):
    return 33 * (w * h // 16) + 64


class OracleEncoder:
    """stateful E4-E9 (restatement)."""

    def __init__(self, w, h, mode512, p0, p1, first_frame_count=0):
        self.L = oracle()
        self.w, self.h = w, h
        self.p = self.L.orc_encoder_new(w, h, int(mode512), np.ascontiguousarray(p0, np.uint32),
                                        np.ascontiguousarray(p1, np.uint32), first_frame_count)
        self.buf = np.zeros(max_usize(w, h), np.uint8)

    def encode(self, pix, want_entries=False):
        pix = np.ascontiguousarray(pix, np.uint32).reshape(-1)
        ent = np.zeros(self.w * self.h, np.uint16) if want_entries else None
        n = self.L.orc_encode_frame(self.p, pix, self.buf,
                                    ent.ctypes.data_as(C.c_void_p) if want_entries else None)
        out = self.buf[:n].copy()
        return (out, ent) if want_entries else out

    def close(self):
        if self.p:
            self.L.orc_encoder_free(self.p)
            self.p = None

    __del__ = close


class RefEncoder:
    """loops A+B of the compiled reference (refshim_encode_frame_hot)."""

    def __init__(self, w, h, mode512, p0, p1, first_frame_count=0):
        self.L = ref()
        self.w, self.h = w, h
        self.p = self.L.refshim_create(w, h, OPT_III if mode512 else OPT_II, LZSS,
                                       np.ascontiguousarray(p0, np.uint32),
                                       np.ascontiguousarray(p1, np.uint32))
        self.L.refshim_set_frame_count(self.p, first_frame_count)
        self.buf = np.zeros(3 * w * h + 64, np.uint8)

    def encode(self, pix, want_entries=False):
        pix = np.ascontiguousarray(pix, np.uint32).reshape(-1)
        ent = np.zeros(self.w * self.h, np.uint16) if want_entries else None
        n = self.L.refshim_encode_frame_hot(self.p, pix, self.buf,
                                            ent.ctypes.data_as(C.c_void_p) if want_entries else None)
        out = self.buf[:n].copy()
        return (out, ent) if want_entries else out

    def close(self):
        if self.p:
            self.L.refshim_destroy(self.p)
            self.p = None

    __del__ = close


def oracle_decode_file(data, want_tables=False):
    """Decode a whole .agmv image (bytes) with the restatement. Yields per-frame dicts."""
    L = oracle()
    buf = np.frombuffer(data, np.uint8).copy()
    info = _FileInfo()
    p0 = np.zeros(256, np.uint32)
    p1 = np.zeros(256, np.uint32)
    err = L.orc_parse_header(buf, len(buf), C.byref(info), p0, p1)
    if err:
        return err, None, []
    d = L.orc_decoder_new(info.w, info.h, info.version, p0, p1)
    dec = C.cast(d, C.POINTER(_OrcDecoder)).contents
    pos = info.first_chunk
    frames = []
    nblk = info.w * info.h // 16
    for _ in range(info.num_frames):
        pos = L.orc_find_next_frame_chunk(buf, len(buf), pos)
        if pos + 16 > len(buf):
            break
        usize = int.from_bytes(buf[pos + 8:pos + 12].tobytes(), "little")
        csize = int.from_bytes(buf[pos + 12:pos + 16].tobytes(), "little")
        payload = pos + 16
        used = L.orc_decoder_lz(d, buf.ctypes.data + payload, len(buf) - payload, usize, csize)
        rec = {"usize": usize, "csize": csize, "bpos": int(dec.bpos)}
        if want_tables:
            rec["bitstream"] = np.ctypeslib.as_array(dec.bitstream, (dec.bpos + 16,)).copy()
            offs = np.zeros(nblk, np.uint32)
            n_ent = C.c_uint32(0)
            L.orc_decoder_parse(d, offs.ctypes.data_as(C.c_void_p), C.byref(n_ent))
            rec["offsets"] = offs
            rec["n_entered"] = n_ent.value
        else:
            L.orc_decoder_parse(d, None, None)
        rec["pix"] = np.ctypeslib.as_array(dec.img, (info.w * info.h,)).copy()
        frames.append(rec)
        pos = payload + used
    L.orc_decoder_free(d)
    return 0, info, frames


class OracleDecoder:
    """stateful D2-D4 (restatement) fed with already-decompressed bitstreams."""

    def __init__(self, w, h, mode512, p0, p1):
        self.L = oracle()
        self.w, self.h = w, h
        self.p = self.L.orc_decoder_new(w, h, 1 if mode512 else 2, np.ascontiguousarray(p0, np.uint32),
                                        np.ascontiguousarray(p1, np.uint32))
        self.s = C.cast(self.p, C.POINTER(_OrcDecoder)).contents

    def decode(self, bits, want_tables=False):
        bits = np.ascontiguousarray(bits, np.uint8)
        self.L.orc_decoder_set_bitstream(self.p, bits, len(bits))
        padded = np.ctypeslib.as_array(self.s.bitstream, (len(bits) + 16,)).copy()
        offs = np.zeros(self.w * self.h // 16, np.uint32)
        n_ent = C.c_uint32(0)
        self.L.orc_decoder_parse(self.p, offs.ctypes.data_as(C.c_void_p), C.byref(n_ent))
        pix = np.ctypeslib.as_array(self.s.img, (self.w * self.h,)).copy()
        if want_tables:
            return pix, padded, offs, n_ent.value
        return pix

    def close(self):
        if self.p:
            self.L.orc_decoder_free(self.p)
            self.p = None

    __del__ = close


class _OrcDecoder(C.Structure):
    _fields_ = [("w", C.c_uint32), ("h", C.c_uint32), ("version", C.c_int),
                ("p0", C.c_uint32 * 256), ("p1", C.c_uint32 * 256), ("frame_count", C.c_uint32),
                ("img", C.POINTER(C.c_uint32)), ("iframe", C.POINTER(C.c_uint32)),
                ("bitstream", C.POINTER(C.c_uint8)), ("bitstream_cap", C.c_size_t),
                ("bpos", C.c_uint32)]


def ref_decode_file(path, want_bitstream=False):
    """Decode with the compiled reference. Returns (err, (w,h,n,version), frames)."""
    L = ref()
    err = C.c_int(0)
    w, h, n, ver = (C.c_uint32(0) for _ in range(4))
    d = L.refshim_decoder_open(path.encode(), C.byref(err), C.byref(w), C.byref(h), C.byref(n),
                               C.byref(ver))
    frames = []
    if err.value == 0:
        for _ in range(n.value):
            pix = np.zeros(w.value * h.value, np.uint32)
            us, cs, bp = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
            e = L.refshim_decoder_next(d, pix.ctypes.data_as(C.c_void_p), C.byref(us), C.byref(cs),
                                       C.byref(bp))
            rec = {"pix": pix, "usize": us.value, "csize": cs.value, "bpos": bp.value, "err": e}
            if want_bitstream:
                bs = np.zeros(bp.value + 16, np.uint8)
                L.refshim_decoder_bitstream(d, bs, len(bs))
                rec["bitstream"] = bs
            frames.append(rec)
    L.refshim_decoder_close(d)
    return err.value, (w.value, h.value, n.value, ver.value), frames


def fnv1a64(arr, seed=0):
    a = np.ascontiguousarray(arr)
    return oracle().orc_fnv1a64(a.ctypes.data, a.nbytes, seed)
