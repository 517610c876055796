"""ctypes view of libagmv_amd/libagmv.so (the libagmv-compatible host library) for tests."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "libagmv_amd", "libagmv.so")

u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u64p = np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS")

_lib = None


def lib():
    global _lib
    if _lib is None:
        from libagmv_amd import build
        build.build()
        L = C.CDLL(SO)
        L.agmv_lzss_mem.restype = C.c_ulong
        L.agmv_lzss_mem.argtypes = [u8p, C.c_size_t, u8p]
        L.agmv_lz77_mem.restype = C.c_ulong
        L.agmv_lz77_mem.argtypes = [u8p, C.c_size_t, u8p]
        L.agmv_lz_decode_mem.restype = C.c_ulong
        L.agmv_lz_decode_mem.argtypes = [C.c_int, u8p, C.c_size_t, C.c_ulong, C.c_ulong, u8p, C.c_size_t, C.POINTER(C.c_size_t)]
        L.AGMV_BuildPalette.restype = None
        L.AGMV_BuildPalette.argtypes = [u32p, C.c_int, C.c_int, u64p, u64p]
        L.AGMV_SynthFrame.restype = None
        L.AGMV_SynthFrame.argtypes = [u32p, C.c_uint, C.c_uint, C.c_uint, C.c_ulonglong]
        L.AGMV_BubbleSort.restype = None
        L.AGMV_BubbleSort.argtypes = [u64p, u64p, C.c_ulong]
        L.agmv_bmp_save.argtypes = [C.c_char_p, u32p, C.c_uint32, C.c_uint32]
        L.agmv_bmp_load.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.CreateAGMV.restype = C.c_void_p
        L.CreateAGMV.argtypes = [C.c_ulong] * 4
        L.DestroyAGMV.argtypes = [C.c_void_p]
        for f in ("AGMV_EncodeAGMV", "AGMV_EncodeFullAGMV"):
            getattr(L, f).restype = None
            getattr(L, f).argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_char_p, C.c_ubyte] + [C.c_ulong] * 5 + [C.c_int] * 3
        L.AGMV_EncodeVideo.restype = None
        L.AGMV_EncodeVideo.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_ubyte] + [C.c_ulong] * 5 + [C.c_int] * 3
        L.AGMV_DecodeAGMV.restype = C.c_int
        L.AGMV_DecodeAGMV.argtypes = [C.c_char_p, C.c_ubyte, C.c_int]
        L.AGMV_DecodeVideo.restype = C.c_int
        L.AGMV_DecodeVideo.argtypes = [C.c_char_p, C.c_ubyte]
        L.AGMV_QuantizeColor.restype = C.c_ulong
        L.AGMV_QuantizeColor.argtypes = [C.c_ulong, C.c_int]
        L.AGMV_ReverseQuantizeColor.restype = C.c_ulong
        L.AGMV_ReverseQuantizeColor.argtypes = [C.c_ulong, C.c_int]
        L.AGMV_GetVersionFromOPT.restype = C.c_ubyte
        L.AGMV_GetVersionFromOPT.argtypes = [C.c_int, C.c_int]
        L.AGMV_SetBatchFrames.argtypes = [C.c_uint]
        L.AGMV_SetLZThreads.argtypes = [C.c_uint]
        _lib = L
    return _lib


def lzss(data):
    data = np.ascontiguousarray(data, np.uint8)
    out = np.zeros(2 * len(data) + 64, np.uint8)
    cs = lib().agmv_lzss_mem(np.concatenate([data, np.zeros(8, np.uint8)]), len(data), out)
    return out[:cs].copy(), int(cs)


def lz77(data, tail_byte=0):
    data = np.ascontiguousarray(data, np.uint8)
    out = np.zeros(4 * len(data) + 64, np.uint8)
    buf = np.concatenate([data, np.full(8, tail_byte, np.uint8)])
    cs = lib().agmv_lz77_mem(buf, len(data), out)
    return out[:cs].copy(), int(cs)


def write_bmp(path, frame):
    frame = np.ascontiguousarray(frame, np.uint32)
    h, w = frame.shape
    assert lib().agmv_bmp_save(path.encode(), frame.reshape(-1), w, h) == 0
