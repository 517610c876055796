"""ctypes binding of include/agmv_hip.h.

The product is the shared library; this module only marshals pointers.  torch is used for
device memory and streams (plumbing).  Every failure is loud: a missing library raises
HipUnavailable at load time, a failing call raises RuntimeError with the library's message.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

# every symbol include/agmv_hip.h declares (tests check the built library exports them all)
ABI_SYMBOLS = [
    "agmv_hip_max_usize", "agmv_hip_device_count", "agmv_hip_create", "agmv_hip_destroy",
    "agmv_hip_last_error", "agmv_hip_set_palette", "agmv_hip_quantise_dev",
    "agmv_hip_encode_frames_dev", "agmv_hip_encode_frames", "agmv_hip_encode_entries_dev",
    "agmv_hip_encode_entries", "agmv_hip_nearest", "agmv_hip_within2_count", "agmv_hip_parse_frames_dev",
    "agmv_hip_decode_frames_dev", "agmv_hip_parse_decode_frames_dev", "agmv_hip_decode_bitstreams_dev", "agmv_hip_pack_frames_dev", "agmv_hip_unpack_frames_dev", "agmv_hip_parse_fallback_frames", "agmv_hip_decode_frames", "agmv_hip_decode_prior_dependent", "agmv_hip_synth_dev",
    "agmv_hip_interp_dev", "agmv_hip_histogram_dev", "agmv_hip_check", "agmv_hip_malloc",
    "agmv_hip_free", "agmv_hip_memcpy_h2d", "agmv_hip_memcpy_d2h", "agmv_hip_memset",
    "agmv_hip_sync", "agmv_hip_enable_timing", "agmv_hip_last_kernel_ms",
    "agmv_hip_stream_create", "agmv_hip_stream_destroy", "agmv_hip_stream_sync", "agmv_hip_host_alloc",
    "agmv_hip_host_free", "agmv_hip_malloc_on", "agmv_hip_free_on", "agmv_hip_memcpy_async",
    "agmv_hip_memset_async", "agmv_hip_ctx_device",
]


class HipUnavailable(RuntimeError):
    pass


def lib_path():
    # AGMV_HIP_LIB: tools/ablate.sh points the probe at instrumented builds of the same source
    return os.environ.get("AGMV_HIP_LIB") or os.path.join(HERE, "libagmv_hip.so")


_libs = {}


def load_library(path=None):
    """dlopen libagmv_hip.so (built in-tree by libagmv_amd/build.py). No fallback.
    `path` selects another build of the same source (tools/probe_multi.py times several in one process)."""
    p = path or lib_path()
    if p in _libs:
        return _libs[p]
    if not os.path.exists(p):
        raise HipUnavailable("%s is missing: run `python -m libagmv_amd.build` (or "
                             "__graft_entry__.build()); the AGMV hot path has no CPU fallback" % p)
    # PyTorch ships its own copies of the HIP / HSA runtimes.  This module works on torch device tensors, so torch's runtime is
    # loaded FIRST: with libagmv_hip.so (linked against /opt/rocm) in the process before `import torch`, the two resolve against
    # each other's libraries and hipGetDeviceCount then reports no device (seen with build() followed by smoke() in one process).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    try:
        L = C.CDLL(p)
    except OSError as e:
        raise HipUnavailable("cannot load %s: %s" % (p, e))
    vp, sz, u32 = C.c_void_p, C.c_size_t, C.c_uint32
    L.agmv_hip_max_usize.restype = sz
    L.agmv_hip_max_usize.argtypes = [u32, u32, C.c_int]
    L.agmv_hip_device_count.restype = C.c_int
    L.agmv_hip_create.restype = vp
    L.agmv_hip_create.argtypes = [C.c_int]
    L.agmv_hip_destroy.argtypes = [vp]
    L.agmv_hip_last_error.restype = C.c_char_p
    L.agmv_hip_set_palette.argtypes = [vp, vp, vp, C.c_int, vp]
    L.agmv_hip_quantise_dev.argtypes = [vp, vp, sz, vp, vp]
    L.agmv_hip_encode_frames_dev.argtypes = [vp, vp, u32, u32, u32, u32, vp, sz, vp, vp, vp]
    L.agmv_hip_encode_frames.argtypes = [vp, vp, u32, u32, u32, u32, vp, sz, vp, vp]
    L.agmv_hip_encode_entries_dev.argtypes = [vp, vp, u32, u32, u32, u32, vp, sz, vp, vp, vp]
    L.agmv_hip_encode_entries.argtypes = [vp, vp, u32, u32, u32, u32, vp, sz, vp, vp]
    L.agmv_hip_nearest.argtypes = [vp, vp, vp, C.c_int, vp, sz, vp]
    L.agmv_hip_within2_count.argtypes = [vp, vp, vp]
    L.agmv_hip_within2_count.restype = C.c_int
    L.agmv_hip_parse_frames_dev.argtypes = [vp, vp, sz, vp, u32, u32, u32, vp, vp, vp]
    L.agmv_hip_decode_frames_dev.argtypes = [vp, vp, sz, vp, vp, vp, u32, u32, u32, u32, vp, vp, vp, vp]
    L.agmv_hip_decode_frames.argtypes = [vp, vp, sz, vp, u32, u32, u32, u32, vp, vp, vp]
    L.agmv_hip_parse_decode_frames_dev.argtypes = [vp, vp, sz, vp, u32, u32, u32, u32, vp, vp, vp, vp, vp, vp]
    if path is None or hasattr(L, "agmv_hip_decode_bitstreams_dev"):    # (an explicit `path` may be an older build kept for A/B timing, tools/variants/)
        L.agmv_hip_decode_bitstreams_dev.argtypes = [vp, vp, sz, vp, u32, u32, u32, u32, vp, vp, vp, vp, vp]
        L.agmv_hip_decode_bitstreams_dev.restype = C.c_int
    L.agmv_hip_decode_prior_dependent.argtypes = [vp, u32, u32, vp]
    L.agmv_hip_parse_fallback_frames.argtypes = [vp, vp]
    if path is None or hasattr(L, "agmv_hip_pack_frames_dev"):      # (older builds under tools/variants/ lack it)
        L.agmv_hip_pack_frames_dev.argtypes = [vp, vp, C.c_size_t, vp, C.c_uint32, vp, vp, vp]
        L.agmv_hip_pack_frames_dev.restype = C.c_int
        L.agmv_hip_unpack_frames_dev.argtypes = [vp, vp, vp, C.c_uint32, vp, C.c_size_t, vp, vp]
        L.agmv_hip_unpack_frames_dev.restype = C.c_int
    L.agmv_hip_parse_fallback_frames.restype = C.c_int
    L.agmv_hip_decode_prior_dependent.restype = C.c_int
    L.agmv_hip_synth_dev.argtypes = [vp, vp, u32, u32, u32, u32, C.c_uint64, vp]
    L.agmv_hip_interp_dev.argtypes = [vp, vp, vp, vp, sz, vp]
    L.agmv_hip_histogram_dev.argtypes = [vp, vp, sz, C.c_int, vp, vp]
    L.agmv_hip_check.argtypes = [vp, vp]
    L.agmv_hip_enable_timing.argtypes = [vp, C.c_int]
    L.agmv_hip_enable_timing.restype = C.c_int
    L.agmv_hip_last_kernel_ms.argtypes = [vp, C.c_int]
    L.agmv_hip_last_kernel_ms.restype = C.c_float
    L.agmv_hip_malloc.restype = vp
    L.agmv_hip_malloc.argtypes = [sz]
    L.agmv_hip_free.argtypes = [vp]
    for f in (L.agmv_hip_set_palette, L.agmv_hip_quantise_dev, L.agmv_hip_encode_frames_dev,
              L.agmv_hip_encode_frames, L.agmv_hip_encode_entries_dev, L.agmv_hip_encode_entries, L.agmv_hip_nearest, L.agmv_hip_parse_frames_dev, L.agmv_hip_decode_frames_dev,
              L.agmv_hip_decode_frames, L.agmv_hip_parse_decode_frames_dev, L.agmv_hip_synth_dev, L.agmv_hip_interp_dev,
              L.agmv_hip_histogram_dev, L.agmv_hip_check):
        f.restype = C.c_int
    _libs[p] = L
    return L


def _np_ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class AgmvHip:
    """One context = one GPU + one palette.  Device-resident calls take torch CUDA tensors
    (uint8 / int16 / int32 storage: torch has no unsigned 16/32-bit arithmetic types, the bytes
    are what matters) and run on torch's current stream."""

    def __init__(self, device=0, lib=None):
        self.L = load_library(lib)
        self.ctx = self.L.agmv_hip_create(int(device))
        if not self.ctx:
            raise HipUnavailable(self.L.agmv_hip_last_error().decode())
        self.device = int(device)
        self.mode512 = None

    def close(self):
        if getattr(self, "ctx", None):
            self.L.agmv_hip_destroy(self.ctx)
            self.ctx = None

    __del__ = close

    # ------------------------------------------------------------------ helpers
    def _ck(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.agmv_hip_last_error().decode())

    @staticmethod
    def _stream():
        import torch
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def max_usize(self, w, h):
        return self.L.agmv_hip_max_usize(w, h, int(self.mode512))

    def set_palette(self, p0, p1, mode512=True):
        p0 = np.ascontiguousarray(p0, np.uint32)
        p1 = np.ascontiguousarray(p1, np.uint32) if p1 is not None else np.zeros(256, np.uint32)
        self.mode512 = bool(mode512)
        self._ck(self.L.agmv_hip_set_palette(self.ctx, _np_ptr(p0), _np_ptr(p1), int(mode512), self._stream()))

    def enable_timing(self, on=True):
        self._ck(self.L.agmv_hip_enable_timing(self.ctx, int(on)))

    def last_kernel_ms(self, which):
        return float(self.L.agmv_hip_last_kernel_ms(self.ctx, which))

    def check(self):
        self._ck(self.L.agmv_hip_check(self.ctx, self._stream()))

    # ------------------------------------------------------------------ device-resident path
    def quantise_dev(self, pix):
        import torch
        out = torch.empty(pix.numel(), dtype=torch.int16, device=pix.device)
        self._ck(self.L.agmv_hip_quantise_dev(self.ctx, pix.data_ptr(), pix.numel(), out.data_ptr(), self._stream()))
        return out

    def encode_dev(self, pix, n_frames, w, h, first_frame_count=0, out=None, sizes=None, ientries=None):
        """pix: int32 CUDA tensor of n_frames*w*h pixels. Returns (out u8 [n, stride], sizes i32 [n])."""
        import torch
        stride = self.max_usize(w, h)
        if out is None:
            out = torch.empty((n_frames, stride), dtype=torch.uint8, device=pix.device)
        if sizes is None:
            sizes = torch.empty(n_frames, dtype=torch.int32, device=pix.device)
        self._ck(self.L.agmv_hip_encode_frames_dev(
            self.ctx, pix.data_ptr(), n_frames, w, h, first_frame_count, out.data_ptr(), out.stride(0),
            sizes.data_ptr(), ientries.data_ptr() if ientries is not None else None, self._stream()))
        return out, sizes

    def encode_entries_host(self, entries, first_frame_count=0, ientries=None):
        """entries: uint32 ndarray [n, h, w] of pal_num << 8 | index. Returns the per-frame bitstreams."""
        entries = np.ascontiguousarray(entries, np.uint32)
        n, h, w = entries.shape
        stride = self.max_usize(w, h)
        out = np.zeros((n, stride), np.uint8)
        sizes = np.zeros(n, np.uint32)
        self._ck(self.L.agmv_hip_encode_entries(self.ctx, _np_ptr(entries), n, w, h, first_frame_count,
                                                _np_ptr(out), stride, _np_ptr(sizes), _np_ptr(ientries)))
        return [out[i, :sizes[i]].copy() for i in range(n)]

    def nearest_host(self, p0, p1, pix, mode512=True):
        """exact nearest entries of `pix` against (p0, p1) without building a table"""
        p0 = np.ascontiguousarray(p0, np.uint32)
        p1 = np.ascontiguousarray(p1 if p1 is not None else np.zeros(256), np.uint32)
        pix = np.ascontiguousarray(pix, np.uint32).reshape(-1)
        out = np.zeros(pix.size, np.uint16)
        self._ck(self.L.agmv_hip_nearest(self.ctx, _np_ptr(p0), _np_ptr(p1), int(mode512), _np_ptr(pix), pix.size, _np_ptr(out)))
        return out

    def parse_dev(self, bits, bpos, n_frames, w, h, offsets=None, nentered=None):
        import torch
        nblk = (w // 4) * (h // 4)
        if offsets is None:
            offsets = torch.empty((n_frames, nblk), dtype=torch.int32, device=bits.device)
        if nentered is None:
            nentered = torch.empty(n_frames, dtype=torch.int32, device=bits.device)
        self._ck(self.L.agmv_hip_parse_frames_dev(self.ctx, bits.data_ptr(), bits.stride(0), bpos.data_ptr(),
                                                  n_frames, w, h, offsets.data_ptr(), nentered.data_ptr(),
                                                  self._stream()))
        return offsets, nentered

    def decode_dev(self, bits, bpos, offsets, nentered, n_frames, w, h, first_frame_count=0, out=None,
                   prev=None, prev_iframe=None):
        import torch
        if out is None:
            out = torch.empty((n_frames, h, w), dtype=torch.int32, device=bits.device)
        self._ck(self.L.agmv_hip_decode_frames_dev(
            self.ctx, bits.data_ptr(), bits.stride(0), bpos.data_ptr(), offsets.data_ptr(), nentered.data_ptr(),
            n_frames, w, h, first_frame_count, out.data_ptr(),
            prev.data_ptr() if prev is not None else None,
            prev_iframe.data_ptr() if prev_iframe is not None else None, self._stream()))
        return out

    def pack_frames_dev(self, out, sizes, total=None):
        """slab [n, stride] u8 + sizes [n] i32 (CUDA) -> (packed u8 [sum(sizes)], offsets i64 [n + 1]); `total` = sum(sizes) if the
        caller knows it (otherwise one device -> host read)"""
        import torch
        n = int(sizes.numel())
        if total is None:
            total = int(sizes.sum().item()) if n else 0
        packed = torch.empty(total, dtype=torch.uint8, device=out.device)
        offs = torch.zeros(n + 1, dtype=torch.int64, device=out.device)
        if n and total:                                        # (nothing but empty frames: the offsets are all 0)
            self._ck(self.L.agmv_hip_pack_frames_dev(self.ctx, out.data_ptr(), out.stride(0), sizes.data_ptr(), n,
                                                     packed.data_ptr(), offs.data_ptr(), self._stream()))
        return packed, offs

    def unpack_frames_dev(self, packed, sizes, stride, out=None):
        """the inverse: packed u8 + sizes [n] i32 (CUDA) -> slab [n, stride] u8 (rows zero behind their size unless `out` is given)"""
        import torch
        n = int(sizes.numel())
        if out is None:
            out = torch.zeros((n, stride), dtype=torch.uint8, device=packed.device)
        offs = torch.empty(n + 1, dtype=torch.int64, device=packed.device)
        if n and packed.numel():
            self._ck(self.L.agmv_hip_unpack_frames_dev(self.ctx, packed.data_ptr(), sizes.data_ptr(), n, out.data_ptr(), out.stride(0),
                                                       offs.data_ptr(), self._stream()))
        return out

    def parse_fallback_frames(self):
        """frames of the last parse that went to the robust kernels (a statistic)"""
        rc = self.L.agmv_hip_parse_fallback_frames(self.ctx, self._stream())
        if rc < 0:
            raise RuntimeError(self.L.agmv_hip_last_error().decode())
        return rc

    def parse_decode_dev(self, bits, bpos, n_frames, w, h, first_frame_count=0, out=None, offsets=None, nentered=None,
                         prev=None, prev_iframe=None):
        """parse + reconstruct in one overlapped call; returns (pixels, offsets, nentered)"""
        import torch
        nblk = (w // 4) * (h // 4)
        if offsets is None:
            offsets = torch.empty((n_frames, nblk), dtype=torch.int32, device=bits.device)
        if nentered is None:
            nentered = torch.empty(n_frames, dtype=torch.int32, device=bits.device)
        if out is None:
            out = torch.empty((n_frames, h, w), dtype=torch.int32, device=bits.device)
        self._ck(self.L.agmv_hip_parse_decode_frames_dev(
            self.ctx, bits.data_ptr(), bits.stride(0), bpos.data_ptr(), n_frames, w, h, first_frame_count,
            offsets.data_ptr(), nentered.data_ptr(), out.data_ptr(),
            prev.data_ptr() if prev is not None else None,
            prev_iframe.data_ptr() if prev_iframe is not None else None, self._stream()))
        return out, offsets, nentered

    def decode_bitstreams_dev(self, bits, bpos, n_frames, w, h, first_frame_count=0, out=None, nentered=None,
                              prev=None, prev_iframe=None):
        """parse + reconstruct without offsets[] (entry bitmaps straight into k_decode); returns the pixels"""
        import torch
        if out is None:
            out = torch.empty((n_frames, h, w), dtype=torch.int32, device=bits.device)
        self._ck(self.L.agmv_hip_decode_bitstreams_dev(
            self.ctx, bits.data_ptr(), bits.stride(0), bpos.data_ptr(), n_frames, w, h, first_frame_count,
            nentered.data_ptr() if nentered is not None else None, out.data_ptr(),
            prev.data_ptr() if prev is not None else None,
            prev_iframe.data_ptr() if prev_iframe is not None else None, self._stream()))
        return out

    def decode_depends_on_prior(self, w, h):
        """after decode_dev: does any pixel of that batch derive from prev / prev_iframe (see agmv_hip.h)?"""
        rc = self.L.agmv_hip_decode_prior_dependent(self.ctx, w, h, self._stream())
        if rc < 0:
            raise RuntimeError(self.L.agmv_hip_last_error().decode())
        return bool(rc)

    def synth_dev(self, w, h, t0, n_frames, seed=0xA6D5, out=None, device=None):
        import torch
        if out is None:
            out = torch.empty((n_frames, h, w), dtype=torch.int32, device=device or ("cuda:%d" % self.device))
        self._ck(self.L.agmv_hip_synth_dev(self.ctx, out.data_ptr(), w, h, t0, n_frames, seed, self._stream()))
        return out

    def interp_dev(self, f1, f2):
        import torch
        out = torch.empty_like(f1)
        self._ck(self.L.agmv_hip_interp_dev(self.ctx, out.data_ptr(), f1.data_ptr(), f2.data_ptr(), f1.numel(),
                                            self._stream()))
        return out

    def histogram_dev(self, pix, quality=1, hist=None):
        import torch
        if hist is None:
            hist = torch.zeros(1 << 19, dtype=torch.int32, device=pix.device)
        self._ck(self.L.agmv_hip_histogram_dev(self.ctx, pix.data_ptr(), pix.numel(), quality, hist.data_ptr(),
                                               self._stream()))
        return hist

    # ------------------------------------------------------------------ host-buffer path
    def encode_host(self, frames, first_frame_count=0, ientries=None):
        """frames: uint32 ndarray [n, h, w]. Returns list of per-frame bitstreams (uint8 arrays)."""
        frames = np.ascontiguousarray(frames, np.uint32)
        n, h, w = frames.shape
        stride = self.max_usize(w, h)
        out = np.zeros((n, stride), np.uint8)
        sizes = np.zeros(n, np.uint32)
        self._ck(self.L.agmv_hip_encode_frames(self.ctx, _np_ptr(frames), n, w, h, first_frame_count,
                                               _np_ptr(out), stride, _np_ptr(sizes), _np_ptr(ientries)))
        return [out[i, :sizes[i]].copy() for i in range(n)]

    def decode_host(self, bits_list, w, h, first_frame_count=0, prev=None, prev_iframe=None, pad=None):
        """bits_list: per-frame decompressed bitstreams. `pad`[f] = the 16 bytes that follow bpos in
        the reference's persistent buffer (stale bytes), zeros if None."""
        n = len(bits_list)
        stride = (max(len(b) for b in bits_list) + 16 + 255) & ~255
        bits = np.zeros((n, stride), np.uint8)
        bpos = np.zeros(n, np.uint32)
        for i, b in enumerate(bits_list):
            bits[i, :len(b)] = b
            bpos[i] = len(b)
            if pad is not None:
                bits[i, len(b):len(b) + 16] = pad[i]
        out = np.zeros((n, h, w), np.uint32)
        self._ck(self.L.agmv_hip_decode_frames(self.ctx, _np_ptr(bits), stride, _np_ptr(bpos), n, w, h,
                                               first_frame_count, _np_ptr(out), _np_ptr(prev), _np_ptr(prev_iframe)))
        return out
