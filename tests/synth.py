"""numpy statement of the canonical synthetic clip ``agmv_synth_v1``.

Independent of the product's C (libagmv_amd/csrc/agmv_synth.c) and HIP
(agmv_hip.hip: synth kernel) statements; tests check all three agree bit for bit.
Integer-only so every platform agrees.
"""
import numpy as np

M64 = np.uint64(0xFFFFFFFFFFFFFFFF)
DEFAULT_SEED = 0xA6D5


def splitmix64(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & M64
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & M64
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & M64
    return z ^ (z >> np.uint64(31))


def synth_frame(W, H, t, seed=DEFAULT_SEED):
    """frame t as (H, W) uint32 0x00RRGGBB."""
    with np.errstate(over="ignore"):
        x = np.arange(W, dtype=np.uint64)[None, :].repeat(H, 0)
        y = np.arange(H, dtype=np.uint64)[:, None].repeat(W, 1)
        te = np.where(x < np.uint64(W // 4), np.uint64(0), np.uint64(t))          # region A: static
        h = splitmix64(np.uint64(seed) ^ (te * np.uint64(0x9E3779B97F4A7C15)) ^ ((y << np.uint64(32)) | x))
        # region B: flat 32x32 tiles whose colour changes every 8 frames
        tile = ((y // np.uint64(32)) << np.uint64(40)) | ((x // np.uint64(32)) << np.uint64(20)) | (te // np.uint64(8))
        flat = splitmix64(np.uint64(seed) ^ tile) & np.uint64(0xFFFFFF)
        # elsewhere: moving gradient with sparse low-bit noise
        r = (x * np.uint64(255) // np.uint64(W - 1) + np.uint64(2) * te) & np.uint64(255)
        g = (y * np.uint64(255) // np.uint64(H - 1) + te) & np.uint64(255)
        b = ((x + y) // np.uint64(2) + np.uint64(3) * te) & np.uint64(255)
        noisy = (h & np.uint64(15)) == np.uint64(0)
        r = np.where(noisy, r ^ ((h >> np.uint64(8)) & np.uint64(7)), r)
        g = np.where(noisy, g ^ ((h >> np.uint64(16)) & np.uint64(7)), g)
        b = np.where(noisy, b ^ ((h >> np.uint64(24)) & np.uint64(7)), b)
        grad = (r << np.uint64(16)) | (g << np.uint64(8)) | b
        out = np.where(y >= np.uint64(3 * H // 4), flat, grad)
    return out.astype(np.uint32)


def random_palettes(seed, spread=True):
    """two 256-entry palettes of 0x00RRGGBB (test helper, not the reference's builder)."""
    rng = np.random.default_rng(seed)
    p = rng.integers(0, 1 << 24, size=512, dtype=np.uint32)
    if not spread:
        p = (p & np.uint32(0x3F3F3F)) + np.uint32(0x404040)
    return np.ascontiguousarray(p[:256]), np.ascontiguousarray(p[256:])


def content_palettes(frames, n=512):
    """deterministic test palette drawn from the clip itself: the sorted unique colours of the
    given frames, n of them evenly spaced (not the reference's histogram builder, which is the
    host-side 'next' row N3)."""
    u = np.unique(np.concatenate([np.asarray(f, np.uint32).reshape(-1) for f in frames]))
    idx = (np.arange(n, dtype=np.int64) * len(u)) // n
    p = u[idx].astype(np.uint32)
    return np.ascontiguousarray(p[0::2]), np.ascontiguousarray(p[1::2])
