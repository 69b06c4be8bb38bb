"""Synthetic light-field generator of SURVEY.md section 8(d): integer-exact, numpy only.

Background = 3 octaves of bilinearly interpolated lattice value-noise (pitches 64/16/4, amplitudes
64/32/16) + 128 on a canvas, foreground = 12 constant-colour axis-aligned rectangles; SAI (s, t)
samples the background at integer offset 1*(s-cs, t-ct) and the foreground at 2*(s-cs, t-ct)
(disparities of 1 and 2 px per view), centre crop W x H, clipped to [0,255], integer valued.
All randomness comes from one LCG stream x <- 1664525 x + 1013904223 (mod 2^32), seed 12345,
top 16 bits of each output.
"""
import numpy as np


class _LCG:
    def __init__(self, seed):
        self.x = seed & 0xFFFFFFFF

    def take(self, n):
        out = np.empty(n, np.int64)
        x = self.x
        for i in range(n):
            x = (1664525 * x + 1013904223) & 0xFFFFFFFF
            out[i] = x >> 16
        self.x = x
        return out


def _canvas(size, rng):
    """[3][size][size] int64 background, scaled by 1 (grey levels)."""
    acc = np.zeros((3, size, size), np.int64)
    yy = np.arange(size)
    for pitch, amp in ((64, 64), (16, 32), (4, 16)):
        n = size // pitch + 2
        for c in range(3):
            lat = (rng.take(n * n).reshape(n, n) * (2 * amp) // 65536) - amp   # ints in [-amp, amp)
            i0, f = yy // pitch, yy % pitch
            w1 = f[:, None] * np.ones((1, size), np.int64)          # fy
            w2 = np.ones((size, 1), np.int64) * f[None, :]          # fx
            a = lat[i0][:, i0]
            b = lat[i0][:, i0 + 1]
            cc = lat[i0 + 1][:, i0]
            d = lat[i0 + 1][:, i0 + 1]
            P = pitch
            v = a * (P - w1) * (P - w2) + b * (P - w1) * w2 + cc * w1 * (P - w2) + d * w1 * w2
            acc[c] += np.floor_divide(v, P * P)
    return acc + 128


def make_lf(aheight, awidth, H, W, seed=12345):
    """Returns uint8 array [aheight*awidth][3][H][W], st = s*awidth + t (row-major)."""
    cs, ct = aheight // 2, awidth // 2
    margin = 2 * max(cs, ct) + 8
    size = ((max(H, W) + 2 * margin + 63) // 64) * 64
    rng = _LCG(seed)
    bg = _canvas(size, rng)
    r = rng.take(12 * 7).reshape(12, 7)
    rects = []
    for q in r:
        x0, y0 = q[0] * size // 65536, q[1] * size // 65536
        w, h = 24 + q[2] * 96 // 65536, 24 + q[3] * 96 // 65536
        col = (q[4] * 256 // 65536, q[5] * 256 // 65536, q[6] * 256 // 65536)
        rects.append((int(x0), int(y0), int(w), int(h), col))
    oy, ox = (size - H) // 2, (size - W) // 2
    # foreground layer on its own canvas (later rectangles overwrite earlier ones)
    fg = np.zeros((3, size, size), np.uint8)
    bg = np.clip(bg, 0, 255).astype(np.uint8)
    fgm = np.zeros((size, size), bool)
    for (x0, y0, w, h, col) in rects:
        y1, x1 = min(size, y0 + h), min(size, x0 + w)
        fgm[y0:y1, x0:x1] = True
        for c in range(3):
            fg[c, y0:y1, x0:x1] = col[c]
    out = np.empty((aheight * awidth, 3, H, W), np.uint8)
    for s in range(aheight):
        for t in range(awidth):
            ds, dt = s - cs, t - ct
            img = bg[:, oy + ds:oy + ds + H, ox + dt:ox + dt + W]
            y2, x2 = oy + 2 * ds, ox + 2 * dt                     # foreground sampled at 2*(ds, dt)
            m = fgm[y2:y2 + H, x2:x2 + W]
            img = np.where(m[None], fg[:, y2:y2 + H, x2:x2 + W], img)
            out[s * awidth + t] = img
    return out


def add_noise_mt19937(clean, sigma, seed=1, out=None):
    """Additive white Gaussian noise exactly as the reference's add_noise draws it (utilities.cpp:176-183 with
    mt19937ar.c): ONE MT19937 stream seeded with init_genrand(seed), samples in memory order (SAIs in st order,
    planar pixels), per sample a = genrand_res53(), b = genrand_res53(), z = sigma * sqrt(-2 ln a) * cos(2 pi b)
    in double, added as float.  numpy's legacy RandomState(seed) is that generator (init_genrand seeding, and
    random_sample() is genrand_res53), so no C code is needed.  clean: float32 array of any shape; returns float32.
    """
    import os
    from concurrent.futures import ThreadPoolExecutor
    clean = np.ascontiguousarray(clean, dtype=np.float32)
    res = np.empty_like(clean) if out is None else out
    flat_in, flat_out = clean.reshape(-1), res.reshape(-1)
    rs = np.random.RandomState(int(seed) & 0xFFFFFFFF)
    workers = max(1, min(32, (os.cpu_count() or 1)))
    piece = 1 << 18

    def box_muller(args):          # numpy releases the GIL inside log / sqrt / cos
        u, i, n = args
        z = float(sigma) * np.sqrt(-2.0 * np.log(u[0:2 * n:2])) * np.cos(2.0 * np.pi * u[1:2 * n:2])
        flat_out[i:i + n] = flat_in[i:i + n] + z.astype(np.float32)

    with ThreadPoolExecutor(workers) as pool:
        chunk = piece * workers
        for i0 in range(0, flat_in.size, chunk):      # the stream itself is drawn sequentially, in order
            n0 = min(chunk, flat_in.size - i0)
            u = rs.random_sample(2 * n0)
            jobs = [(u[2 * j:2 * min(j + piece, n0)], i0 + j, min(piece, n0 - j)) for j in range(0, n0, piece)]
            list(pool.map(box_muller, jobs))
    return res
