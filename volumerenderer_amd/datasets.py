"""Synthetic volumes for the configurations whose datasets the reference does not ship.

BONSAI.img (256^3 u8) is listed by the reference as a missing large blob and has no loader there, so
the sweep configuration that names it runs on a seeded procedural stand-in of the same size and
value range: a dozen Gaussian blobs (a trunk and a crown of lobes) over low-amplitude value noise.
"""
import numpy as np


def bonsai_standin(n=256, seed=1234):
    """Deterministic n^3 u8 volume, x slowest / z fastest like Data.data (src/Data.zig:90-92)."""
    rng = np.random.default_rng(seed)
    ax = np.linspace(-1.0, 1.0, n, dtype=np.float32)
    x, y, z = ax[:, None, None], ax[None, :, None], ax[None, None, :]
    vol = np.zeros((n, n, n), dtype=np.float32)
    # trunk
    vol += 0.9 * np.exp(-((x * 6.0) ** 2 + (z * 6.0) ** 2)) * (y < 0.1)
    # crown: 12 lobes
    for _ in range(12):
        c = rng.uniform(-0.55, 0.55, 3).astype(np.float32)
        c[1] = abs(c[1]) * 0.8 + 0.05
        s = np.float32(rng.uniform(4.0, 9.0))
        a = np.float32(rng.uniform(0.45, 1.0))
        vol += a * np.exp(-(((x - c[0]) * s) ** 2 + ((y - c[1]) * s) ** 2 + ((z - c[2]) * s) ** 2))
    # coarse value noise, trilinearly upsampled by repetition (blocky on purpose: cheap and seeded)
    k = max(1, n // 16)
    coarse = rng.random((n // k + 1,) * 3, dtype=np.float32)
    vol += 0.08 * np.repeat(np.repeat(np.repeat(coarse, k, 0), k, 1), k, 2)[:n, :n, :n]
    vol = np.clip(vol / vol.max(), 0.0, 1.0)
    return np.ascontiguousarray((vol * 255.0).astype(np.uint8))


def gyroid_noise(n=1024, seed=42, periods=6.0, noise=24, shell=0.35):
    """Deterministic n^3 u8 volume beyond the reference's coordinate range (its SPACESIZE is u8): a gyroid
    shell -- density falls off with the distance |sin x cos y + sin y cos z + sin z cos x| from the gyroid
    surface -- plus uniform noise of amplitude `noise`.  x slowest / z fastest like Data.data
    (src/Data.zig:90-92).  Built slab by slab: the float intermediates never exceed one x-slab."""
    rng = np.random.default_rng(seed)
    w = np.float32(2.0 * np.pi * periods / n)
    ax = np.arange(n, dtype=np.float32) * w
    s, c = np.sin(ax), np.cos(ax)
    out = np.empty((n, n, n), dtype=np.uint8)
    yz = s[:, None] * c[None, :]  # sin y cos z
    for i in range(n):
        g = s[i] * c[:, None] + yz + s[None, :] * c[i]  # sin x cos y + sin y cos z + sin z cos x
        d = np.clip(1.0 - np.abs(g) / np.float32(shell), 0.0, 1.0) * np.float32(255 - noise)
        out[i] = d.astype(np.uint8) + rng.integers(0, noise + 1, (n, n), dtype=np.uint8)
    return out
