"""Synthetic stand-in volumes and camera paths.

The reference ships no volumes (README.md:68 points at a release download), so every
benchmark/test volume is generated here, deterministically.  All volumes are fp32 in [0,1],
C-order ``[z][y][x]``, with values < 0.02 set to exactly 0 so that empty space is sparse
(mirrors ``ExternalImporter.cpp:154`` + ``copyFromDense(..., 0.001f)`` at ``:181``).
"""
import math

import numpy as np

SPARSITY_THRESHOLD = 0.02


def _sparsify(v):
    v[v < SPARSITY_THRESHOLD] = 0.0
    return v


def sphere64():
    """V64-sphere (BASELINE config #1): soft sphere of radius 20 voxels, iso 0.5."""
    z, y, x = np.meshgrid(np.arange(64, dtype=np.float32), np.arange(64, dtype=np.float32),
                          np.arange(64, dtype=np.float32), indexing="ij")
    r = np.sqrt((x - 31.5) ** 2 + (y - 31.5) ** 2 + (z - 31.5) ** 2)
    v = np.clip((20.0 - r) / 4.0 + 0.5, 0.0, 1.0).astype(np.float32)
    return _sparsify(v)


def slab64():
    """A field that is LINEAR along x inside a box (0 outside): node-centred trilinear interpolation reproduces a linear field exactly,
    so the isosurface of the relative isovalue q is the plane x = 15 + 32 q, for every ray that enters through the low-x face -- an
    analytic answer of a different kind than the sphere's (flat faces, long runs through one leaf row, a constant gradient)."""
    v = np.zeros((64, 64, 64), np.float32)
    ramp = (np.arange(16, 48, dtype=np.float32) - 15.0) / 32.0           # 1/32 .. 1 at x = 16 .. 47
    v[12:52, 12:52, 16:48] = ramp[None, None, :]
    return v


def _upsample_axis(a, n, axis):
    """Linear resampling of ``a`` along ``axis`` to ``n`` samples, endpoints aligned."""
    m = a.shape[axis]
    pos = np.arange(n, dtype=np.float64) * ((m - 1) / (n - 1))
    i0 = np.minimum(np.floor(pos).astype(np.int64), m - 2)
    w = (pos - i0).astype(np.float32)
    shape = [1] * a.ndim
    shape[axis] = n
    w = w.reshape(shape)
    lo = np.take(a, i0, axis=axis)
    hi = np.take(a, i0 + 1, axis=axis)
    return lo + (hi - lo) * w


def _value_noise(rng, lattices, n, zrange=None):
    """Sum of trilinearly upsampled uniform lattices, weights 1, 1/2, 1/4, ..., normalised."""
    total = None
    wsum = 0.0
    for o, l in enumerate(lattices):
        lat = rng.random((l, l, l), dtype=np.float32)
        wgt = 0.5 ** o
        a = _upsample_axis(lat, n, 0)
        if zrange is not None:
            a = a[zrange[0]:zrange[1]]
        a = _upsample_axis(a, n, 1)
        a = _upsample_axis(a, n, 2)
        a *= np.float32(wgt)
        total = a if total is None else total + a
        wsum += wgt
    total /= np.float32(wsum)
    return total


def _radius(n, zrange=None):
    c = (np.arange(n, dtype=np.float32) + 0.5) / n - 0.5
    cz = c if zrange is None else c[zrange[0]:zrange[1]]
    return np.sqrt(cz[:, None, None] ** 2 + c[None, :, None] ** 2 + c[None, None, :] ** 2)


def ejecta(n=256, seed=272):
    """V256-ejecta (BASELINE config #2): value noise x spherical shell, iso 0.34."""
    rng = np.random.default_rng(seed)
    lattices = [8, 16, 32] if n >= 64 else [4, 8, 16]
    v = _value_noise(rng, lattices, n)
    r = _radius(n)
    v *= np.exp(-((r - 0.30) / 0.12) ** 2).astype(np.float32)
    v /= v.max()
    return _sparsify(v.astype(np.float32))


def cloud(n=512, seed=49):
    """V512-cloud (BASELINE config #4): 5-octave value noise x gaussian blob, iso 0.30."""
    rng = np.random.default_rng(seed)
    lattices = [4, 8, 16, 32, 64]
    v = _value_noise(rng, lattices, n)
    r = _radius(n)
    v *= np.exp(-(r / 0.35) ** 2).astype(np.float32)
    v /= v.max()
    return _sparsify(v.astype(np.float32))


class EjectaField:
    """The ejecta recipe evaluated region by region, for volumes no single process wants to hold
    (V1024, BASELINE config #5: "8 tiles of 512^3 generated tile-wise from the same global lattice",
    SURVEY.md 8(d)).  ``raw(box)`` is the un-normalised field of an index box, ``finalize(raw, gmax)``
    rescales by the GLOBAL maximum (the maximum over all tiles' raw maxima -- one scalar max-reduction
    between ranks) and applies the sparsity threshold.  Every step is element-wise, so
    ``finalize(raw(everything), raw(everything).max())`` is bit for bit ``ejecta(n, seed)``."""

    def __init__(self, n=1024, seed=1024):
        self.n = n
        rng = np.random.default_rng(seed)
        self.lattices = [rng.random((l, l, l), dtype=np.float32) for l in ([8, 16, 32] if n >= 64 else [4, 8, 16])]

    def raw(self, box):
        z0, z1, y0, y1, x0, x1 = box
        n = self.n
        total, wsum = None, 0.0
        for o, lat in enumerate(self.lattices):
            wgt = 0.5 ** o
            a = _upsample_axis(lat, n, 0)[z0:z1]
            a = _upsample_axis(a, n, 1)[:, y0:y1]
            a = _upsample_axis(a, n, 2)[:, :, x0:x1]
            a = a * np.float32(wgt)
            total = a if total is None else total + a
            wsum += wgt
        total /= np.float32(wsum)
        c = (np.arange(n, dtype=np.float32) + 0.5) / n - 0.5
        r = np.sqrt(c[z0:z1, None, None] ** 2 + c[None, y0:y1, None] ** 2 + c[None, None, x0:x1] ** 2)
        total *= np.exp(-((r - 0.30) / 0.12) ** 2).astype(np.float32)
        return total

    @staticmethod
    def finalize(raw, gmax):
        v = raw / np.float32(gmax)
        return _sparsify(v.astype(np.float32))


VOLUMES = {
    "sphere64": (sphere64, 0.5),
    "ejecta256": (lambda: ejecta(256), 0.34),
    "ejecta128": (lambda: ejecta(128), 0.34),
    "cloud512": (lambda: cloud(512), 0.30),
}


def orbit_camera(k, K=64, distance=2.0, pitch=0.38):
    """Camera origin on the orbit of SURVEY.md 8(d): distance 2.0, pitch 0.38 rad, yaw 2*pi*k/K.
    Follows the spherical convention of ``inference/camera.py:61-66`` (orientation Yp)."""
    yaw = 2.0 * math.pi * k / K
    return [math.cos(pitch) * math.cos(yaw) * distance,
            math.sin(pitch) * distance,
            math.cos(pitch) * math.sin(yaw) * distance]


def fmt3(v):
    """Format a 3-vector exactly as the reference's callers do (``mainGUI.py:667-671``)."""
    return "%5.3f,%5.3f,%5.3f" % (v[0], v[1], v[2])


def quantize3(v):
    """Value a renderer sees after the caller's %5.3f formatting."""
    return [float(s) for s in fmt3(v).split(",")]
