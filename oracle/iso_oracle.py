"""ctypes wrapper around oracle/libiso_oracle.so -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
(see oracle/iso_oracle.h for the "parity unpinned" statement).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class IsoParams(ctypes.Structure):
    _fields_ = [
        ("width", ctypes.c_int), ("height", ctypes.c_int),
        ("fov_deg", ctypes.c_double),
        ("origin", ctypes.c_double * 3), ("lookat", ctypes.c_double * 3), ("up", ctypes.c_double * 3),
        ("last_origin", ctypes.c_double * 3), ("last_lookat", ctypes.c_double * 3),
        ("isovalue", ctypes.c_double),
        ("ambient", ctypes.c_double * 3), ("diffuse", ctypes.c_double * 3), ("specular", ctypes.c_double * 3),
        ("specular_exponent", ctypes.c_int),
        ("light_from_camera", ctypes.c_int),
        ("light_dir", ctypes.c_double * 3),
        ("viewport", ctypes.c_int * 4),
        ("ao_samples", ctypes.c_int),
        ("ao_radius", ctypes.c_double),
    ]


def build(force=False):
    so = os.path.join(_HERE, "libiso_oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("iso_oracle.c", "iso_oracle_gvdb.c", "iso_oracle.h", "iso_oracle_priv.h")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libiso_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libiso_oracle.so")
        if not os.path.exists(so):
            build()
        L = ctypes.CDLL(so)
        L.iso_volume_create.restype = ctypes.c_void_p
        L.iso_volume_create.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.iso_volume_create_tile.restype = ctypes.c_void_p
        L.iso_volume_create_tile.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int] + [ctypes.c_void_p] * 3 + [ctypes.c_float] + [ctypes.c_void_p] * 2
        L.iso_volume_free.argtypes = [ctypes.c_void_p]
        L.iso_volume_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.iso_params_default.argtypes = [ctypes.POINTER(IsoParams)]
        L.iso_render.restype = ctypes.c_int
        L.iso_render.argtypes = [ctypes.c_void_p, ctypes.POINTER(IsoParams), ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.iso_num_threads.restype = ctypes.c_int
        L.iso_render_gvdb.restype = ctypes.c_int
        L.iso_render_gvdb.argtypes = [ctypes.c_void_p, ctypes.POINTER(IsoParams), ctypes.c_void_p, ctypes.c_int]
        L.iso_ao_tables.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        _LIB = L
    return _LIB


class OracleVolume:
    def __init__(self, dense, tile=None):
        """tile: dict(origin, gmin, gmax, gmaxval, clip_lo, clip_hi) in (x,y,z) order for a tile of a larger volume."""
        dense = np.ascontiguousarray(dense, dtype=np.float32)
        assert dense.ndim == 3
        nz, ny, nx = dense.shape
        if tile is not None:
            i3 = lambda v: (ctypes.c_int * 3)(*[int(a) for a in v])
            self._h = lib().iso_volume_create_tile(dense.ctypes.data, nx, ny, nz, i3(tile['origin']), i3(tile['gmin']), i3(tile['gmax']),
                                                   ctypes.c_float(tile['gmaxval']), i3(tile['clip_lo']), i3(tile['clip_hi']))
        else:
            self._h = lib().iso_volume_create(dense.ctypes.data, nx, ny, nz)
        if not self._h:
            raise ValueError("oracle: empty or oversized volume")
        self.shape = dense.shape

    def info(self):
        info = (ctypes.c_int * 13)()
        st = (ctypes.c_double * 4)()
        mx = ctypes.c_float()
        lib().iso_volume_info(self._h, info, st, ctypes.byref(mx))
        return {"node_bbox_min": list(info[0:3]), "node_bbox_max": list(info[3:6]),
                "active_bbox_min": list(info[6:9]), "active_bbox_max": list(info[9:12]),
                "num_leaves": info[12], "scale": st[0], "translation": list(st[1:4]), "max_value": mx.value}

    def __del__(self):
        if getattr(self, "_h", None) and _LIB is not None:
            try:
                _LIB.iso_volume_free(self._h)
            except Exception:
                pass
            self._h = None


def default_params():
    p = IsoParams()
    lib().iso_params_default(ctypes.byref(p))
    return p


def make_params(width, height, origin, lookat=(0, 0, 0), up=(0, 1, 0), fov=45.0, isovalue=0.5,
                last_origin=None, last_lookat=None, viewport=None, ao_samples=0, ao_radius=0.01, **material):
    p = default_params()
    p.width, p.height = int(width), int(height)
    p.fov_deg = float(fov)
    for k in range(3):
        p.origin[k] = float(origin[k]); p.lookat[k] = float(lookat[k]); p.up[k] = float(up[k])
        p.last_origin[k] = float((last_origin if last_origin is not None else origin)[k])
        p.last_lookat[k] = float((last_lookat if last_lookat is not None else lookat)[k])
    p.isovalue = float(isovalue)
    p.ao_samples = int(ao_samples)
    p.ao_radius = float(ao_radius)
    vp = viewport if viewport is not None else (0, 0, width, height)
    for k in range(4):
        p.viewport[k] = int(vp[k])
    for name in ("ambient", "diffuse", "specular"):
        if name in material:
            for k in range(3):
                getattr(p, name)[k] = float(material[name][k])
    if "specular_exponent" in material:
        p.specular_exponent = int(material["specular_exponent"])
    return p


def render(volume, params, threads=0, with_stats=True):
    """Returns (image[H,W,12] float32, stats dict)."""
    out = np.empty((params.height, params.width, 12), dtype=np.float32)
    stats = (ctypes.c_longlong * 4)()
    lib().iso_render(volume._h, ctypes.byref(params), out.ctypes.data, stats if with_stats else None, int(threads))
    return out, {"hits": stats[0], "samples": stats[1], "bricks_touched": stats[2], "steps": stats[3]}


def render_gvdb(volume, params, threads=0):
    """The frame with the CUDA renderer's arithmetic (iso_oracle_gvdb.c); params.isovalue is absolute.  [H,W,12]."""
    out = np.empty((params.height, params.width, 12), dtype=np.float32)
    lib().iso_render_gvdb(volume._h, ctypes.byref(params), out.ctypes.data, int(threads))
    return out


def num_threads():
    return lib().iso_num_threads()


def ao_tables():
    hemi = np.zeros((512, 4), np.float32)
    rot = np.zeros((16, 4), np.float32)
    lib().iso_ao_tables(hemi.ctypes.data, rot.ctypes.data)
    return hemi, rot
