"""Raw volume import (the ``.dat`` descriptor + binary object file pairs of the reference's
``CPURenderer/ExternalImporter.cpp:25-232``) and conversion to ``.vbx`` for ``loadGrid``.

Descriptor lines: ``ObjectFileName: <file>``, ``Resolution: X Y Z``, ``Format: UCHAR|BYTE|USHORT``.
The object file may carry a header; the payload is the LAST X*Y*Z entries (``:95-106``).  Values are
normalised to [0,1] (``/255`` or ``/65535``), optionally box-downsampled, and everything below
``lower_threshold`` becomes 0 (``:154``) -- the sparsity the bricked renderer relies on.
Returned layout: float32 ``[z][y][x]``.
"""
import os

import numpy as np

_FORMATS = {"UCHAR": (np.uint8, 255.0), "BYTE": (np.uint8, 255.0), "USHORT": (np.dtype("<u2"), 65535.0)}


def import_raw(dat_path, downsampling=1, lower_threshold=0.02):
    if not dat_path.endswith(".dat"):
        raise ValueError("Filename does not point to the .dat file")
    obj, res, fmt = None, None, None
    with open(dat_path) as f:
        for line in f:
            tok = line.split()
            if not tok:
                continue
            if tok[0] == "ObjectFileName:":
                obj = tok[1]
            elif tok[0] == "Resolution:":
                res = tuple(int(v) for v in tok[1:4])
            elif tok[0] == "Format:":
                fmt = tok[1]
    if obj is None or res is None or fmt is None:
        raise ValueError("Descriptor file does not contain ObjectFileName, Resolution and Format")
    if fmt not in _FORMATS:
        raise ValueError("Unknown format " + fmt)
    dtype, scale = _FORMATS[fmt]
    nx, ny, nz = res
    count = nx * ny * nz
    path = os.path.join(os.path.dirname(dat_path), obj)
    size = os.path.getsize(path)
    header = size - count * np.dtype(dtype).itemsize
    if header < 0:
        raise ValueError("File is too small, %d bytes missing" % (-header))
    raw = np.fromfile(path, dtype=dtype, count=count, offset=header).reshape(nz, ny, nx)
    vol = raw.astype(np.float32) / np.float32(scale)
    d = int(downsampling)
    if d > 1:
        vol = vol[:nz // d * d, :ny // d * d, :nx // d * d].reshape(nz // d, d, ny // d, d, nx // d, d).mean(axis=(1, 3, 5))
    vol[vol < lower_threshold] = 0.0
    return np.ascontiguousarray(vol, dtype=np.float32)


def export_raw(dat_path, volume, fmt="UCHAR"):
    """Inverse of ``import_raw`` (tests, data exchange)."""
    dtype, scale = _FORMATS[fmt]
    nz, ny, nx = volume.shape
    obj = os.path.basename(dat_path)[:-4] + ".raw"
    np.clip(np.rint(volume * scale), 0, scale).astype(dtype).tofile(os.path.join(os.path.dirname(dat_path), obj))
    with open(dat_path, "w") as f:
        f.write("ObjectFileName: %s\nResolution: %d %d %d\nFormat: %s\n" % (obj, nx, ny, nz, fmt))


def convert_to_vbx(dat_path, vbx_path, downsampling=1, lower_threshold=0.02):
    """``Vdb2Vbx``-style conversion (``GPURendererDirect/Vdb2Vbx.cpp:70-324``) for raw inputs."""
    from . import vbx
    vol = import_raw(dat_path, downsampling, lower_threshold)
    return vbx.write_vbx(vbx_path, vol)
