"""Locates, builds and loads the in-tree HIP shared libraries.

The libraries are plain C-ABI (see ``include/*.h``) and are bound with ctypes.  They live in
``isosurfacesuperresolution_amd/lib`` so that they travel with the source tree; nothing is
installed into site-packages.  There is no CPU fallback: if a library is missing and cannot be
built, loading raises.
"""
import ctypes
import os
import subprocess

_PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_PKG, "csrc")
LIBDIR = os.path.join(_PKG, "lib")
# ISR_RENDERER_LIB: another build of the renderer (A/B measurements of ray-marcher changes on one box)
RENDERER_LIB = os.environ.get("ISR_RENDERER_LIB") or os.path.join(LIBDIR, "libGPURendererDirect.so")
# The PRODUCT build of the super-resolution kernels (csrc/sr_diag.h: no isrDebug* export, no ablation switch in a kernel, none of the
# experimental kernel forms) and the DIAGNOSTICS build of the same sources (`make diag`: what tools/ and the timeout-path / form-parity
# tests load -- ops.diagnostics_library()).
SR_DIAG_LIB = os.path.join(LIBDIR, "libisr_sr_diag.so")
# ISR_SR_LIB: another build of the same library (A/B measurements of kernel changes on one box: tools/ab_bench.sh);
# ISR_SR_DIAG=1: the whole process on the diagnostics build (the scripts under tools/ that flip isrDebug* switches)
SR_LIB = os.environ.get("ISR_SR_LIB") or (SR_DIAG_LIB if os.environ.get("ISR_SR_DIAG") == "1" else os.path.join(LIBDIR, "libisr_sr.so"))

_cache = {}


def build(force=False, verbose=False):
    """Compile every HIP extension for gfx950 (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-j8", "-C", CSRC, "all", "diag"]
    if force:
        cmd.insert(1, "-B")
    out = None if verbose else subprocess.DEVNULL
    subprocess.check_call(cmd, stdout=out)
    return [os.path.join(LIBDIR, "libGPURendererDirect.so"), os.path.join(LIBDIR, "libisr_sr.so"), SR_DIAG_LIB]


def _preload_hip_runtime():
    # When torch is in the process its bundled libamdhip64 (SONAME libamdhip64.so.7) must be the
    # one our libraries bind to, otherwise two HIP runtimes would own separate contexts.
    try:
        import torch  # noqa: F401  (import order is the point)
    except ImportError:
        pass


def load(path):
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        build()
    if not os.path.exists(path):
        raise RuntimeError("native library %s is missing and could not be built" % path)
    _preload_hip_runtime()
    lib = ctypes.CDLL(path)
    _cache[path] = lib
    return lib
