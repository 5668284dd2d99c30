"""Locates, (re)builds and loads librmn_ez_hip.so.  In-tree only: the .so lives next to this file."""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("EZHIP_LIBRARY") or os.path.join(_HERE, "librmn_ez_hip.so")   # EZHIP_LIBRARY: development builds
_lib = None


def library_path():
    return _SO


def build_library(force=False):
    """hipcc --offload-arch=gfx950 build of csrc/ (cross-compiles without a GPU)."""
    cmd = ["make", "-s", "-C", os.path.join(_HERE, "csrc")]
    if force:
        subprocess.check_call(cmd + ["clean"])
    subprocess.check_call(cmd)
    return _SO


def _share_torch_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, requested by
    torch as plain "libamdhip64.so").  If this library bound the system copy and torch later loaded
    its bundled one, the process would hold TWO HIP runtimes: streams and device pointers could not
    be shared and the second runtime to initialise sees no device.  Pre-loading torch's copy (when
    torch is installed) makes both resolve to one runtime, whatever the import order.  Pure C
    callers without torch simply get the system runtime through the library's RUNPATH."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec and spec.submodule_search_locations:
            cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
            if os.path.exists(cand):
                ctypes.CDLL(cand, mode=ctypes.RTLD_GLOBAL)
    except Exception:   # noqa: BLE001 -- best effort; the system runtime is the fallback
        pass


def load_library():
    """Returns the ctypes handle of librmn_ez_hip.so; raises (loudly) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise OSError(f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(there is no CPU fallback for the MI355X hot path)")
        _share_torch_hip_runtime()
        _lib = ctypes.CDLL(_SO)
    return _lib
