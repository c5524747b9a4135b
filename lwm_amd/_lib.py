"""Loader of liblwm_hip.so -- the hand-written HIP kernels behind the C ABI.

There is NO fallback: if the shared library is missing or does not export the
full ABI this raises, and every op in lwm_amd raises with it.
"""
import ctypes as C
import os

from . import _capi

_HERE = os.path.dirname(os.path.abspath(__file__))
# LWM_HIP_LIB: another build of the same ABI (kernel A/B measurements, scripts/ab_build.sh)
LIB_PATH = os.environ.get("LWM_HIP_LIB") or os.path.join(_HERE, "liblwm_hip.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; "
                "g.build()'` (hipcc --offload-arch=gfx950). lwm_amd has no CPU/PyTorch fallback.")
        # torch must be imported first so that the HIP runtime already mapped in
        # the process (torch/lib/libamdhip64.so, soname libamdhip64.so.7) is the
        # one this library binds to; two HIP runtimes cannot share streams.
        import torch  # noqa: F401
        _lib = _capi.bind(C.CDLL(LIB_PATH))
    return _lib
