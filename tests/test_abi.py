"""liblwm_hip.so loads and exports every symbol include/lwm_hip.h declares, and
argument validation fails loudly -- no kernels are launched (runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

from lwm_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "lwm_amd", "liblwm_hip.so")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(SO):
        import __graft_entry__ as g
        g.build()
    return _capi.bind(C.CDLL(SO))


def test_every_declared_symbol_is_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "lwm_hip.h")).read()
    declared = set(re.findall(r"\b(lwm_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    assert declared == set(_capi.PROTOTYPES), (declared ^ set(_capi.PROTOTYPES))
    for name in declared:
        assert hasattr(lib, name), name
    assert lib.lwm_version() >= 100


def test_validation_errors(lib):
    a = _capi.LwmAttnArgs()
    assert lib.lwm_attn_fwd(None, None) == _capi.LWM_EINVAL
    a.D, a.B, a.H, a.Sq, a.Sk = 64, 1, 1, 16, 16
    assert lib.lwm_attn_fwd(C.byref(a), None) == _capi.LWM_EUNSUPPORTED
    assert b"head_dim" in lib.lwm_last_error()
    a.D = 128
    a.scale = 0.1
    assert lib.lwm_attn_fwd(C.byref(a), None) == _capi.LWM_EINVAL   # q is null
    assert b"q" in lib.lwm_last_error()
    buf = (C.c_char * 4096)()
    base = C.addressof(buf)
    base += (-base) % 16
    a.q = _capi.LwmTensor4(base + 2, 128, 128, 128)                  # misaligned
    assert lib.lwm_attn_fwd(C.byref(a), None) == _capi.LWM_EINVAL
    a.q = a.k = a.v = _capi.LwmTensor4(base, 128, 128, 128)
    a.segment_ids_q = base                                           # seg_q without seg_k
    assert lib.lwm_attn_fwd(C.byref(a), None) == _capi.LWM_EINVAL
    a.segment_ids_q = None
    a.final_out = 1                                                  # out / lse missing
    assert lib.lwm_attn_fwd(C.byref(a), None) == _capi.LWM_EINVAL
    assert lib.lwm_attn_bwd_dq(C.byref(a), None) == _capi.LWM_EINVAL
    assert lib.lwm_cast_f32_to_bf16(None, None, 8, None) == _capi.LWM_EINVAL
    assert lib.lwm_cast_f32_to_bf16(None, None, 0, None) == _capi.LWM_OK
    import ctypes as _C
    one = (_C.c_void_p * 1)(None)
    assert lib.lwm_sum_f32_to_bf16(one, 0, None, 8, None) == _capi.LWM_EINVAL      # n_src out of range
    assert lib.lwm_sum_f32_to_bf16(one, 1, None, 8, None) == _capi.LWM_EINVAL      # null pointers
    assert lib.lwm_sum_f32_to_bf16(one, 1, None, 0, None) == _capi.LWM_OK


def test_product_has_no_cpu_fallback():
    """ops refuse CPU tensors; nothing under lwm_amd imports the oracle or the emulator."""
    import torch
    from lwm_amd import ops
    x = torch.zeros(1, 8, 1, 128, dtype=torch.bfloat16)
    with pytest.raises(ValueError):
        ops.attn_fwd_block(x, x, x)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "lwm_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".inc")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|tests)\b", src, re.M), f
                assert "emu/" not in src or f in ("wave_ops.h", "api.inc", "vqgan_api.inc"), f


def test_header_is_plain_c_and_links_from_a_c_program(tmp_path):
    """The boundary is a C ABI: include/lwm_hip.h must compile as C99 (no C++ in the signatures) and a
    plain C program linked against the shared library must see the same struct sizes the library was
    built with, get error codes (not exceptions) for bad arguments, and read the error string."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    if not os.path.exists(SO):
        import __graft_entry__ as g
        g.build()
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "lwm_hip.h"
int main(void) {
    LwmAttnArgs a;
    LwmConvArgs c;
    memset(&a, 0, sizeof a);
    memset(&c, 0, sizeof c);
    if (lwm_version() < 111) return 1;
    if (lwm_sizeof(0) != (int)sizeof(LwmAttnArgs)) return 2;
    if (lwm_sizeof(1) != (int)sizeof(LwmConvArgs)) return 3;
    if (lwm_attn_fwd(NULL, NULL) >= 0) return 4;
    a.D = 64; a.B = 1; a.H = 1; a.Sq = 16; a.Sk = 16;
    if (lwm_attn_fwd(&a, NULL) >= 0) return 5;
    if (strstr(lwm_last_error(), "head_dim") == NULL) return 6;
    if (lwm_conv2d_nhwc_f32(&c, NULL) >= 0) return 7;
    printf("ok %d\n", lwm_version());
    return 0;
}
''')
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-pedantic", "-I", os.path.join(ROOT, "include"),
                    str(src), "-o", str(exe), SO, f"-Wl,-rpath,{os.path.dirname(SO)}"], check=True)
    env = dict(os.environ)
    import torch
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([str(exe)], capture_output=True, text=True, env=env)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.returncode, r.stdout, r.stderr)


def test_c_example_builds():
    """examples/attn_fwd_from_c.c -- the hot path called from plain C with only the HIP runtime -- compiles and
    links against the header and the library (it needs a GPU to run: tests/test_gpu_probe.py runs it)."""
    import shutil
    import subprocess
    if shutil.which("gcc") is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("no gcc / ROCm headers")
    exe = os.path.join(ROOT, "examples", "attn_fwd_from_c")
    subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"),
                    "-I", "/opt/rocm/include", os.path.join(ROOT, "examples", "attn_fwd_from_c.c"), SO,
                    "-L", "/opt/rocm/lib", "-lamdhip64", "-lm", f"-Wl,-rpath,{os.path.dirname(SO)}", "-o", exe],
                   check=True)
    assert os.path.exists(exe)
