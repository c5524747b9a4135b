"""INTEGRATION.md's stand-alone ctypes stubs are executable documentation: a maintainer copies them
into the reference tree.  The text is exec'd here against the built library -- the struct mirrors must
have the library's layout (lwm_sizeof) and, on a GPU, the stub's forward must agree with the package's
own binding bit for bit (same kernel, same arguments)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _python_blocks(section):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    body = text.split(f"\n## {section}", 1)[1].split("\n## ", 1)[0]
    return re.findall(r"```python\n(.*?)```", body, flags=re.S)


def _exec_stubs():
    cwd = os.getcwd()
    os.chdir(ROOT)                      # the stub loads "lwm_amd/liblwm_hip.so" relative to the checkout
    try:
        ns = {}
        exec(compile(_python_blocks("3.")[0], "INTEGRATION.md#3", "exec"), ns)
        conv = [b for b in _python_blocks("4.") if "class ConvArgs" in b][0]
        exec(compile(conv, "INTEGRATION.md#4", "exec"), ns)
        return ns
    finally:
        os.chdir(cwd)


def test_stub_struct_mirrors_match_the_library():
    import ctypes as C
    from lwm_amd import _capi
    ns = _exec_stubs()                  # raises ImportError if lwm_sizeof disagrees
    assert C.sizeof(ns["AttnArgs"]) == C.sizeof(_capi.LwmAttnArgs)
    assert [f[0] for f in ns["AttnArgs"]._fields_] == [f[0] for f in _capi.LwmAttnArgs._fields_]
    assert C.sizeof(ns["ConvArgs"]) == C.sizeof(_capi.LwmConvArgs)
    # a mirror that stops at k_splits (the round-1 text) must be caught by the same check
    short = type("Short", (C.Structure,), {"_fields_": ns["AttnArgs"]._fields_[:-2]})
    assert C.sizeof(short) != ns["_lib"].lwm_sizeof(0)


@pytest.mark.gpu
def test_stub_forward_and_conv_agree_with_the_package():
    import torch
    from lwm_amd import ops
    ns = _exec_stubs()
    g = torch.Generator().manual_seed(3)
    q, k, v = (torch.randn(1, 320, 2, 128, generator=g).to(torch.bfloat16).cuda() for _ in range(3))
    seg = (torch.arange(320) // 100).to(torch.int32)[None].cuda()
    out, lse = ns["attn_fwd_one_block"](q, k, v, 0, 0, seg_q=seg, seg_k=seg)
    ro, rl = ops.attn_fwd_block(q, k, v, causal=True, seg_q=seg, seg_k=seg)
    torch.cuda.synchronize()
    assert torch.equal(out, ro) and torch.equal(lse, rl)
    # the float32 flavour of the stub == the package's dispatch on the operands' dtype
    of, lf = ns["attn_fwd_one_block_f32"](q.float(), k.float(), v.float(), 0, 0)
    rof, rlf = ops.attn_fwd_block(q.float(), k.float(), v.float(), causal=True)
    torch.cuda.synchronize()
    assert of.dtype == torch.float32 and torch.equal(of, rof) and torch.equal(lf, rlf)
    x = torch.randn(1, 16, 16, 32, generator=g).cuda()
    w = torch.randn(3, 3, 32, 64, generator=g).cuda() / 17
    b = torch.randn(64, generator=g).cuda()
    assert torch.equal(ns["conv3x3_same"](x, w, b), ops.conv2d_nhwc(x, w, b))
    from lwm_amd.llama_ops import gemv
    xd = torch.randn(2, 4096, generator=g).to(torch.bfloat16).cuda()
    kd = (torch.randn(4096, 4096, generator=g) * 0.02).to(torch.bfloat16).cuda()
    assert torch.equal(ns["dense_decode"](xd, kd), gemv(xd, kd))
