"""GPU confirmation of the lane maps assumed by the kernels and by the host
emulator (tests/emu/wave_ops.h): MFMA 32x32x16 bf16 fragments, ds_read_b128 row
fragments and ds_read_b64_tr_b16 transposed fragments of the swizzled tile."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe():
    import torch  # noqa: F401  (maps torch's HIP runtime first)
    lib = C.CDLL(os.path.join(ROOT, "tests", "probe", "libprobe.so"))
    lib.probe_run_mfma.argtypes = [C.c_void_p] * 4
    lib.probe_run_frags.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]
    return lib


def test_mfma_32x32x16_layout(probe):
    import torch
    g = torch.Generator().manual_seed(3)
    A = torch.randn(32, 16, generator=g).to(torch.bfloat16).cuda()
    B = torch.randn(16, 32, generator=g).to(torch.bfloat16).cuda()  # asymmetric on purpose
    Cm = torch.zeros(32, 32, dtype=torch.float32, device="cuda")
    assert probe.probe_run_mfma(A.data_ptr(), B.data_ptr(), Cm.data_ptr(), None) == 0
    torch.cuda.synchronize()
    ref = A.float().cpu().double() @ B.float().cpu().double()
    err = (Cm.cpu().double() - ref).abs().max().item()
    assert err < 1e-4, f"MFMA fragment map differs from the documented one (max err {err})"


@pytest.mark.parametrize("row0,step,row0t,d0", [(0, 0, 0, 0), (32, 5, 16, 32), (32, 7, 48, 96), (0, 3, 32, 64)])
def test_lds_fragments(probe, row0, step, row0t, d0):
    import torch
    T = torch.arange(64 * 128, dtype=torch.float32).reshape(64, 128)
    T = (T % 251 - 125).to(torch.bfloat16)  # exactly representable, position-revealing
    Td = T.cuda()
    rows = torch.zeros(64, 8, dtype=torch.bfloat16, device="cuda")
    cols = torch.zeros(64, 8, dtype=torch.bfloat16, device="cuda")
    rc = probe.probe_run_frags(Td.data_ptr(), rows.data_ptr(), cols.data_ptr(), row0, step, row0t, d0, None)
    assert rc == 0
    torch.cuda.synchronize()
    Tf = T.float().numpy()
    exp_rows = np.zeros((64, 8), np.float32)
    exp_cols = np.zeros((64, 8), np.float32)
    for l in range(64):
        l31, hi = l & 31, l >> 5
        for j in range(8):
            exp_rows[l, j] = Tf[row0 + l31, 16 * step + 8 * hi + j]
            exp_cols[l, j] = Tf[row0t + 4 * hi + (j & 3) + 8 * (j >> 2), d0 + l31]
    assert np.array_equal(rows.float().cpu().numpy(), exp_rows), "ds_read_b128 row fragment map"
    assert np.array_equal(cols.float().cpu().numpy(), exp_cols), "ds_read_b64_tr_b16 fragment map"


def test_mfma_f32_is_ordered_fma_chain(probe):
    """v_mfma_f32_32x32x2_f32 == fmaf(a[k1], b[k1], fmaf(a[k0], b[k0], c)), bitwise.
    The VQGAN kernels' bit-exact contract (oracle/vqgan_ref.c) and the host
    emulator both rest on this.  Data with large cancellations makes any other
    association or a wider internal accumulator visible."""
    import torch
    probe.probe_run_mfma_f32.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
    K = 64
    g = np.random.default_rng(11)
    A = (g.standard_normal((32, K)) * 10.0 ** g.integers(-3, 4, (32, K))).astype(np.float32)
    B = (g.standard_normal((K, 32)) * 10.0 ** g.integers(-3, 4, (K, 32))).astype(np.float32)
    C0 = g.standard_normal((32, 32)).astype(np.float32)
    Ad, Bd, Cd = (torch.from_numpy(t).cuda() for t in (A, B, C0))
    out = torch.zeros(32, 32, dtype=torch.float32, device="cuda")
    assert probe.probe_run_mfma_f32(Ad.data_ptr(), Bd.data_ptr(), Cd.data_ptr(), out.data_ptr(), K, None) == 0
    torch.cuda.synchronize()
    # float32 fma chain on the host through libm's correctly rounded fmaf
    libm = C.CDLL("libm.so.6")
    libm.fmaf.restype = C.c_float
    libm.fmaf.argtypes = [C.c_float] * 3
    ref = C0.copy()
    for i in range(32):
        for j in range(32):
            c = float(C0[i, j])
            for k in range(K):
                c = libm.fmaf(float(A[i, k]), float(B[k, j]), c)
            ref[i, j] = c
    got = out.cpu().numpy()
    assert np.array_equal(got, ref), f"max diff {np.abs(got - ref).max()}"


def test_plain_c_program_runs_the_forward():
    """examples/attn_fwd_from_c.c: C99 + HIP runtime + liblwm_hip.so, no Python in the call path."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "attn_fwd_from_c")
    if not os.path.exists(exe):
        so = os.path.join(root, "lwm_amd", "liblwm_hip.so")
        subprocess.run(["gcc", "-std=c99", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(root, "include"), "-I",
                        "/opt/rocm/include", os.path.join(root, "examples", "attn_fwd_from_c.c"), so, "-L",
                        "/opt/rocm/lib", "-lamdhip64", "-lm", f"-Wl,-rpath,{os.path.dirname(so)}", "-o", exe], check=True)
    import torch
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = os.path.join(os.path.dirname(torch.__file__), "lib") + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    r = subprocess.run([exe], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0 and r.stdout.startswith("ok"), (r.returncode, r.stdout, r.stderr)


def test_xcc_id_register_and_persistent_grid_placement(probe):
    """The attention kernels map all tiles of one (batch, head) onto one XCD by block index modulo 8, so that its K/V
    (or Q/dO) stream is shared in that XCD's L2.  s_getreg(HW_REG_XCC_ID) must report all 8 XCDs of an MI355X, and a
    grid of one workgroup per CU (131 KB of LDS each, so one per CU) must put workgroups on every XCD, 32 each -- the
    kernels are correct on any placement, their speed assumes this one."""
    import torch
    n = torch.cuda.get_device_properties(0).multi_processor_count
    out = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    per = torch.zeros(16, dtype=torch.int32, device="cuda")
    probe.probe_run_xcc.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    assert probe.probe_run_xcc(out.data_ptr(), per.data_ptr(), n, 131 * 1024, None) == 0
    torch.cuda.synchronize()
    ids, counts = out.cpu().tolist(), per.cpu().tolist()
    print("xcc id of blocks 0..15:", ids[:16], "workgroups per xcc:", counts)
    assert sorted(set(ids)) == list(range(8)), sorted(set(ids))
    assert sum(counts) == n and max(counts[:8]) - min(counts[:8]) <= 2, counts
