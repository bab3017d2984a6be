"""GPU tests of the legacy C-ABI through ctypes (run with -m gpu): the call sequences
of the reference's ctypes consumers (python/aule/vulkan.py, tests/test_paged_python.py:
31-118, tests/benchmark_mi300x.py:75-153) with numerics checked against the oracle."""
import ctypes

import numpy as np
import pytest

from util import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def A():
    from aule.hip import Aule
    a = Aule()
    yield a
    a.close()


def test_init_idempotent_and_info(A):
    lib = A._lib
    assert lib.aule_init() == 0 and lib.aule_init() == 0          # src/lib.zig:60-63
    assert lib.aule_get_backend_name() == b"HIP/ROCm"
    assert lib.aule_get_vendor() == 1 and lib.aule_get_gpu_vendor() == 1
    assert lib.aule_is_amd_optimized() == 1 and lib.aule_has_fp16() == 1
    assert lib.aule_get_subgroup_size() == 64 and lib.aule_supports_backward() == 1
    assert lib.aule_set_shader_variant(0) == 0 and lib.aule_get_shader_variant() == 0
    assert lib.aule_has_shader_variant(0) == 1 and lib.aule_has_shader_variant(2) == 0
    assert lib.aule_set_shader_variant(3) == -2
    assert len(A.device_name) > 0
    buf = ctypes.create_string_buffer(4)
    assert lib.aule_get_device_name(buf, 4) == 3 and len(buf.value) == 3   # truncated to len-1


def test_handle_lifecycle_and_errors(A):
    lib = A._lib
    lib.aule_tensor_clear_all()
    assert lib.aule_tensor_count() == 0 and lib.aule_tensor_max() == 1024
    h = lib.aule_tensor_create(1, 2, 3, 4)
    assert h == 1 and lib.aule_tensor_size(h) == 24 and lib.aule_tensor_count() == 1
    data = np.arange(24, dtype=np.float32)
    fp = ctypes.POINTER(ctypes.c_float)
    assert lib.aule_tensor_upload(h, data.ctypes.data_as(fp), 24) == 0
    assert lib.aule_tensor_upload(h, data.ctypes.data_as(fp), 23) == -3      # size mismatch
    assert b"size mismatch" in lib.aule_get_error()
    back = np.zeros(24, np.float32)
    assert lib.aule_tensor_download(h, back.ctypes.data_as(fp), 24) == 0
    assert np.array_equal(back, data)                                          # padded pitch is invisible
    u = np.zeros(24, np.uint32)
    assert lib.aule_tensor_download_u32(h, u.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 24) == 0
    assert np.array_equal(u.view(np.float32), data)
    assert lib.aule_tensor_upload(0, data.ctypes.data_as(fp), 24) == -1
    assert lib.aule_tensor_upload(7, data.ctypes.data_as(fp), 24) == -1
    h2 = lib.aule_tensor_create_u32(1, 1, 1, 8)
    assert h2 == 2
    lib.aule_tensor_destroy(h)
    assert lib.aule_tensor_size(h) == 0 and lib.aule_tensor_count() == 1
    assert lib.aule_tensor_create(1, 1, 1, 1) == 1                             # first free slot is reused
    lib.aule_tensor_clear_all()
    # 1024-slot limit (src/lib.zig:17)
    hs = [lib.aule_tensor_create(1, 1, 1, 1) for _ in range(1024)]
    assert hs[-1] == 1024 and lib.aule_tensor_create(1, 1, 1, 1) == 0
    assert lib.aule_get_error() == b"Max tensors reached"
    lib.aule_tensor_clear_all()
    assert lib.aule_tensor_count() == 0


def test_out_of_scope_stubs_return_minus3(A):
    lib = A._lib
    h = lib.aule_tensor_create(1, 1, 4, 32)
    assert lib.aule_attention_forward_paged(h, h, h, h, 0, 0, 0, -1) == -3
    assert lib.aule_spatial_sort(h, h, h, 0) == -3
    assert lib.aule_attention_forward_gravity(h, h, h, h, 0, 0, h, 0, 4, -1) == -3
    assert lib.aule_attention_forward_gpu(h, h, h, h, h, h, 0, -1) == -3        # RoPE handles
    assert lib.aule_attention_forward_gpu(h, h, h, h, 0, 0, 0, 8) == 0          # sliding window is supported
    assert lib.aule_attention_forward_gpu(h, h, h, 99, 0, 0, 0, -1) == -1       # bad handle
    lib.aule_tensor_destroy(h)


@pytest.mark.parametrize("shape", [(1, 1, 16, 16), (1, 1, 32, 32), (1, 1, 64, 64), (1, 2, 32, 32),
                                   (1, 4, 64, 32), (1, 8, 64, 64), (2, 8, 64, 64)])
def test_zig_test_shapes_forward_vs_ref(A, oracle_mod, shape):
    """tests/test_attention.zig:18-31 shapes, inputs (u*2-1)*0.5, rule max_abs<1e-4 or max_rel<1e-3;
    we hold 1e-5 against the fp64 judge and 1e-4 against the zig-order fp32 restatement."""
    rng = np.random.RandomState(42)
    q, k, v = (((rng.rand(*shape) * 2 - 1) * 0.5).astype(np.float32) for _ in range(3))
    out = A.forward_host(q, k, v, causal=False)
    assert_close(out, oracle_mod.fwd_f64(q, k, v, False)[0], 1e-5, 1e-5, "host fwd")
    assert np.abs(out - oracle_mod.ref_forward(q, k, v, False)).max() < 1e-4
    out_c = A.forward_host(q, k, v, causal=True)
    assert np.abs(out_c - oracle_mod.ref_forward(q, k, v, True)).max() < 1e-4


def test_known_answer_cases_on_gpu(A):
    # KAT-1 (attention_ref.zig:250-298), D=4 exercises the padded-pitch path
    q = np.full((1, 1, 2, 4), 0.5, np.float32)
    v = np.array([[1, 2, 3, 4], [5, 6, 7, 8]], np.float32).reshape(1, 1, 2, 4)
    np.testing.assert_allclose(A.forward_host(q, q, v).reshape(2, 4), [[3, 4, 5, 6]] * 2, atol=1e-3)
    # KAT-2 (tests/test_attention.zig:158-219)
    q = np.full((1, 1, 4, 8), 0.5, np.float32)
    v = np.arange(32, dtype=np.float32).reshape(1, 1, 4, 8)
    np.testing.assert_allclose(A.forward_host(q, q, v).reshape(4, 8), np.tile(np.arange(12, 20), (4, 1)), atol=0.01)
    # KAT-3 (:221-270)
    q = (10.0 * np.eye(8, 8, dtype=np.float32)).reshape(1, 1, 8, 8)
    v = (0.1 * np.arange(8, dtype=np.float32))[:, None].repeat(8, 1).reshape(1, 1, 8, 8)
    assert np.abs(A.forward_host(q, q, v) - v).max() < 0.1
    # stability (:272-325)
    rng = np.random.RandomState(1)
    q, k, v = (rng.uniform(-5, 5, (1, 2, 32, 32)).astype(np.float32) for _ in range(3))
    assert np.isfinite(A.forward_host(q, k, v)).all()


def test_handle_path_gqa_cross_attention(A, oracle_mod):
    """aule_attention_forward_gpu: Hkv from K's shape, Sk from K (tests/test_gqa_unit.py:46-55,
    tests/test_cross_attn.py:54-60, 1e-3 there)."""
    rng = np.random.RandomState(2)
    q = rng.randn(2, 12, 16, 64).astype(np.float32)
    k = rng.randn(2, 2, 32, 64).astype(np.float32)
    v = rng.randn(2, 2, 32, 64).astype(np.float32)
    for causal in (False, True):
        out = A.attention(q, k, v, causal=causal)
        assert_close(out, oracle_mod.fwd_f64(q, k, v, causal)[0], 1e-5, 1e-5, f"gqa causal={causal}")
    with pytest.raises(Exception):
        A.attention(q, k[:, :, :, :32], v[:, :, :, :32])
    # shape mismatch surfaces as -3 "ShapeMismatch"
    qt, kt = A.tensor((1, 2, 8, 32)), A.tensor((1, 2, 8, 64))
    from aule import AuleError
    with pytest.raises(AuleError, match="ShapeMismatch"):
        A.attention_gpu(qt, kt, kt, qt)
    qt.destroy(); kt.destroy()


def test_training_path_host_pointers(A, oracle_mod):
    """aule_attention_forward_with_lse + aule_attention_backward (src/lib.zig:765, :639)."""
    rng = np.random.RandomState(3)
    for shape, causal in (((1, 4, 48, 64), True), ((2, 2, 33, 32), False), ((1, 2, 40, 128), True),
                          ((1, 2, 24, 20), True)):
        q, k, v, do = (rng.randn(*shape).astype(np.float32) for _ in range(4))
        out, lse = A.attention_forward_with_lse(q, k, v, causal=causal)
        ref, ref_lse = oracle_mod.fwd_f64(q, k, v, causal)
        assert_close(out, ref, 1e-5, 1e-5, "out")
        assert_close(lse, ref_lse, 1e-5, 1e-5, "lse")
        dq, dk, dv = A.attention_backward(q, k, v, out, do, lse, causal=causal)
        rq, rk, rv = oracle_mod.bwd_f64(q, k, v, do, causal)
        for name, a, b in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
            assert_close(a, b, 3e-5 * max(1.0, float(np.abs(b).max())), 3e-5, name)


def test_ex_rejects_bad_arguments(A):
    from aule import _capi
    lib = A._lib
    d = _capi.AttnDesc()
    assert lib.aule_attention_forward_ex(ctypes.byref(d)) == -3                # struct_size 0
    d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype, d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = 2, 1, 3, 2, 8, 8, 64
    assert lib.aule_attention_forward_ex(ctypes.byref(d)) == -3                # Hq % Hkv
    d.heads_q = 4
    d.head_dim = 48
    assert lib.aule_attention_forward_ex(ctypes.byref(d)) == -3                # head_dim
    d.head_dim = 64
    d.window_size = 4                                                          # accepted (sliding window)
    assert lib.aule_attention_forward_ex(ctypes.byref(d)) == -3                # null pointers
    assert b"null" in lib.aule_get_error()


def test_library_loaded_before_torch_still_finds_the_device():
    """ONE HIP runtime per process: PyTorch-ROCm wheels bundle their own libamdhip64, libaule.so links the system's.  Loading
    libaule.so first used to leave both in the process -- torch's took the device and aule_init() failed with "no ROCm-capable
    device" (build() followed by smoke() in one interpreter).  _capi.load() now loads torch's runtime first when torch is installed
    but not imported yet; this runs that order in a fresh interpreter and counts the runtimes mapped."""
    import subprocess
    import sys
    from conftest import ROOT
    prog = r'''
import os, sys
sys.path.insert(0, os.path.join(sys.argv[1], "aule-attention_amd"))
from aule import _capi
assert "torch" not in sys.modules
lib = _capi.load()
import torch
assert torch.cuda.is_available()
_capi.get_lib()          # aule_init()
hip = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
assert len(hip) == 1, hip
import aule
q = torch.randn(1, 2, 128, 64, device="cuda", dtype=torch.bfloat16)
o = aule.flash_attention(q, q, q, causal=True)
torch.cuda.synchronize()
assert torch.isfinite(o.float()).all()
print("ORDER_OK", hip[0])
'''
    r = subprocess.run([sys.executable, "-c", prog, ROOT], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0 and "ORDER_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-1500:]
