"""The host side of libaule.so -- C-ABI entry points, handle table, dispatch rules, launch plans -- under AddressSanitizer
and UndefinedBehaviorSanitizer (SURVEY.md section 5 aux plan: the reference has no sanitizer job; its Zig debug builds
trap on UB).  `make san` builds build/variants/libaule_san.so (host code instrumented, device code plain); the no-GPU
contract and a sweep of the pure host logic then run in a subprocess with the ASan runtime preloaded.  Any report fails
the run (-fno-sanitize-recover, ASAN_OPTIONS=halt_on_error=1)."""
import glob
import os
import subprocess
import sys

import pytest

from conftest import ROOT

SAN_LIB = os.path.join(ROOT, "build", "variants", "libaule_san.so")
CSRC = os.path.join(ROOT, "aule-attention_amd", "csrc")

CHILD = r'''
import ctypes, itertools, os, sys
sys.path.insert(0, os.path.join(%(root)r, "aule-attention_amd"))
from aule import _capi
lib = _capi.load()
assert _capi.library_path().endswith("libaule_san.so"), _capi.library_path()
has_gpu = os.path.exists("/dev/kfd")
# --- uninitialised contract (on a box without a GPU aule_init itself fails)
if not has_gpu:
    assert lib.aule_init() == -1 and b"Failed to initialize backend" in lib.aule_get_error()
    assert lib.aule_tensor_create(1, 1, 1, 1) == 0
    buf = (ctypes.c_float * 16)()
    assert lib.aule_attention_forward(buf, buf, buf, buf, 1, 1, 2, 2, 0) == -1
# --- handle-table misuse never touches memory it does not own
for h in (0, 1, 1023, 1024, 2 ** 31, 2 ** 63 + 5):
    lib.aule_tensor_destroy(h)
    assert lib.aule_tensor_size(h) == 0
assert lib.aule_tensor_count() == 0 and lib.aule_tensor_max() == 1024
lib.aule_tensor_clear_all()
# --- descriptors: wrong sizes, null, every causal code, zero / huge extents
lib.aule_hip_debug_forward_route.restype = ctypes.c_int32
lib.aule_hip_debug_forward_route.argtypes = [ctypes.POINTER(_capi.AttnDesc)]
assert lib.aule_hip_debug_forward_route(None) == -3
d = _capi.AttnDesc()
assert lib.aule_hip_debug_forward_route(ctypes.byref(d)) == -3          # struct_size 0
assert lib.aule_attention_forward_ex(ctypes.byref(d)) in (-1, -3)
routes = {}
n = 0
for dtype, B, Hq, g, Sq, Sk, D, causal, W in itertools.product(
        (0, 1, 2), (1, 4, 64), (1, 8, 32), (1, 4, 8), (1, 7, 64, 256, 300, 4096), (1, 64, 192, 193, 777, 8192, 131072),
        (32, 64, 128), (0, 1, 2), (-1, 5, 100000)):
    if Hq %% g:
        continue
    d = _capi.AttnDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = dtype
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hq // g, Sq, Sk, D
    d.causal, d.window_size, d.scale = causal, W, 0.0
    r = lib.aule_hip_debug_forward_route(ctypes.byref(d))
    routes[r] = routes.get(r, 0) + 1
    # the launch plans of the two-launch paths (dry runs: no device work, no allocation)
    ws = lib.aule_attention_forward_workspace_size(ctypes.byref(d))
    assert ws >= 0 and (ws == 0 or r in (0, 4, 5, 7)), (r, ws)   # (0: the fp32 forward's key-range pieces on small grids, round 5)
    if r == 0 and ws:
        assert ws %% (B * Hq * Sq * (D + 4) * 4) == 0 and 2 <= ws // (B * Hq * Sq * (D + 4) * 4) <= 8, (ws, B, Hq, Sq, D)
    # the causal-split plan (route 7) into a buffer of exactly the size it asks for, and into one that is too small
    need = lib.aule_hip_debug_forward_split_plan(ctypes.byref(d), None, 0)
    assert (need < 0) == (r == 7), (r, need)
    if r == 7:
        buf = (ctypes.c_int32 * (-need))()
        assert lib.aule_hip_debug_forward_split_plan(ctypes.byref(d), buf, -need) == -need
        assert lib.aule_hip_debug_forward_split_plan(ctypes.byref(d), buf, -need - 1) == need
        assert ws == buf[0] * B * Hq * Sq * (D + 4) * 4, (ws, buf[0])
    # the fused-rotation rule: host logic, the table pointers are never dereferenced
    rp = _capi.AttnRope()
    rp.struct_size = ctypes.sizeof(_capi.AttnRope)
    rp.table_len, rp.table_pitch, rp.q_pos_offset, rp.cos, rp.sin = max(Sq, Sk) + 1, D // 2, (Sk - Sq if causal == 2 and Sk >= Sq else 0), 0x10000, 0x20000
    f = lib.aule_attention_forward_rope_fusable(ctypes.byref(d), ctypes.byref(rp))
    assert f in (0, 1) and (f == 0 or (dtype in (1, 2) and D in (64, 128) and (W == -1 or W >= Sq)))   # (a window >= Sq masks nothing)
    n += 1
assert {0, 1, 4, 5, 7, 8} <= set(routes) and 6 not in routes, routes   # (6: the retired two-waves-per-SIMD stream)
b = _capi.AttnBwdDesc()
b.struct_size = ctypes.sizeof(_capi.AttnBwdDesc)
for B, Hq, Hkv, S, D, causal in ((1, 1, 1, 1, 32, 0), (4, 32, 8, 2048, 128, 1), (2, 64, 1, 8192, 64, 1), (64, 32, 32, 300, 128, 0)):
    b.dtype = 2
    b.batch, b.heads_q, b.heads_kv, b.seq_q, b.seq_k, b.head_dim, b.causal = B, Hq, Hkv, S, S, D, causal
    assert lib.aule_attention_backward_workspace_size(ctypes.byref(b)) >= 4 * B * Hq * S
print("SANITIZED-OK", n, sorted(routes.items()))
'''


def _asan_runtime():
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    return hits[-1] if hits else None


def _sources_newer_than(lib):
    t = os.path.getmtime(lib)
    for pat in ("*.hip", "*.cpp", "*.h", "Makefile"):
        for f in glob.glob(os.path.join(CSRC, pat)):
            if os.path.getmtime(f) > t:
                return True
    return os.path.getmtime(os.path.join(ROOT, "include", "aule.h")) > t


def test_host_code_under_asan_and_ubsan():
    rt = _asan_runtime()
    if rt is None or not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no clang ASan runtime / hipcc in this image")
    if not os.path.exists(SAN_LIB) or _sources_newer_than(SAN_LIB):
        r = subprocess.run(["make", "-C", CSRC, "san", "-j8"], capture_output=True, text=True, timeout=1500)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    env = dict(os.environ)
    env.update({"AULE_LIBRARY_PATH": SAN_LIB, "LD_PRELOAD": rt,
                "ASAN_OPTIONS": "detect_leaks=0:halt_on_error=1:abort_on_error=0:verify_asan_link_order=0",
                "UBSAN_OPTIONS": "halt_on_error=1:print_stacktrace=1"})
    env.pop("AULE_BACKEND", None)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "SANITIZED-OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]


def test_aule_backend_env(monkeypatch):
    """AULE_BACKEND (src/backends/backend.zig:86-100): 'hip' is a no-op for a library that IS the HIP backend, unknown values
    fall through to auto-detection like in the reference, 'vulkan' / 'cpu' are refused loudly (this build has neither)."""
    sys.path.insert(0, os.path.join(ROOT, "aule-attention_amd"))
    from aule import _capi
    lib = _capi.load()
    lib.aule_shutdown()
    for name in ("vulkan", "cpu"):
        monkeypatch.setenv("AULE_BACKEND", name)
        assert lib.aule_init() == -1
        msg = lib.aule_get_error()
        assert b"AULE_BACKEND=" + name.encode() in msg and b"HIP" in msg, msg
    if not os.path.exists("/dev/kfd"):
        for name in ("hip", "auto", "something-else"):
            monkeypatch.setenv("AULE_BACKEND", name)
            assert lib.aule_init() == -1
            assert b"no HIP device" in lib.aule_get_error()    # went on to the device probe: the variable was a no-op
