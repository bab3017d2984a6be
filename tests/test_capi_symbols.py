"""CPU tests of the drop-in boundary: libaule.so loads, exports every symbol that
include/aule.h declares, and behaves per the reference contract when NOT initialised
(no compute calls are made here; the GPU tests do that)."""
import ctypes
import itertools
import os
import re

import pytest

from conftest import ROOT

import aule
from aule import _capi


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "aule.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(aule_[a-z0-9_]+)\s*\(", txt)))


def test_header_lists_reference_abi():
    """The 31 exports of src/lib.zig must all be declared."""
    ref = """aule_init aule_supports_backward aule_shutdown aule_get_error aule_get_backend_name aule_get_vendor
    aule_is_amd_optimized aule_has_fp16 aule_set_shader_variant aule_get_shader_variant aule_has_shader_variant
    aule_get_device_name aule_get_gpu_vendor aule_get_subgroup_size aule_attention_forward aule_tensor_count
    aule_tensor_max aule_tensor_clear_all aule_tensor_create aule_tensor_create_u32 aule_tensor_destroy
    aule_tensor_upload aule_tensor_download aule_tensor_download_u32 aule_attention_forward_gpu
    aule_attention_forward_paged aule_spatial_sort aule_attention_forward_gravity aule_tensor_size
    aule_attention_backward aule_attention_forward_with_lse""".split()
    assert len(ref) == 31
    missing = set(ref) - set(header_symbols())
    assert not missing, missing


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_capi.find_library())
    for name in header_symbols():
        assert hasattr(lib, name), f"libaule.so does not export {name}"


def test_binding_covers_header():
    bound = {s[0] for s in _capi.SIGNATURES}
    assert bound == set(header_symbols())
    _capi.load()   # declares argtypes/restypes for all of them


def test_descriptor_layout():
    # x86-64 SysV layout of the structs in include/aule.h
    # abi2: optional workspace / workspace_bytes appended to the forward and paged descriptors
    assert ctypes.sizeof(_capi.AttnDesc) == 112
    assert _capi.AttnDesc.stream.offset == 48 and _capi.AttnDesc.lse.offset == 88
    assert _capi.AttnDesc.workspace.offset == 96 and _capi.AttnDesc.workspace_bytes.offset == 104
    assert ctypes.sizeof(_capi.PagedDesc) == 120 and _capi.PagedDesc.workspace.offset == 104
    assert ctypes.sizeof(_capi.AttnBwdDesc) == 144
    assert _capi.AttnBwdDesc.workspace_bytes.offset == 136


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="needs a box WITHOUT a GPU")
def test_uninitialised_contract_without_gpu():
    """src/lib.zig: aule_init -> -1 + error text; everything else reports 'not initialised'."""
    lib = _capi.load()
    assert lib.aule_init() == -1
    assert b"Failed to initialize backend" in lib.aule_get_error()
    assert lib.aule_get_backend_name() == b"Not initialized"        # backend.zig:496-502
    assert lib.aule_get_vendor() == -1 and lib.aule_has_fp16() == -1 and lib.aule_get_subgroup_size() == -1
    assert lib.aule_tensor_create(1, 1, 1, 1) == 0
    assert lib.aule_get_error() == b"Not initialized"               # lib.zig:416
    assert lib.aule_tensor_max() == 1024 and lib.aule_tensor_count() == 0
    assert lib.aule_tensor_size(0) == 0 and lib.aule_tensor_size(5000) == 0
    lib.aule_tensor_destroy(0)
    lib.aule_tensor_destroy(99999)
    assert lib.aule_supports_backward() == 0
    assert lib.aule_attention_forward_paged(1, 1, 1, 1, 0, 0, 0, -1) == -1
    buf = (ctypes.c_float * 4)()
    assert lib.aule_attention_forward(buf, buf, buf, buf, 1, 1, 1, 4, 0) == -1
    assert b"not initialized" in lib.aule_get_error().lower()
    d = _capi.AttnDesc()
    assert lib.aule_attention_forward_ex(ctypes.byref(d)) == -1


@pytest.mark.skipif(os.path.exists("/dev/kfd"), reason="needs a box WITHOUT a GPU")
def test_product_fails_loudly_without_gpu(small_qkv):
    """No CPU fallback: the public API must raise, not silently compute elsewhere."""
    q, k, v = small_qkv
    with pytest.raises(aule.AuleError):
        aule.flash_attention(q, k, v)
    import torch
    with pytest.raises(aule.AuleError):
        aule.flash_attention(torch.from_numpy(q), torch.from_numpy(k), torch.from_numpy(v))
    assert aule.get_available_backends() == []
    assert "hip" in aule.get_backend_errors()


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure; nothing under aule-attention_amd/ may reference it."""
    pkg = os.path.join(ROOT, "aule-attention_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                txt = open(os.path.join(dp, f), errors="replace").read()
                assert "import oracle" not in txt and "liboracle" not in txt and "oracle/" not in txt, (dp, f)


def _route(B, Hq, Hkv, Sq, Sk, D, dtype=2, causal=0, window=-1, scale=0.0):
    lib = ctypes.CDLL(_capi.find_library())
    lib.aule_hip_debug_forward_route.restype = ctypes.c_int32
    lib.aule_hip_debug_forward_route.argtypes = [ctypes.POINTER(_capi.AttnDesc)]
    d = _capi.AttnDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = dtype
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, Sq, Sk, D
    d.causal, d.window_size = causal, window
    d.scale = scale
    return lib.aule_hip_debug_forward_route(ctypes.byref(d))


def test_forward_routing_rule(monkeypatch):
    """The dispatcher is host logic (no device needed).  Non-causal 16-bit problems that the plain tiled launch would
    run badly go to one of two kernels by a MEASURED rule (tools/ppsplit_grid.py, tools/ppsplit_decode.py):
    5 = tiled kernel with packed rows + KV splits (most shapes), 4 = wave-per-chunk split-KV kernel (>= 32 units,
    <= 16 packed rows, K+V >= 100 MB); everything else 16-bit goes to the one-wave-per-SIMD kernel (8; 7 = its small-grid form:
    pairs of Q blocks cut into key ranges + merge) when every Q block has at least four KV tiles, there is no window, D = 128 / 64
    and the scale is positive -- else to the ping-pong kernel (1); fp32 to 0.  (6, the two-waves-per-SIMD stream of rounds 2-3, is
    gone.)"""
    for var in ("AULE_HIP_FWD_KERNEL", "AULE_HIP_FWD_SPLITKV", "AULE_HIP_FWD_PPSPLIT", "AULE_HIP_W4_WINDOW"):
        monkeypatch.delenv(var, raising=False)
    WAVE, TILED_SPLIT, PP, F32, PS_SPLIT, W4 = 4, 5, 1, 0, 7, 8
    assert _route(1, 32, 1, 1, 16384, 64, dtype=1) == TILED_SPLIT    # C5b: 32 -> 17 us
    assert _route(1, 32, 1, 64, 16384, 64, dtype=1) == TILED_SPLIT   # C5c: 44 -> 28 us
    assert _route(8, 32, 8, 1, 2048, 128) == TILED_SPLIT             # 67 MB of K+V: below the streaming corner
    assert _route(8, 32, 8, 1, 8192, 128) == WAVE                    # 64 units, 4 rows, 268 MB
    assert _route(8, 32, 8, 1, 32768, 128) == WAVE                   # 209 vs 243 us
    assert _route(8, 32, 8, 1, 8192, 64) == WAVE                     # D = 64 streaming: 35 vs 47 us
    assert _route(4, 64, 8, 1, 65536, 128) == WAVE                   # 32 units, 8 rows
    assert _route(1, 32, 8, 1, 131072, 128) == TILED_SPLIT           # 8 units: 133 vs 117 us
    assert _route(8, 32, 8, 16, 8192, 128) == TILED_SPLIT            # 64 packed rows: 103 -> 67 us
    assert _route(8, 32, 8, 64, 8192, 128) == TILED_SPLIT            # 194 -> 100 us (the wave kernel: 318)
    assert _route(16, 32, 8, 1, 8192, 128) == TILED_SPLIT            # large-batch decode: 376 -> 117 us
    assert _route(64, 32, 8, 1, 8192, 128) == TILED_SPLIT            # 1506 -> 461 us
    assert _route(1, 8, 8, 300, 8192, 128) == TILED_SPLIT            # Sq > 64, 16 tiled workgroups: 162 -> 34 us
    assert _route(8, 32, 32, 1, 2048, 128) == W4                     # MHA decode: nothing to pack, chip already full
    assert _route(16, 32, 32, 1, 8192, 128) == W4
    assert _route(1, 32, 32, 2048, 2048, 128) == W4                  # 256 full Q blocks
    assert _route(4, 32, 32, 4096, 4096, 128) == W4                  # C2 shape, non-causal
    assert _route(1, 32, 8, 1, 8192, 128, causal=1) == W4            # causal
    # bottom-right aligned causal (additive mode): one query sees every key -> the non-causal problem; short chunks
    # (Sq <= 256, no window) take the tiled kernel's SPLIT instances with a mask; the rest stays on the plain kernel
    assert _route(1, 32, 8, 1, 8192, 128, causal=2) == TILED_SPLIT   # = non-causal, 8 units
    assert _route(8, 32, 8, 1, 8192, 128, causal=2) == WAVE          # = non-causal, the streaming corner
    assert _route(8, 32, 8, 64, 8192, 128, causal=2) == TILED_SPLIT  # 210 -> 94 us
    assert _route(1, 32, 8, 8, 32768, 128, causal=2) == TILED_SPLIT  # 641 -> 44 us
    assert _route(4, 32, 8, 1024, 4096, 128, causal=2) == W4         # Sq > 256
    assert _route(8, 32, 8, 64, 8192, 128, causal=2, window=16) == PP
    assert _route(8, 32, 8, 64, 8192, 128, causal=1) == W4           # top-left: sees the first Sq keys only
    assert _route(1, 32, 8, 1, 8192, 128, causal=2, window=128) == W4   # a window from the end needs the mask (round 6: the window instances; W < 128: PP)
    assert _route(1, 32, 8, 8, 8192, 128, window=4) == PP            # window
    assert _route(1, 32, 8, 8, 8192, 128, window=64) == TILED_SPLIT  # W >= Sq masks nothing: dropped
    # round 6: causal sliding windows of at least two key tiles on the one-wave-per-SIMD kernel's window instances (every query's diagonal key inside
    # Sk); smaller windows, non-causal windows and rows without a visible key stay on the ping-pong kernel; never the key-range split
    assert _route(4, 32, 32, 8192, 8192, 128, causal=1, window=256) == W4
    assert _route(4, 32, 8, 4096, 4096, 64, dtype=1, causal=1, window=1024) == W4
    assert _route(1, 8, 8, 8192, 8192, 128, causal=1, window=256) == W4     # (without the window: PS_SPLIT)
    assert _route(1, 32, 32, 700, 1500, 128, causal=2, window=200) == W4
    assert _route(4, 32, 32, 8192, 8192, 128, causal=1, window=127) == PP
    assert _route(4, 32, 32, 8192, 8192, 128, causal=1, window=128) == W4
    assert _route(4, 32, 32, 4096, 4096, 128, causal=0, window=256) == PP
    assert _route(1, 2, 2, 600, 200, 128, causal=1, window=64) == PP
    assert _route(4, 32, 32, 4096, 4096, 32, causal=1, window=256) == PP
    assert _route(1, 32, 8, 1, 8192, 128, dtype=0) == F32
    assert _route(4, 32, 32, 4096, 4096, 128, causal=1) == W4        # the headline shape
    assert _route(1, 8, 8, 128, 128, 128, causal=1) == PP            # fewer than four KV tiles per Q block
    assert _route(2, 8, 8, 512, 192, 128) == PP
    assert _route(2, 8, 8, 512, 193, 128) == W4
    # small causal grids: every pair of Q blocks cut in two when the doubled item count still fits one round of the chip
    assert _route(1, 8, 8, 8192, 8192, 128, causal=1) == PS_SPLIT    # 128 paired items on 256 CUs
    assert _route(1, 32, 8, 2048, 2048, 128, causal=1) == W4         # 128 pairs of 36 tiles: two pieces of 18 on a full chip lose (round 4)
    assert _route(1, 16, 16, 4096, 4096, 128, causal=1) == PS_SPLIT  # 128 pairs of 68 tiles: two pieces of 34 win
    assert _route(1, 4, 4, 2048, 2048, 128, causal=1) == PS_SPLIT    # 16 pairs: the chip is nowhere near full, two pieces of 18
    assert _route(2, 8, 8, 8192, 8192, 128, causal=1) == W4          # 256 paired items: already one per CU
    assert _route(1, 32, 32, 2048, 2048, 64, dtype=1, causal=1) == W4         # D = 64, the same 128 x 36
    assert _route(1, 8, 8, 4096, 4096, 64, dtype=1, causal=1) == PS_SPLIT     # D = 64: 64 pairs of 68 tiles, four pieces
    assert _route(1, 8, 8, 8192, 8192, 128) == W4                    # non-causal: 256 blocks, one per CU
    assert _route(1, 8, 8, 4096, 4096, 128) == PS_SPLIT              # non-causal: 128 blocks, each cut in two
    assert _route(1, 8, 8, 8192, 8192, 32, causal=1) == PP           # D = 32: the ping-pong kernel
    assert _route(4, 32, 32, 4096, 4096, 128, causal=1, scale=-0.1) == W4    # (round 6) negative scales: the same kernel on negated Q fragments
    assert _route(1, 8, 8, 8192, 8192, 128, causal=1, scale=-0.1) == PS_SPLIT  # ... and its key-range split
    assert _route(8, 32, 32, 2048, 2048, 64, dtype=1, causal=1) == W4         # D = 64 too
    assert _route(1, 8, 8, 300, 300, 128, causal=1) == W4            # one pair whose far block is too short to cut
    assert _route(1, 3, 2, 1, 8192, 128) == -3                       # heads not divisible
    # (AULE_HIP_FWD_SPLITKV=0 is read once per process into a static, so the off-switch is not testable here;
    #  tools/split_grid.py exercises it in a process of its own)


_ROPE_RULE_CHILD = r'''
import ctypes, os, sys
sys.path.insert(0, os.path.join(%(root)r, "aule-attention_amd"))
from aule import _capi
lib = _capi.load()

def fusable(B=4, Hq=32, Hkv=32, Sq=2048, Sk=2048, D=128, dtype=2, causal=1, window=-1, rows=2048, pitch=0, pos=0,
            layout=0, cos=0x1000, sin=0x2000, size=None):
    d = _capi.AttnDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype = dtype
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, Sq, Sk, D
    d.causal, d.window_size = causal, window
    r = _capi.AttnRope()
    r.struct_size = ctypes.sizeof(_capi.AttnRope) if size is None else size
    r.layout, r.table_len, r.table_pitch, r.q_pos_offset, r.cos, r.sin = layout, rows, pitch, pos, cos, sin
    return lib.aule_attention_forward_rope_fusable(ctypes.byref(d), ctypes.byref(r))

assert ctypes.sizeof(_capi.AttnRope) == 40
# (the one-wave-per-SIMD kernel rotates Q itself)
assert fusable() == 1
assert fusable(D=64, dtype=1, causal=0) == 1
assert fusable(Sq=1024, Sk=4096, causal=2, rows=4096, pos=3072) == 1      # bottom-right: queries at Sk - Sq + i
assert fusable(pitch=68) == 1                                              # 16-byte rows
assert fusable(B=1, Hq=8, Hkv=8, Sq=8192, Sk=8192, rows=8192) == 1         # small causal grid (split over the keys)
assert fusable(D=32) == 0                                                  # no fused instance
assert fusable(dtype=0) == 0                                               # fp32 kernel
assert fusable(window=128) == 1                                            # (round 6) the window instances of the same kernel
assert fusable(window=64) == 0                                             # windows shorter than two key tiles: ping-pong kernel
assert fusable(Sq=128, Sk=128) == 0                                        # fewer than four KV tiles: ping-pong kernel
assert fusable(Hq=8, Hkv=8, B=1, Sq=1, Sk=8192, causal=0, rows=8192) == 0  # short-query split paths
assert fusable(layout=1) == 0                                              # interleaved pairs: separate pass
assert fusable(rows=2047) == 0 and fusable(pos=1) == 0                     # table too short
assert fusable(pitch=66) == 0                                              # rows not 16-byte multiples
assert fusable(cos=0x1004) == 0 and fusable(sin=0) == 0                    # alignment, null
assert fusable(size=32) == 0
assert lib.aule_attention_forward_rope_fusable(None, None) == 0
print("RULE OK")
'''


@pytest.mark.parametrize("kernel", [""], ids=["default"])
def test_fused_query_rotation_rule(kernel):
    """aule_attention_forward_rope_fusable() (host logic): the one-wave-per-SIMD kernel rotates Q itself for fp16 / bf16, head_dim
    64 / 128, half-split pairs, 16-byte aligned tables of pitch % 4 == 0 that cover seq_q + q_pos_offset rows (a child process:
    the library reads its kernel switches once per process)."""
    import subprocess
    import sys
    e = {k: v for k, v in os.environ.items() if k not in ("AULE_HIP_FWD_KERNEL", "AULE_HIP_FWD_SOFTMAX")}
    if kernel:
        e["AULE_HIP_FWD_KERNEL"] = kernel
    r = subprocess.run([sys.executable, "-c", _ROPE_RULE_CHILD % {"root": ROOT}], env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "RULE OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_causal_split_plan_invariants(monkeypatch):
    """Route 7's plan (split_cuts in fa_fwd_split.h, through aule_hip_debug_forward_split_plan; host logic): for every pair
    of Q blocks the pieces tile the pair's key tiles exactly; every range of a block has at least four tiles; a cut inside
    a block lies within the keys every row of the block sees whole; the pieces fit the chip in one round and are balanced."""
    monkeypatch.delenv("AULE_HIP_FWD_KERNEL", raising=False)
    monkeypatch.delenv("AULE_HIP_FWD_SPLIT", raising=False)
    lib = _capi.load()
    seen, ns = 0, set()
    for (B, Hq, Hkv, Sq, Sk, D, causal) in itertools.product((1, 2, 3), (4, 8, 32, 40), (1, 4), (300, 512, 777, 1024, 1792, 2048, 4096, 8192, 9000),
                                                           (0, 1000, 4096, 20000), (64, 128), (0, 1, 2)):
        Sk = Sq if Sk == 0 else Sk
        if causal and Sk < Sq:
            continue
        d = _capi.AttnDesc()
        d.struct_size = ctypes.sizeof(_capi.AttnDesc)
        d.dtype = 2
        d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, Sq, Sk, D
        d.causal, d.window_size = causal, -1
        if lib.aule_hip_debug_forward_route(ctypes.byref(d)) != 7:
            assert lib.aule_hip_debug_forward_split_plan(ctypes.byref(d), None, 0) == 0
            continue
        need = -lib.aule_hip_debug_forward_split_plan(ctypes.byref(d), None, 0)
        buf = (ctypes.c_int32 * need)()
        assert lib.aule_hip_debug_forward_split_plan(ctypes.byref(d), buf, need) == need
        n, nwork = buf[0], buf[1]
        coff = Sk - Sq if causal == 2 else (0 if causal else 1 << 30)    # non-causal: every key visible to every row
        nqb = (Sq + 255) // 256
        assert 2 <= n <= 8 and nwork == ((nqb + 1) // 2 if causal else nqb)  # pairs of blocks, or single blocks
        assert n * nwork * B * Hq <= 256 * (2 if D == 64 else 1)
        cut_pairs, longest, total = 0, 0, 0
        for near in range(nwork):
            o = buf[2 + near * 11: 2 + (near + 1) * 11]
            ntf, ntn, b = o[0], o[1], list(o[2:])
            far = nqb - 1 - near if causal else near
            tiles = lambda qb: (max(1, min(Sk, qb * 256 + 256 + coff)) + 63) // 64
            assert ntf == tiles(far) and ntn == (tiles(near) if far != near else 0)
            T = ntf + ntn
            assert b[0] == 0 and all(x == T for x in b[n:]) and all(b[j] <= b[j + 1] for j in range(8))
            pieces = [(b[j], b[j + 1]) for j in range(n) if b[j + 1] > b[j]]
            cut_pairs += len(pieces) >= 2
            for lo, hi in pieces:
                longest = max(longest, hi - lo)
                for (blo, bhi, qb) in ((0, ntf, far), (ntf, T, near)):      # the piece's range of each block
                    t0, t1 = max(lo, blo) - blo, min(hi, bhi) - blo
                    if t1 <= t0:
                        continue
                    assert t1 - t0 >= 4, (B, Hq, Sq, Sk, D, causal, near, b)
                    vis = (qb * 256 + coff) // 64                             # tiles every row of the block sees whole
                    assert t0 == 0 or t0 <= vis, (B, Hq, Sq, Sk, D, causal, near, b)
            total += T
        assert 2 * cut_pairs >= nwork
        assert longest <= -(-max(buf[2 + i * 11] + buf[3 + i * 11] for i in range(nwork)) // n) + 8   # balance: within 8 tiles of ideal
        seen += 1
        ns.add(n)
    assert seen > 100 and {2, 4, 8} <= ns, (seen, ns)


@pytest.mark.parametrize("ranked", [0, 1])
def test_work_order_covers_every_work_item_once(ranked):
    """blockIdx -> (batch, kv head, q head, block) of the tiled kernels (fa_device.h: decode_work, unit by unit; decode_work_ranked,
    block rank by block rank over all units -- the fp32 kernels since round 4), through aule_hip_debug_work_order (host logic):
    every (batch, q head, block) exactly once, the kv head is the q head's group, a unit stays on one XCD (blockIdx % 8) when the
    unit count is a multiple of 8, and in rank order the blocks come heaviest first over ALL units."""
    from aule import _capi
    lib = _capi.load()
    f = lib.aule_hip_debug_work_order
    f.restype = ctypes.c_int32
    f.argtypes = [ctypes.c_int32] * 7 + [ctypes.POINTER(ctypes.c_int32)]
    out = (ctypes.c_int32 * 4)()
    assert f(ranked, 0, 1, 3, 2, 4, 0, out) == -3 and f(ranked, 8, 1, 2, 2, 4, 0, out) == -3 and f(ranked, 0, 1, 2, 2, 4, 0, None) == -3
    for (B, Hq, Hkv, nblk) in ((1, 8, 8, 16), (4, 32, 8, 7), (3, 6, 2, 5), (2, 4, 4, 1), (1, 5, 1, 9), (16, 16, 16, 3), (1, 1, 1, 1)):
        for flag in (0, 1):
            n = B * Hq * nblk
            seen, xcd_of_unit, blks = set(), {}, []
            for bid in range(n):
                assert f(ranked, bid, B, Hq, Hkv, nblk, flag, out) == 0
                b, hk, h, blk = out[0], out[1], out[2], out[3]
                assert 0 <= b < B and 0 <= h < Hq and 0 <= blk < nblk and hk == h // (Hq // Hkv)
                seen.add((b, h, blk))
                blks.append(blk)
                if (B * Hkv) % 8 == 0:
                    assert xcd_of_unit.setdefault((b, hk), bid % 8) == bid % 8
            assert len(seen) == n
            if ranked and (B * Hkv) % 8 != 0:
                want = sorted(blks, reverse=bool(flag))
                assert blks == want                       # one global sequence of ranks
            elif ranked:
                for x in range(8):                        # per XCD: its sequence of ranks
                    mine = blks[x::8]
                    assert mine == sorted(mine, reverse=bool(flag))


_WS_CHILD = r'''
import ctypes, os, sys
sys.path.insert(0, sys.argv[1])
from aule import _capi
lib = _capi.load()
def bwd_ws(dtype, B, Hq, Hkv, Sq, Sk, D, causal):
    d = _capi.AttnBwdDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnBwdDesc)
    d.dtype, d.causal, d.window_size = dtype, causal, -1
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, Sq, Sk, D
    return int(lib.aule_attention_backward_workspace_size(ctypes.byref(d)))
def fwd_ws(dtype, B, Hq, Hkv, Sq, Sk, D, causal):
    d = _capi.AttnDesc()
    d.struct_size = ctypes.sizeof(_capi.AttnDesc)
    d.dtype, d.causal, d.window_size = dtype, causal, -1
    d.batch, d.heads_q, d.heads_kv, d.seq_q, d.seq_k, d.head_dim = B, Hq, Hkv, Sq, Sk, D
    return int(lib.aule_attention_forward_workspace_size(ctypes.byref(d)))
print("B1", bwd_ws(2, 1, 32, 32, 2048, 2048, 128, 1))
print("C3", bwd_ws(2, 4, 32, 8, 2048, 2048, 128, 1))
print("C2", bwd_ws(2, 4, 32, 32, 4096, 4096, 128, 1))
print("F32B", bwd_ws(0, 4, 8, 8, 512, 512, 64, 0))
print("F32F", fwd_ws(0, 4, 8, 8, 512, 512, 64, 0))
print("F32BIG", fwd_ws(0, 4, 32, 32, 2048, 2048, 64, 1))
'''


def _ws_sizes(env):
    import subprocess
    import sys
    e = dict(os.environ)
    for k in ("AULE_HIP_BWD_MODE", "AULE_HIP_BWD_DS_AUTO_MB", "AULE_HIP_BWD_DS_CAP_MB", "AULE_HIP_BWD_DKV", "AULE_HIP_F32_SPLIT"):
        e.pop(k, None)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", _WS_CHILD, os.path.join(ROOT, "aule-attention_amd")], env=e, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    return {l.split()[0]: int(l.split()[1]) for l in r.stdout.splitlines() if l.strip()}


def test_backward_workspace_plans_without_a_device():
    """Host logic of the round-5 dispatch, no device (the mode switches are read once per process: subprocesses).  Default ("auto"): the dS
    workspace of the 5-matmul backward is asked for when the touched dS fits AULE_HIP_BWD_DS_AUTO_MB (B1 H32 S2048: 134 MB) and not for the
    large shapes (C3, C2); AULE_HIP_BWD_MODE=spill asks for it wherever the mode can run, within AULE_HIP_BWD_DS_CAP_MB (whole batch elements:
    the call then runs in chunks); =recompute never.  fp32 small grids: the pieces' planes, forward and backward; AULE_HIP_F32_SPLIT=0: none."""
    rows = lambda B, H, S: ((B * H * S * 4 + 255) // 256) * 256
    ds = lambda B, Hq, Hkv, Sq, Sk: B * Hkv * (4 * ((Sk + 127) // 128)) * (Hq // Hkv) * ((Sq + 31) // 32) * 2048
    c3_partials = 2 * 2 * 4 * 8 * 2048 * 128 * 4     # C3 on the predecessor dK/dV kernel would split its group's heads over 2 workgroups: fp32 dK / dV planes
    auto = _ws_sizes({})
    assert auto["B1"] == 3 * rows(1, 32, 2048) + ds(1, 32, 32, 2048, 2048)
    assert auto["C3"] == 3 * rows(4, 32, 2048) + c3_partials and auto["C2"] == 3 * rows(4, 32, 4096)
    rec = _ws_sizes({"AULE_HIP_BWD_MODE": "recompute"})
    assert rec["B1"] == 3 * rows(1, 32, 2048) and rec["C3"] == auto["C3"]
    sp = _ws_sizes({"AULE_HIP_BWD_MODE": "spill"})
    assert sp["C3"] == auto["C3"] + ds(4, 32, 8, 2048, 2048)                      # 1.07 GB: the whole batch fits the default 8 GB cap
    assert sp["C2"] == auto["C2"] + ds(4, 32, 32, 4096, 4096)                    # 4 x 1.07 GB
    cap = _ws_sizes({"AULE_HIP_BWD_MODE": "spill", "AULE_HIP_BWD_DS_CAP_MB": "2500"})
    assert cap["C2"] == auto["C2"] + 2 * ds(1, 32, 32, 4096, 4096)               # two batch elements per chunk
    none = _ws_sizes({"AULE_HIP_BWD_MODE": "spill", "AULE_HIP_BWD_DS_CAP_MB": "500"})
    assert none["C2"] == auto["C2"]                                              # not even one element fits: the recompute pair
    # fp32 small grids (the reference's Zig benchmark shape: 128 work items for 512 slots -> 4 pieces)
    assert auto["F32F"] == 4 * (4 * 8 * 512) * (64 + 4) * 4 and auto["F32BIG"] == 0
    assert auto["F32B"] == rows(4, 8, 512) + 2 * 4 * (4 * 8 * 512) * 64 * 4        # delta + max(dQ planes, dK + dV planes) of 4 pieces
    off = _ws_sizes({"AULE_HIP_F32_SPLIT": "0"})
    assert off["F32F"] == 0 and off["F32B"] == rows(4, 8, 512)
