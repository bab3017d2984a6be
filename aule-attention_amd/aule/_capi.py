"""ctypes binding of libaule.so (include/aule.h) -- the HIP build of the Aule C-ABI.

Counterpart of the reference's python/aule/vulkan.py:31-69 (library lookup) and
:224-406 (signatures).  The shared object is built in-tree by
aule-attention_amd/csrc/Makefile into <pkg>/lib/libaule.so, the first location the
reference binding searches too (vulkan.py:31-69).

There is no fallback: if the library is missing or no HIP device is visible,
`load()` / `get_lib()` raise AuleError.
"""
import ctypes
import os
import sys
import threading

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_CANDIDATES = [
    os.path.join(_PKG_DIR, "lib", "libaule.so"),
]


class AuleError(RuntimeError):
    """Raised for any failure reported by libaule (name kept from vulkan.py)."""


class AttnDesc(ctypes.Structure):
    """struct aule_attn_desc (include/aule.h)."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("dtype", ctypes.c_int32),
        ("batch", ctypes.c_uint32),
        ("heads_q", ctypes.c_uint32),
        ("heads_kv", ctypes.c_uint32),
        ("seq_q", ctypes.c_uint32),
        ("seq_k", ctypes.c_uint32),
        ("head_dim", ctypes.c_uint32),
        ("scale", ctypes.c_float),
        ("causal", ctypes.c_int32),
        ("window_size", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("stream", ctypes.c_void_p),
        ("q", ctypes.c_void_p),
        ("k", ctypes.c_void_p),
        ("v", ctypes.c_void_p),
        ("out", ctypes.c_void_p),
        ("lse", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p),
        ("workspace_bytes", ctypes.c_uint64),
    ]


class AttnBwdDesc(ctypes.Structure):
    """struct aule_attn_bwd_desc (include/aule.h)."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("dtype", ctypes.c_int32),
        ("batch", ctypes.c_uint32),
        ("heads_q", ctypes.c_uint32),
        ("heads_kv", ctypes.c_uint32),
        ("seq_q", ctypes.c_uint32),
        ("seq_k", ctypes.c_uint32),
        ("head_dim", ctypes.c_uint32),
        ("scale", ctypes.c_float),
        ("causal", ctypes.c_int32),
        ("window_size", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("stream", ctypes.c_void_p),
        ("q", ctypes.c_void_p),
        ("k", ctypes.c_void_p),
        ("v", ctypes.c_void_p),
        ("out", ctypes.c_void_p),
        ("dout", ctypes.c_void_p),
        ("lse", ctypes.c_void_p),
        ("dq", ctypes.c_void_p),
        ("dk", ctypes.c_void_p),
        ("dv", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p),
        ("workspace_bytes", ctypes.c_uint64),
    ]


class PagedDesc(ctypes.Structure):
    """struct aule_paged_desc (include/aule.h)."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("dtype", ctypes.c_int32),
        ("batch", ctypes.c_uint32),
        ("heads_q", ctypes.c_uint32),
        ("heads_kv", ctypes.c_uint32),
        ("head_dim", ctypes.c_uint32),
        ("block_size", ctypes.c_uint32),
        ("max_blocks", ctypes.c_uint32),
        ("scale", ctypes.c_float),
        ("window_size", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("stream", ctypes.c_void_p),
        ("q", ctypes.c_void_p),
        ("k_cache", ctypes.c_void_p),
        ("v_cache", ctypes.c_void_p),
        ("block_tables", ctypes.c_void_p),
        ("context_lens", ctypes.c_void_p),
        ("out", ctypes.c_void_p),
        ("workspace", ctypes.c_void_p),
        ("workspace_bytes", ctypes.c_uint64),
    ]


class RopeDesc(ctypes.Structure):
    """struct aule_rope_desc (include/aule.h)."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("dtype", ctypes.c_int32),
        ("rows_bh", ctypes.c_uint64),
        ("seq", ctypes.c_uint32),
        ("head_dim", ctypes.c_uint32),
        ("row_pitch", ctypes.c_uint32),
        ("table_len", ctypes.c_uint32),
        ("table_pitch", ctypes.c_uint32),
        ("layout", ctypes.c_int32),
        ("inverse", ctypes.c_int32),
        ("pos_offset", ctypes.c_uint32),
        ("device", ctypes.c_int32),
        ("stream", ctypes.c_void_p),
        ("in_", ctypes.c_void_p),
        ("out", ctypes.c_void_p),
        ("cos", ctypes.c_void_p),
        ("sin", ctypes.c_void_p),
    ]


class AttnRope(ctypes.Structure):
    """struct aule_attn_rope (include/aule.h): the query rotation fused into the forward kernel."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32),
        ("layout", ctypes.c_int32),
        ("table_len", ctypes.c_uint32),
        ("table_pitch", ctypes.c_uint32),
        ("q_pos_offset", ctypes.c_uint32),
        ("reserved", ctypes.c_uint32),
        ("cos", ctypes.c_void_p),
        ("sin", ctypes.c_void_p),
    ]


ROPE_HALF, ROPE_INTERLEAVED = 0, 1
DTYPE_F32, DTYPE_F16, DTYPE_BF16 = 0, 1, 2

# Every symbol include/aule.h declares: (name, restype, argtypes)
_FP = ctypes.POINTER(ctypes.c_float)
_U32, _I32, _U64, _U8 = ctypes.c_uint32, ctypes.c_int32, ctypes.c_uint64, ctypes.c_uint8
class IpcHandle(ctypes.Structure):
    """aule_ipc_handle (include/aule.h): 64 opaque bytes, a hipIpcMemHandle_t."""
    _fields_ = [("bytes", ctypes.c_ubyte * 64)]


SIGNATURES = [
    ("aule_init", _I32, []),
    ("aule_shutdown", None, []),
    ("aule_get_error", ctypes.c_char_p, []),
    ("aule_get_backend_name", ctypes.c_char_p, []),
    ("aule_get_vendor", _I32, []),
    ("aule_get_gpu_vendor", _I32, []),
    ("aule_is_amd_optimized", _I32, []),
    ("aule_has_fp16", _I32, []),
    ("aule_get_subgroup_size", _I32, []),
    ("aule_get_device_name", _I32, [ctypes.c_char_p, _U32]),
    ("aule_set_shader_variant", _I32, [_U8]),
    ("aule_get_shader_variant", _I32, []),
    ("aule_has_shader_variant", _I32, [_U8]),
    ("aule_supports_backward", _I32, []),
    ("aule_attention_forward", _I32, [_FP, _FP, _FP, _FP, _U32, _U32, _U32, _U32, _I32]),
    ("aule_tensor_create", _U64, [_U32, _U32, _U32, _U32]),
    ("aule_tensor_create_u32", _U64, [_U32, _U32, _U32, _U32]),
    ("aule_tensor_destroy", None, [_U64]),
    ("aule_tensor_upload", _I32, [_U64, _FP, _U32]),
    ("aule_tensor_download", _I32, [_U64, _FP, _U32]),
    ("aule_tensor_download_u32", _I32, [_U64, ctypes.POINTER(_U32), _U32]),
    ("aule_tensor_size", _U32, [_U64]),
    ("aule_tensor_count", _U32, []),
    ("aule_tensor_max", _U32, []),
    ("aule_tensor_clear_all", None, []),
    ("aule_attention_forward_gpu", _I32, [_U64, _U64, _U64, _U64, _U64, _U64, _I32, _I32]),
    ("aule_attention_forward_with_lse", _I32, [_FP, _FP, _FP, _FP, _FP, _U32, _U32, _U32, _U32, _I32]),
    ("aule_attention_backward", _I32, [_FP] * 9 + [_U32, _U32, _U32, _U32, _I32]),
    ("aule_attention_forward_paged", _I32, [_U64] * 6 + [_I32, _I32]),
    ("aule_spatial_sort", _I32, [_U64, _U64, _U64, _U32]),
    ("aule_attention_forward_gravity", _I32, [_U64] * 7 + [_I32, _U32, _I32]),
    ("aule_attention_forward_ex", _I32, [ctypes.POINTER(AttnDesc)]),
    ("aule_attention_backward_ex", _I32, [ctypes.POINTER(AttnBwdDesc)]),
    ("aule_attention_backward_workspace_size", _U64, [ctypes.POINTER(AttnBwdDesc)]),
    ("aule_attention_paged_decode_ex", _I32, [ctypes.POINTER(PagedDesc)]),
    ("aule_rope_ex", _I32, [ctypes.POINTER(RopeDesc)]),
    ("aule_attention_forward_rope_ex", _I32, [ctypes.POINTER(AttnDesc), ctypes.POINTER(AttnRope)]),
    ("aule_attention_forward_rope_fusable", _I32, [ctypes.POINTER(AttnDesc), ctypes.POINTER(AttnRope)]),
    ("aule_attention_forward_workspace_size", ctypes.c_uint64, [ctypes.POINTER(AttnDesc)]),
    ("aule_attention_paged_decode_workspace_size", ctypes.c_uint64, [ctypes.POINTER(PagedDesc)]),
    ("aule_peer_alloc", _I32, [_I32, _U64, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(IpcHandle)]),
    ("aule_peer_free", _I32, [_I32, ctypes.c_void_p]),
    ("aule_peer_open", _I32, [_I32, ctypes.POINTER(IpcHandle), ctypes.POINTER(ctypes.c_void_p)]),
    ("aule_peer_close", _I32, [_I32, ctypes.c_void_p]),
    ("aule_peer_copy_async", _I32, [_I32, ctypes.c_void_p, ctypes.c_void_p, _U64, ctypes.c_void_p]),
    ("aule_hip_build_info", ctypes.c_char_p, []),
    ("aule_hip_debug_forward_route", _I32, [ctypes.POINTER(AttnDesc)]),
    ("aule_hip_debug_last_backward_route", _I32, []),
    ("aule_hip_debug_forward_split_plan", _I32, [ctypes.POINTER(AttnDesc), ctypes.POINTER(_I32), _I32]),
    ("aule_hip_debug_work_order", _I32, [_I32] * 7 + [ctypes.POINTER(_I32)]),
]

_lib = None
_lib_path = None
_initialized = False
_lock = threading.Lock()


def find_library():
    """Path of libaule.so (AULE_LIBRARY_PATH override, then <pkg>/lib)."""
    env = os.environ.get("AULE_LIBRARY_PATH")
    cands = ([env] if env else []) + _LIB_CANDIDATES
    for p in cands:
        if p and os.path.exists(p):
            return p
    raise AuleError(
        "libaule.so (HIP build) not found; build it with `make -C aule-attention_amd/csrc` "
        "or `python -c 'import __graft_entry__ as g; g.build()'`. Searched: " + ", ".join(cands))


def _preload_torch_hip_runtime():
    """ONE HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 (torch/lib); libaule.so
    links the system's (/opt/rocm).  If libaule.so is loaded BEFORE torch, the process ends up with both, torch's takes the device
    and aule_init() then reports "no ROCm-capable device" (found in round 4: build() followed by smoke() in one process).  Loaded
    after torch, libaule.so binds to the copy that is already there.  So when torch is installed but not imported yet, its runtime is
    loaded first (by path, without importing torch); a later `import torch` reuses it.  AULE_HIP_RUNTIME=system keeps the system's."""
    if os.environ.get("AULE_HIP_RUNTIME", "") == "system" or "torch" in sys.modules:
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except Exception:  # noqa: BLE001
        return
    if spec is None or not spec.origin:
        return
    p = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(p):
        try:
            ctypes.CDLL(p, mode=ctypes.RTLD_GLOBAL)
        except OSError:
            pass


def _mapped_hip_runtimes():
    """paths of every libamdhip64 mapped into this process (two = the double-runtime condition _preload_torch_hip_runtime avoids)"""
    out = []
    try:
        with open("/proc/self/maps") as fh:
            for line in fh:
                p = line.split()[-1] if line.strip() else ""
                if "libamdhip64" in p and p not in out:
                    out.append(p)
    except OSError:
        pass
    return out


def load():
    """dlopen libaule.so and declare every signature. Does NOT need a GPU."""
    global _lib, _lib_path
    with _lock:
        if _lib is None:
            path = find_library()
            _preload_torch_hip_runtime()
            lib = ctypes.CDLL(path)
            for name, restype, argtypes in SIGNATURES:
                fn = getattr(lib, name)  # AttributeError if the export is missing
                fn.restype = restype
                fn.argtypes = argtypes
            _lib, _lib_path = lib, path
    return _lib


def library_path():
    load()
    return _lib_path


def get_lib():
    """Loaded AND initialised library (aule_init succeeded => a HIP device exists)."""
    global _initialized
    lib = load()
    if not _initialized:
        with _lock:
            if not _initialized:
                rc = lib.aule_init()
                if rc != 0:
                    raise AuleError("aule_init failed (%d): %s [HIP runtime(s) mapped in this process: %s; if torch's bundled copy and the "
                                    "system's are both listed, or the wrong one is bound, set AULE_HIP_RUNTIME=system or import torch first]"
                                    % (rc, last_error(lib), ", ".join(_mapped_hip_runtimes()) or "none found in /proc/self/maps"))
                _initialized = True
    return lib


def last_error(lib=None):
    lib = lib or load()
    msg = lib.aule_get_error()
    return msg.decode("utf-8", "replace") if msg else "unknown error"


def check(rc, what):
    if rc != 0:
        raise AuleError("%s failed (%d): %s" % (what, rc, last_error()))
