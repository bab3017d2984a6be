"""Handle-based wrapper over the legacy Aule C-ABI, HIP build.

Host-side mirror of the reference's python/aule/vulkan.py (class Aule :164-1168,
GpuTensor :77-161, module helpers :1170-1300) for the hot path only: same class and
method names, argument meaning and error behaviour (AuleError / ValueError), bound
to libaule.so's legacy symbols (aule_tensor_*, aule_attention_forward_gpu,
aule_attention_forward_with_lse, aule_attention_backward).

Differences from the Vulkan binding, all widenings: head_dim <= 128 (vulkan.py:707-711
caps at 64), and the work runs on the gfx950 MFMA kernels.
"""
import ctypes
from typing import Optional, Tuple

import numpy as np

from . import _capi
from ._capi import AuleError

_FP = ctypes.POINTER(ctypes.c_float)


def _fptr(a):
    return a.ctypes.data_as(_FP)


class GpuTensor:
    """A persistent fp32 device tensor behind a 1-based slot handle (vulkan.py:77-161)."""

    def __init__(self, aule, handle, shape, dtype=np.float32):
        self._aule = aule
        self._handle = handle
        self._shape = tuple(int(x) for x in shape)
        self._dtype = np.dtype(dtype)
        self._size = int(np.prod(self._shape))

    dtype = property(lambda self: self._dtype)
    shape = property(lambda self: self._shape)
    size = property(lambda self: self._size)
    handle = property(lambda self: self._handle)

    def upload(self, data):
        if data.size != self._size:
            raise ValueError(f"Size mismatch: tensor has {self._size} elements, got {data.size}")
        data = np.ascontiguousarray(data, dtype=self._dtype).ravel()
        rc = self._aule._lib.aule_tensor_upload(self._handle, _fptr(data), self._size)
        if rc != 0:
            raise AuleError(f"Upload failed: {_capi.last_error()}")

    def download(self):
        out = np.empty(self._size, dtype=self._dtype)
        rc = self._aule._lib.aule_tensor_download(self._handle, _fptr(out), self._size)
        if rc != 0:
            raise AuleError(f"Download failed: {_capi.last_error()}")
        return out.reshape(self._shape)

    def destroy(self):
        if self._handle != 0:
            self._aule._lib.aule_tensor_destroy(self._handle)
            self._handle = 0


class Aule:
    """Library handle (vulkan.py:164-222).  One process-wide context, like the reference."""

    def __init__(self, library_path: Optional[str] = None):
        if library_path is not None:
            import os
            os.environ["AULE_LIBRARY_PATH"] = str(library_path)
        self._lib = _capi.get_lib()   # raises AuleError without a HIP device
        self._tensors = []

    # ---- device info (vulkan.py:409-480)
    @property
    def device_name(self) -> str:
        buf = ctypes.create_string_buffer(256)
        n = self._lib.aule_get_device_name(buf, 256)
        return buf.value.decode() if n > 0 else "Unknown"

    @property
    def vendor(self) -> str:
        return {0: "other", 1: "amd", 2: "nvidia", 3: "intel", 4: "apple"}.get(self._lib.aule_get_vendor(), "unknown")

    @property
    def is_amd_optimized(self) -> bool:
        return self._lib.aule_is_amd_optimized() == 1

    @property
    def fp16_supported(self) -> bool:
        return self._lib.aule_has_fp16() == 1

    @property
    def subgroup_size(self) -> int:
        return int(self._lib.aule_get_subgroup_size())

    @property
    def backend_name(self) -> str:
        return self._lib.aule_get_backend_name().decode()

    def get_device_info(self) -> dict:
        return {"device_name": self.device_name, "vendor": self.vendor, "amd_optimized": self.is_amd_optimized,
                "fp16_supported": self.fp16_supported, "subgroup_size": self.subgroup_size,
                "backend": self.backend_name}

    # ---- tensor pool (vulkan.py:546-611)
    @property
    def tensor_count(self) -> int:
        return int(self._lib.aule_tensor_count())

    @property
    def tensor_max(self) -> int:
        return int(self._lib.aule_tensor_max())

    def clear_tensors(self) -> None:
        self._lib.aule_tensor_clear_all()
        for t in self._tensors:
            t._handle = 0
        self._tensors = []

    def tensor(self, shape: Tuple[int, int, int, int], dtype=np.float32) -> GpuTensor:
        if len(shape) != 4:
            raise ValueError(f"Shape must be 4D [batch, heads, seq, dim], got {len(shape)}D")
        h = self._lib.aule_tensor_create(*[int(x) for x in shape])
        if h == 0:
            raise AuleError(f"Failed to create tensor: {_capi.last_error()}")
        t = GpuTensor(self, h, shape, dtype)
        self._tensors.append(t)
        return t

    # ---- compute (vulkan.py:613-962)
    def attention_gpu(self, Q: GpuTensor, K: GpuTensor, V: GpuTensor, output: GpuTensor,
                      rot_cos: Optional[GpuTensor] = None, rot_sin: Optional[GpuTensor] = None,
                      causal: bool = False, window_size: int = -1) -> None:
        rc = self._lib.aule_attention_forward_gpu(
            Q.handle, K.handle, V.handle, output.handle,
            rot_cos.handle if rot_cos is not None else 0, rot_sin.handle if rot_sin is not None else 0,
            1 if causal else 0, int(window_size))
        if rc != 0:
            raise AuleError(f"GPU attention failed: {_capi.last_error()}")

    def attention(self, query, key, value, rot_cos=None, rot_sin=None, causal: bool = False,
                  window_size: int = -1):
        """fp32 attention on NumPy arrays through the handle ABI (GQA / cross-attention ok)."""
        query, key, value = (np.asarray(x) for x in (query, key, value))
        if query.ndim != 4 or key.ndim != 4 or value.ndim != 4:
            raise ValueError("Inputs must be 4D [batch, heads, seq, dim]")
        if key.shape != value.shape:
            raise ValueError(f"Key and Value must have same shape, got K={key.shape}, V={value.shape}")
        B, Hq, Sq, D = query.shape
        if key.shape[0] != B or key.shape[3] != D:
            raise ValueError(f"Shape mismatch: Q={query.shape}, K={key.shape}")
        if Hq % key.shape[1] != 0:
            raise ValueError(f"Query heads {Hq} must be divisible by KV heads {key.shape[1]}")
        if D > 128:
            raise ValueError(f"head_dim must be <= 128, got {D}")
        q = self.tensor(query.shape)
        k = self.tensor(key.shape)
        v = self.tensor(value.shape)
        o = self.tensor(query.shape)
        extra = []
        try:
            q.upload(query.astype(np.float32, copy=False))
            k.upload(key.astype(np.float32, copy=False))
            v.upload(value.astype(np.float32, copy=False))
            if rot_cos is not None and rot_sin is not None:
                # tables [.., seq, head_dim/2], interleaved pairs (vulkan.py:725-772, attention_f32.comp:98-111)
                rc, rs = (np.asarray(x, dtype=np.float32) for x in (rot_cos, rot_sin))
                rc = rc.reshape((1,) * (4 - rc.ndim) + rc.shape)
                rs = rs.reshape((1,) * (4 - rs.ndim) + rs.shape)
                tc, ts = self.tensor(rc.shape), self.tensor(rs.shape)
                extra += [tc, ts]
                tc.upload(rc)
                ts.upload(rs)
                self.attention_gpu(q, k, v, o, tc, ts, causal=causal, window_size=window_size)
            else:
                self.attention_gpu(q, k, v, o, causal=causal, window_size=window_size)
            return o.download()
        finally:
            for t in [q, k, v, o] + extra:
                t.destroy()
                if t in self._tensors:
                    self._tensors.remove(t)

    def supports_backward(self) -> bool:
        return self._lib.aule_supports_backward() == 1

    @staticmethod
    def _check_mha(query, key, value):
        if query.ndim != 4:
            raise ValueError("Inputs must be 4D [batch, heads, seq, dim]")
        if query.shape != key.shape or query.shape != value.shape:
            raise ValueError("Q, K, V must have same shape (training path is MHA, Sq == Sk)")
        if query.shape[3] > 128:
            raise ValueError(f"head_dim must be <= 128, got {query.shape[3]}")

    def attention_forward_with_lse(self, query, key, value, causal: bool = False):
        """Returns (output, lse) -- vulkan.py:824-889 / src/lib.zig:765."""
        query, key, value = (np.ascontiguousarray(x, dtype=np.float32) for x in (query, key, value))
        self._check_mha(query, key, value)
        B, H, S, D = query.shape
        out = np.empty_like(query)
        lse = np.empty((B, H, S), dtype=np.float32)
        rc = self._lib.aule_attention_forward_with_lse(_fptr(query), _fptr(key), _fptr(value), _fptr(out),
                                                       _fptr(lse), B, H, S, D, 1 if causal else 0)
        if rc != 0:
            raise AuleError(f"Forward with LSE failed: {_capi.last_error()}")
        return out, lse

    def attention_backward(self, query, key, value, output, grad_output, lse, causal: bool = False):
        """Returns (dQ, dK, dV) -- vulkan.py:891-962 / src/lib.zig:639."""
        arrs = [np.ascontiguousarray(x, dtype=np.float32) for x in (query, key, value, output, grad_output, lse)]
        query, key, value, output, grad_output, lse = arrs
        self._check_mha(query, key, value)
        B, H, S, D = query.shape
        dq, dk, dv = np.empty_like(query), np.empty_like(key), np.empty_like(value)
        rc = self._lib.aule_attention_backward(_fptr(query), _fptr(key), _fptr(value), _fptr(output),
                                               _fptr(grad_output), _fptr(lse), _fptr(dq), _fptr(dk), _fptr(dv),
                                               B, H, S, D, 1 if causal else 0)
        if rc != 0:
            raise AuleError(f"Backward failed: {_capi.last_error()}")
        return dq, dk, dv

    def forward_host(self, query, key, value, causal: bool = False):
        """aule_attention_forward (src/lib.zig:312): MHA, Sq == Sk, fp32 host pointers."""
        query, key, value = (np.ascontiguousarray(x, dtype=np.float32) for x in (query, key, value))
        self._check_mha(query, key, value)
        B, H, S, D = query.shape
        out = np.empty_like(query)
        rc = self._lib.aule_attention_forward(_fptr(query), _fptr(key), _fptr(value), _fptr(out), B, H, S, D,
                                              1 if causal else 0)
        if rc != 0:
            raise AuleError(f"Attention failed: {_capi.last_error()}")
        return out

    def close(self):
        # Like the reference (vulkan.py:1152-1156) the process-wide context is NOT shut
        # down here; only this object's tensors are released.
        for t in list(self._tensors):
            t.destroy()
        self._tensors = []

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False


_instance = None


def _get():
    global _instance
    if _instance is None:
        _instance = Aule()
    return _instance


def attention(query, key, value, causal: bool = False, window_size: int = -1):
    """Module-level convenience (vulkan.py:1170-1198)."""
    return _get().attention(query, key, value, causal=causal, window_size=window_size)


def supports_backward() -> bool:
    return _get().supports_backward()


def attention_forward_with_lse(query, key, value, causal: bool = False):
    return _get().attention_forward_with_lse(query, key, value, causal=causal)


def attention_backward(query, key, value, output, grad_output, lse, causal: bool = False):
    return _get().attention_backward(query, key, value, output, grad_output, lse, causal=causal)
