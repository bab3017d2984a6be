"""Shared helpers for the GPU parity tests."""
import numpy as np

TORCH_DTYPES = {}


def torch_dtype(name):
    import torch
    return {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[name]


def quantize(a, name):
    """Round an fp32 array to the storage dtype and back (what the kernel will see)."""
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(torch_dtype(name)).float().numpy()


# Output tolerance |out - ref| <= atol + rtol*|ref|, against an fp64-arithmetic oracle run on
# the SAME quantised inputs.  BASELINE.json's bar: 1e-3 (fp16/bf16), 1e-5 (fp32).  Two terms are
# inherent to the 16-bit ALGORITHM the reference itself runs and are added for fp16/bf16
# (SURVEY.md 7.3 "bf16 tolerance"; the reference's own bf16 kernel is 1.5e-2 off fp32 SDPA):
#   * O is rounded to storage: relative error <= u, u = 2^-9 (bf16) / 2^-12 (fp16); rtol = 2u;
#   * P is cast to the V dtype before the PV product (triton_flash_amd.py:222): every weight
#     carries relative error <= u and the weights sum to 1, so |dO| <= u * max|V|  (atol term,
#     see fwd_tol()).
UNIT_ROUNDOFF = {"fp32": 0.0, "fp16": 2.0 ** -12, "bf16": 2.0 ** -9}
FWD_TOL = {"fp32": (1e-5, 1e-5), "fp16": (1e-3, 2.0 ** -11), "bf16": (1e-3, 2.0 ** -8)}


def fwd_tol(dtype, vmax, sides=1):
    """(atol, rtol) for a forward output whose V has max |V| = vmax.  sides=2 when the
    comparison target is itself a 16-bit result (golden vectors from the reference kernels)."""
    atol, rtol = FWD_TOL[dtype]
    return atol + sides * UNIT_ROUNDOFF[dtype] * float(vmax), sides * rtol
# LSE is fp32 in every variant
LSE_TOL = {"fp32": 1e-5, "fp16": 1e-3, "bf16": 1e-3}
# Gradients: reference bar is 1e-2 (python/tests/test_triton.py:92-94)
BWD_TOL = {"fp32": (2e-5, 2e-5), "fp16": (4e-3, 4e-3), "bf16": (2e-2, 2e-2)}


def assert_close(got, ref, atol, rtol, what=""):
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    assert np.isfinite(got).all(), f"{what}: non-finite values in result"
    err = np.abs(got - ref)
    bound = atol + rtol * np.abs(ref)
    bad = err > bound
    if bad.any():
        idx = np.unravel_index(np.argmax(err - bound), err.shape)
        raise AssertionError(
            f"{what}: {int(bad.sum())}/{bad.size} elements out of tolerance; worst at {idx}: "
            f"got {got[idx]:.6g} ref {ref[idx]:.6g} |err| {err[idx]:.3g} > {bound[idx]:.3g}; "
            f"max|err| {err.max():.3g}")
