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


# Rows that see many keys: the P roundings average out (each weight carries relative error <= u, the weights are ~1/n: the
# absolute error of the row is ~ u max|V| / sqrt(n)), so SURVEY.md 7.3's rule -- BASELINE.json's 1e-3 plus the rounding of O to
# storage, rtol 2u -- holds WITHOUT the u max|V| term.  The full-size configurations (C2, C4 shard, C5, C3 forward) use it for
# every sampled row with at least MANY_KEYS visible keys and keep fwd_tol() for the few rows at the start of a causal sequence;
# they print what they achieved (pytest -s / the captured output on failure; DESIGN.md 4 has the table).
MANY_KEYS = 64


def assert_close_rows(got, ref, nkeys, dtype, vmax, what=""):
    """got, ref: [rows, D]; nkeys[r] = keys row r sees.  Tight bound for rows with >= MANY_KEYS keys, fwd_tol() for the rest.
    Returns (max abs error of the many-key rows, of the few-key rows) and prints them."""
    got = np.asarray(got, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    nkeys = np.asarray(nkeys)
    many = nkeys >= MANY_KEYS
    a_t, r_t = FWD_TOL[dtype]
    res = []
    for name, sel, (atol, rtol) in (("rows with >= %d keys" % MANY_KEYS, many, (a_t, r_t)), ("rows with fewer keys", ~many, fwd_tol(dtype, vmax))):
        if sel.any():
            assert_close(got[sel], ref[sel], atol, rtol, "%s, %s" % (what, name))
            e = np.abs(got[sel] - ref[sel])
            res.append(float(e.max()))
            print("%s: %s (%d): max|err| %.3e, max|err|/(atol+rtol|ref|) %.2f  [atol %.2e rtol %.2e]"
                  % (what, name, int(sel.sum()), e.max(), (e / (atol + rtol * np.abs(ref[sel]))).max(), atol, rtol))
        else:
            res.append(0.0)
    return tuple(res)


# LSE is fp32 in every variant
LSE_TOL = {"fp32": 1e-5, "fp16": 1e-3, "bf16": 1e-3}
# Gradients: the reference's bar is element-wise rtol = atol = 1e-2 (python/tests/test_triton.py:92-94).  Here (round 5):
# |err| <= atol * max(1, max|grad|) + rtol * |ref| with bf16 (5e-3, 1e-2) -- achieved 2.3 .. 3.4e-3 of max|grad| on C3 --, fp16 (4e-3, 4e-3).
BWD_TOL = {"fp32": (2e-5, 2e-5), "fp16": (4e-3, 4e-3), "bf16": (5e-3, 1e-2)}


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
