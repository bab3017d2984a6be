"""A fixed-seed slice of tools/fuzz_parity.py inside the suite: random combinations of dtype, batch, GQA ratio, Sq, Sk,
head_dim, causal mode (none / top-left / bottom-right), window and scale, forward AND backward against the fp64 oracle;
random paged-decode problems (shuffled block tables, lengths 0 and 1, power-of-two and other block sizes, window); random
RoPE passes (both layouts, inverse, offsets, vector and scalar head dims, in place).  Deterministic: the seeds are fixed,
so this is a regression test over ~150 configurations nobody enumerated by hand, not a flaky one.  The tool itself takes
any seed and size (700 + 300 + 300 configurations were clean when it was written)."""
import importlib.util
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def fuzz(oracle_mod):   # (oracle_mod builds liboracle.so first; the tool imports `oracle` itself)
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(_ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_random_attention_configurations_forward_and_backward(fuzz):
    rng = np.random.RandomState(123)
    failures, routes = [], {}
    for i in range(90):
        cfg = fuzz.draw(rng)
        route, errs = fuzz.run(cfg, 1000 + i)
        routes[route] = routes.get(route, 0) + 1
        if errs:
            failures.append((i, route, cfg, errs))
    assert not failures, failures
    assert {0, 1, 5} <= set(routes), routes     # the fp32, tiled and packed / KV-split routes were all exercised


def test_random_paged_decode_problems(fuzz):
    rng = np.random.RandomState(321)
    failures = []
    for i in range(40):
        cfg, errs = fuzz.run_paged(rng, i)
        if errs:
            failures.append((i, cfg, errs))
    assert not failures, failures


def test_random_rope_passes(fuzz):
    rng = np.random.RandomState(231)
    failures = []
    for i in range(40):
        cfg, errs = fuzz.run_rope(rng, i)
        if errs:
            failures.append((i, cfg, errs))
    assert not failures, failures


def test_random_long_sequence_configurations_forward_and_backward(fuzz):
    """Round 6: Sq in {2048, 4096} (MHA / GQA, every causal mode, causal windows, D 64 / 128): the longest streams of the backward
    kernels, gradients judged head by head in fp64 (oracle.bwd_head_f64), forward on sampled rows."""
    rng = np.random.RandomState(606)
    failures = []
    for i in range(10):
        (route, cfg), errs = fuzz.run_long(rng, i)
        if errs:
            failures.append((i, route, cfg, errs))
    assert not failures, failures


def test_random_window_configurations_on_the_window_instances(fuzz):
    """Round 6: causal sliding windows on the one-wave-per-SIMD forward's window instances (windows of 128 .. 2500 keys, aligned and not, bottom-right
    offsets that are no tile multiples, ragged blocks, GQA, D 64 / 128, fp16 and bf16; one draw in four with a spiked key in front of or inside some
    rows' windows): every output row and LSE against the fp64 judge.  The kind that found the fp16 verdict bound."""
    rng = np.random.RandomState(77)
    failures, routes = [], {}
    for i in range(40):
        (route, cfg), errs = fuzz.run_window(rng, i)
        routes[route] = routes.get(route, 0) + 1
        if errs:
            failures.append((i, route, cfg, errs))
    assert not failures, failures
    assert routes.get(8, 0) >= 30, routes     # (a window beyond the sequence is dropped at the boundary: those draws run the plain routes)
