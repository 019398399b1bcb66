"""GPU parity of the workgroup-cooperative pooling kernel (raglite_amd/csrc/pool_norm.hip: pool_norm_coop_kernel, opt-in
with RAGLITE_POOL_COOP=1) -- late-chunking mean-pool + L2 normalise + fp16 cast
(`/root/reference/src/raglite/_embed.py:131-140`) for ordered, (nearly) gap-free spans.  Bars: bit-identical to the default
wave-private kernel -- same fp64 sums in row order, same statements for mean / norm / cast, same order of the norm reduction
-- and within 1 fp16 ulp of `oracle.pool_norm_cast`; span lists the cooperative kernel does not take (unordered,
overlapping, sparse) fall back on the device and give the same results."""

import os

import numpy as np
import pytest

import raglite_amd
from oracle import oracle

pytestmark = pytest.mark.gpu


def _both(tokens, b, e, **kw):
    r = raglite_amd.pool_norm(tokens, b, e, want_f32=True, want_f16=True, **kw)  # the default (wave-private) kernel
    os.environ["RAGLITE_POOL_COOP"] = "1"
    try:
        o = raglite_amd.pool_norm(tokens, b, e, want_f32=True, want_f16=True, **kw)
    finally:
        del os.environ["RAGLITE_POOL_COOP"]
    return o, r


def _same(a, b):
    return np.array_equal(a.view(np.uint32) if a.dtype == np.float32 else a.view(np.uint16),
                          b.view(np.uint32) if b.dtype == np.float32 else b.view(np.uint16))


@pytest.mark.parametrize("dim", [256, 512, 1024])
@pytest.mark.parametrize("layout", ["tight", "gaps", "ones", "empties", "long", "few_rows"])
@pytest.mark.parametrize("normalize,eps", [(True, 0.0), (True, 1e-12), (False, 0.0)])
def test_coop_equals_wave_private_and_oracle(dim, layout, normalize, eps):
    rng = np.random.default_rng(hash((dim, layout)) % 2**31)
    n_spans = 3000 if layout != "few_rows" else 70
    if layout == "tight":
        sizes = rng.integers(4, 61, n_spans); gaps = np.zeros(n_spans, np.int64)
    elif layout == "gaps":  # BOS / EOS rows between segments: covered fraction stays above 3/4
        sizes = rng.integers(4, 61, n_spans); gaps = (rng.random(n_spans) < 0.3) * rng.integers(1, 4, n_spans)
    elif layout == "ones":  # every row its own span: a span ends at every row of every tile
        sizes = np.ones(n_spans, np.int64); gaps = np.zeros(n_spans, np.int64)
    elif layout == "empties":  # np.mean of zero rows = NaN (`_embed.py:135` on an empty sentence), runs of them
        sizes = rng.integers(0, 9, n_spans) * (rng.random(n_spans) < 0.6); gaps = np.zeros(n_spans, np.int64)
    elif layout == "long":
        sizes = rng.choice([1, 3, 200, 700], n_spans // 10); gaps = np.zeros(len(sizes), np.int64)
    else:
        sizes = rng.integers(1, 3, n_spans); gaps = np.zeros(n_spans, np.int64)
    # b[0] = 5 + gaps[0] (the first span does not start at row 0), b[i] = e[i - 1] + gaps[i]
    b = np.concatenate(([5], 5 + np.cumsum(sizes[:-1] + gaps[1:]))).astype(np.int64) + gaps[0]
    e = b + sizes
    n_rows = int(e[-1]) + 3
    tokens = oracle.synth_matrix(9000 + dim, n_rows, dim)
    (f32, f16), (r32, r16) = _both(tokens, b, e, normalize=normalize, eps=eps)
    empty = sizes == 0
    assert np.isnan(f32[empty]).all() and np.isnan(r32[empty]).all()
    assert _same(f32[~empty], r32[~empty]) and _same(f16[~empty], r16[~empty])
    ref64, ref16 = oracle.pool_norm_cast(tokens, b, e, normalize=normalize, eps=eps or None)
    ulp = np.abs(f16[~empty].view(np.int16).astype(np.int32) - ref16[~empty].view(np.int16).astype(np.int32))
    assert ulp.max() <= 1 and (ulp > 0).mean() < 1e-3
    np.testing.assert_allclose(f32[~empty], ref64[~empty], rtol=2e-7, atol=1e-9)


@pytest.mark.parametrize("kind", ["reversed", "overlapping", "sparse", "shuffled"])
def test_span_lists_outside_the_cooperative_kernel_fall_back(kind):
    rng = np.random.default_rng(3)
    dim, n_spans = 1024, 500
    sizes = rng.integers(2, 30, n_spans)
    b = np.concatenate(([0], np.cumsum(sizes[:-1]))).astype(np.int64)
    e = b + sizes
    if kind == "reversed":
        b, e = b[::-1].copy(), e[::-1].copy()
    elif kind == "overlapping":
        e = np.minimum(e + 3, e[-1])
    elif kind == "sparse":  # spans cover a quarter of their row range
        b, e = b * 4, b * 4 + sizes
    else:
        p = rng.permutation(n_spans)
        b, e = b[p], e[p]
    tokens = oracle.synth_matrix(9100, int(e.max()) + 1, dim)
    (f32, f16), (r32, r16) = _both(tokens, b, e, normalize=True, eps=0.0)
    assert _same(f32, r32) and _same(f16, r16)
    ref64, _ = oracle.pool_norm_cast(tokens, b, e, normalize=True, eps=None)
    np.testing.assert_allclose(f32, ref64, rtol=2e-7, atol=1e-9)


def test_coop_device_pointers_and_integer_bit_exact(torch_cuda):
    """CUDA tensors in and out; integer-valued tokens: sums and means exact -> fp32 output equals the float64 oracle cast."""
    torch = torch_cuda
    rng = np.random.default_rng(8)
    dim, n_spans = 1024, 4000
    sizes = rng.integers(1, 40, n_spans)
    b = np.concatenate(([0], np.cumsum(sizes[:-1]))).astype(np.int64)
    e = b + sizes
    tok = oracle.synth_matrix(9200, int(e[-1]), dim, "small_int")
    os.environ["RAGLITE_POOL_COOP"] = "1"
    try:
        out = raglite_amd.pool_norm(torch.as_tensor(tok, device="cuda"), torch.as_tensor(b, device="cuda"), torch.as_tensor(e, device="cuda"),
                                    normalize=False, want_f32=True, want_f16=False)
    finally:
        del os.environ["RAGLITE_POOL_COOP"]
    f32 = out[0] if isinstance(out, tuple) else out
    ref64, _ = oracle.pool_norm_cast(tok, b, e, normalize=False, eps=None)
    np.testing.assert_allclose(f32.cpu().numpy(), ref64.astype(np.float32), rtol=1.2e-7, atol=0)  # (sum * (1 / n) vs sum / n: <= 1 fp64 ulp)
