"""The default headline pipeline (approximate eight-query pass over the HI image + rigorous candidate bound + exact re-scoring) on
the data distribution RAGLite actually stores: unit-norm rows rounded through fp16 (src/raglite/_embed.py:138-140), iid and
clustered.  The bar is the literal north-star one -- scores within 1e-4 ABSOLUTE of float64, nothing of the float64 top-100 missed
beyond ties inside that tolerance -- at a size where the HI-image path engages (>= 64 M elements), and the candidate lists and the
fallback flag of the bound-filtered pipeline are part of what is asserted (they are data-dependent, not constants)."""

import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / "scripts"))


@pytest.mark.parametrize("kind", ["unit_fp16", "clustered"])
def test_headline_pipeline_on_raglite_shaped_data(kind):
    import bench_configs
    import torch

    import raglite_amd

    raglite_amd.set_device(0)
    out = bench_configs.shaped(kind, n=200_000, n_queries=24, steps=3)
    torch.cuda.empty_cache()
    assert out["arithmetic"] == "f16_split"
    assert out["filter"]["kind"] == "maxsim_batch_hi", out["filter"]  # the default path ran, not the small-index one
    chk = out["check"]
    assert chk["score_max_abs_err_vs_f64"] <= 1e-4, chk
    assert chk["recall_at_100_within_tol"] == 1.0, chk
    assert chk["queries_with_a_missed_chunk"] == 0, chk
    assert out["filter"]["candidates_per_query_max"] >= 100
    if kind == "unit_fp16":  # iid data: the bound decides, no fallback
        assert not out["filter"]["fallback"], out["filter"]
