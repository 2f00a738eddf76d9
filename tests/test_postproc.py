"""aha_amd.postproc against the reference's own metric functions (test/tvsum/tvsum_utils.py,
test/hisum/hisum_eval.py: importable from /root/reference) - live when present, and through
tests/golden/postproc.json (made by tests/make_golden.py from those functions)."""
import json
import os
import sys

import numpy as np
import pytest

from conftest import GOLDEN, HAVE_REFERENCE, ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import make_golden as mg  # noqa: E402
import aha_amd  # noqa: E402,F401
from aha_amd import postproc as pp  # noqa: E402


def test_metrics_match_reference_golden():
    gold = json.load(open(os.path.join(GOLDEN, "postproc.json")))
    gt, pred = mg.postproc_inputs()
    m50, m15, top5, spe, ken = pp.evaluate_tvsum(gt, pred)
    got = {"mAP50": m50, "mAP15": m15, "top5": top5, "spearman": spe, "kendall": ken, "f1_15": pp.evaluate_f1(gt, pred)}
    for k, v in gold["tvsum"].items():
        assert got[k] == pytest.approx(v, abs=1e-9), k
    h = pp.hisum_evaluate_scores(gt, pred, spearman_kendall=True)
    for k, v in gold["hisum"].items():
        assert h[k] == pytest.approx(v, abs=1e-9), k


@pytest.mark.skipif(not HAVE_REFERENCE, reason="/root/reference not present (GPU box)")
def test_metrics_match_reference_live():
    sys.path.insert(0, "/root/reference")
    from test.tvsum import tvsum_utils as ref
    from test.hisum import hisum_eval as href
    for seed in (1, 2, 3):
        gt, pred = mg.postproc_inputs(seed=seed, n_videos=3)
        np.testing.assert_allclose(pp.evaluate_tvsum(gt, pred), ref.evaluate_tvsum(gt, pred), rtol=0, atol=1e-9)
        assert pp.evaluate_f1(gt, pred) == pytest.approx(ref.evaluate_f1(gt, pred), abs=1e-12)
        a, b = pp.hisum_evaluate_scores(gt, pred, True), href.hisum_evaluate_scores(gt, pred, True, print_logs=False)
        for k in b:
            assert a[k] == pytest.approx(b[k], abs=1e-9), k
        for v in gt:
            assert pp.map_at_rho(gt[v], pred[v], 0.15) == pytest.approx(ref.map_at_rho(gt[v], pred[v], 0.15), abs=1e-12)
            assert np.array_equal(pp.binarize_gt(gt[v], 0.5), ref.binarize_gt(gt[v], 0.5))


def test_fuse_scores_and_knapsack():
    rows = [{"informative_score": 0.2, "relevance_score": 0.7, "uncertainty_score": 0.03},
            {"informative_score": 0.9, "relevance_score": 0.1, "uncertainty_score": 0.10}]
    # outputs/grid_search_params.json "tvsum": alpha 0, beta -1, epsilon -5, threshold 0.04 (test/evaluate.py:584-589)
    s = pp.fuse_scores(rows, 0.0, -1.0, -5.0, 0.04)
    assert s[0] == pytest.approx(-0.7) and s[1] == pytest.approx(-0.1 + 5.0 * 0.06)
    rng = np.random.RandomState(0)
    frames = [{"idx": i, "informative_score": float(rng.rand()), "relevance_score": float(rng.rand()),
               "uncertainty_score": float(rng.rand())} for i in range(40)]
    sel = pp.knapsack_selection(frames, 7, 1.0, 1.0, 0.5, -0.2)
    vals = np.array([f["informative_score"] + 0.5 * f["relevance_score"] - 0.2 * f["uncertainty_score"] for f in frames])
    assert sel == set(np.argsort(vals)[-7:].tolist())          # unit costs: the knapsack optimum is the top-k
    assert pp.knapsack_selection(frames, 0, 1.0, 1.0, 0.5, -0.2) == set()


def test_primitives_edge_cases():
    assert np.isnan(pp.average_precision([0, 0, 0], [0.1, 0.2, 0.3]))
    assert pp.average_precision([1, 0, 1, 0], [0.9, 0.8, 0.7, 0.1]) == pytest.approx(0.5 * 1.0 + 0.5 * (2 / 3))
    assert pp.average_precision([1, 0], [0.5, 0.5]) == pytest.approx(0.5)          # a tie shares one threshold
    assert np.array_equal(pp.rank_average([10, 20, 10, 30]), [1.5, 3.0, 1.5, 4.0])
    assert pp.kendall_tau_b([1, 2, 3, 4], [1, 2, 3, 4]) == pytest.approx(1.0)
    assert pp.kendall_tau_b([1, 2, 3, 4], [4, 3, 2, 1]) == pytest.approx(-1.0)
    assert np.isnan(pp.spearman_rho([1, 1, 1], [1, 2, 3]))
