"""Three pure functions of the reference, pinned on the reference's OWN code: tests/make_golden.py compiles each function definition
unmodified from its source text (`ast`, no import of the file) and stores outputs only (tests/golden/ref_pooling.npz,
ref_functions.json).  Here the oracle's restatement and the product's host code must reproduce them:
  post_projector_pooling   /root/reference/models/live_llava/video_head_live_llava_qwen.py:117-136
  round_numbers / truncate_sig   /root/reference/test/inference.py:359-375
  find_ticks               /root/reference/test/live_infer_for_video.py:195-228
  knapsack_selection       /root/reference/test/highlight_generator.py:8-37"""
import json
import os
import sys
import types

import numpy as np
import pytest
import torch

from conftest import GOLDEN, ROOT

sys.path.insert(0, os.path.join(ROOT, "tests"))
import make_golden as mg  # noqa: E402


@pytest.mark.parametrize("grid,stride,mode", mg.POOL_CASES)
def test_oracle_pooling_reproduces_the_reference_function(grid, stride, mode):
    from oracle.vision_tower import OracleVision
    gold = np.load(os.path.join(GOLDEN, "ref_pooling.npz"))
    ov = OracleVision.__new__(OracleVision)                      # the method reads two config knobs and the grid only
    ov.cfg = types.SimpleNamespace(video_pooling_stride=stride, mm_spatial_pool_mode=mode)
    ov.v = types.SimpleNamespace(grid=grid)
    x = mg.pool_input(grid)
    assert np.array_equal(ov.post_projector_pooling(x).numpy(), gold[f"{mode}_{grid}_f32"])
    assert np.array_equal(mg.bf16_bits(ov.post_projector_pooling(x.bfloat16())), gold[f"{mode}_{grid}_bf16"])


def test_round_numbers_of_product_and_oracle_reproduce_the_reference_function():
    gold = json.load(open(os.path.join(GOLDEN, "ref_functions.json")))
    import aha_amd  # noqa: F401
    from aha_amd.live_infer import round_numbers as product_round
    from oracle.live_driver import round_numbers as oracle_round
    for fn in (product_round, oracle_round):
        got = [fn(v, 3) for v in mg.ROUND_TABLE]
        assert got == gold["round_numbers_3"], fn.__module__
        assert [type(v).__name__ for v in got] == gold["round_types"], fn.__module__       # an exact zero becomes the int 0


def test_find_ticks_of_the_product_reproduces_the_reference_function():
    gold = json.load(open(os.path.join(GOLDEN, "ref_functions.json")))
    import aha_amd  # noqa: F401
    from aha_amd.live_infer import LiveInferForDemo
    for case in gold["find_ticks"]:
        scores = mg.ticks_input(case["case"])
        assert [float(t) for t in LiveInferForDemo.find_ticks(None, scores, case["fps"])] == case["peaks"]
        assert [float(t) for t in LiveInferForDemo.find_ticks(None, list(scores), case["fps"], min_separation=3)] == case["peaks"]   # the reference overrides the argument


def test_knapsack_selection_of_the_product_reproduces_the_reference_function():
    """Unit-cost 0/1 knapsack over alpha * informative + beta * relevance + epsilon * uncertainty with the reference's tie rule (a frame is
    taken iff it changed the DP cell, walking back from the last frame): same selected index sets on seeded rows with tied scores."""
    gold = json.load(open(os.path.join(GOLDEN, "ref_functions.json")))
    import aha_amd  # noqa: F401
    from aha_amd import postproc as pp
    for (i, budget, w, al, be, ep), want in zip(mg.KNAPSACK_CASES, gold["knapsack_selection"]):
        got = pp.knapsack_selection(mg.knapsack_input(i), budget, w, al, be, ep)
        assert sorted(int(v) for v in got) == want, (i, budget)
        assert len(want) == min(budget, len(mg.knapsack_input(i))) or any(v <= 0 for v in (al, be, ep))
