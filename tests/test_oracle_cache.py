"""Pin oracle/cache_policies.py: (a) against the golden outputs of the reference's own cache
classes (tests/golden/cache_policies.npz, made by tests/make_golden.py from
/root/reference/test/{sink,sliding_window,static}_cache.py), (b) live against those classes
when /root/reference is present."""
import os
import sys
from unittest import mock

import numpy as np
import pytest
import torch

from conftest import GOLDEN, HAVE_REFERENCE, ROOT

sys.path.insert(0, os.path.join(ROOT, "tests"))       # make_golden.py lives beside this file
import make_golden as mg  # noqa: E402
from oracle.cache_policies import SinkPolicy, SlidingPolicy, StaticPolicy, GrowingPolicy, make_policy  # noqa: E402


def _mine(name):
    return {"sink": SinkPolicy(mg.CACHE_W, mg.CACHE_SINK), "sliding": SlidingPolicy(mg.CACHE_W),
            "static": StaticPolicy(mg.CACHE_W)}[name]


@pytest.mark.parametrize("name", ["sink", "sliding", "static"])
def test_policy_matches_reference_golden(name):
    gold = np.load(os.path.join(GOLDEN, "cache_policies.npz"))
    pol = _mine(name)
    for step, (T, layers) in enumerate(mg.cache_inputs()):
        L = pol.get_seq_length()
        assert L == int(gold[f"{name}_len_before"][step])
        cos, sin = mg.rope_table((L + torch.arange(T))[None], mg.CACHE_D, mg.CACHE_THETA, torch.bfloat16)
        for l, (k, v) in enumerate(layers):
            kr, vr = pol.update(k, v, l, {"cos": cos, "sin": sin})
            key = f"{name}_k_s{step}_l{l}"
            if key in gold.files:
                assert np.array_equal(mg.bf16_bits(kr), gold[key])          # bit-exact
                assert np.array_equal(mg.bf16_bits(vr), gold[f"{name}_v_s{step}_l{l}"])
    assert pol.get_seq_length() == int(gold[f"{name}_len_final"])


@pytest.mark.parametrize("name", ["sink", "sliding"])
def test_policy_matches_the_reference_at_the_benchmark_geometry(name):
    """tests/golden/cache_bench.npz: the reference's own SinkCache / SlidingWindowCache at W 2048, sink 32, head_dim 128, theta 1e6,
    4 KV heads over 122 steps (a key lives its whole ~55 bf16 re-rotations): the oracle returns the same K and V bits at EVERY step
    (sha256 per step and layer), the sampled rows included."""
    gold = np.load(os.path.join(GOLDEN, "cache_bench.npz"))
    pol = SinkPolicy(mg.BENCH_W, mg.BENCH_SINK) if name == "sink" else SlidingPolicy(mg.BENCH_W)
    for step, (T, layers) in enumerate(mg.bench_cache_inputs()):
        L = pol.get_seq_length()
        assert L == int(gold[f"{name}_len_before"][step])
        cos, sin = mg.rope_table((L + torch.arange(T))[None], mg.BENCH_D, mg.BENCH_THETA, torch.bfloat16)
        for l, (k, v) in enumerate(layers):
            kr, vr = pol.update(k, v, l, {"cos": cos, "sin": sin})
            assert np.array_equal(mg.kv_digest(kr), gold[f"{name}_digest"][step, l, 0]), (name, step, l, "K")
            assert np.array_equal(mg.kv_digest(vr), gold[f"{name}_digest"][step, l, 1]), (name, step, l, "V")
            if step in mg.BENCH_SAMPLE_STEPS:
                rows = [r for r in mg.BENCH_SAMPLE_ROWS if r < kr.shape[2]]
                assert np.array_equal(mg.bf16_bits(kr[0, 1, rows]), gold[f"{name}_k_s{step}_l{l}"])
                assert np.array_equal(mg.bf16_bits(vr[0, 1, rows]), gold[f"{name}_v_s{step}_l{l}"])
    assert pol.get_seq_length() == int(gold[f"{name}_len_final"]) == mg.BENCH_W


@pytest.mark.skipif(not HAVE_REFERENCE, reason="/root/reference not present (GPU box)")
def test_policy_matches_reference_live():
    sys.path.insert(0, "/root/reference")
    from transformers import Cache
    with mock.patch.object(Cache, "__init__", lambda s, *a, **k: None):
        from test.sink_cache import SinkCache
        from test.sliding_window_cache import SlidingWindowCache
        from test.static_cache import TrulyStaticCache
        for ref, mine in ((SinkCache(48, 4), SinkPolicy(48, 4)), (SlidingWindowCache(40), SlidingPolicy(40)),
                          (TrulyStaticCache(16), StaticPolicy(16))):
            g = torch.Generator().manual_seed(7)
            for T in [18, 5, 5, 5, 11, 5, 5, 5, 5, 3, 5, 5]:
                assert ref.get_seq_length() == mine.get_seq_length()
                cos, sin = mg.rope_table((ref.get_seq_length() + torch.arange(T))[None], 16, 1e6, torch.bfloat16)
                for l in range(2):
                    k = torch.randn(1, 2, T, 16, generator=g).bfloat16()
                    v = torch.randn(1, 2, T, 16, generator=g).bfloat16()
                    kr, vr = ref.update(k, v, l, {"cos": cos, "sin": sin})
                    km, vm = mine.update(k, v, l, {"cos": cos, "sin": sin})
                    assert torch.equal(kr, km) and torch.equal(vr, vm)


def test_sequence_lengths_match_survey_8c():
    # [1,2,3,4] K/V, W=8, sink=2: static stays 3; sliding 3->6->8->8; sink 3->6->8->8
    for pol, want in ((StaticPolicy(8), [3, 3, 3, 3]), (SlidingPolicy(8), [3, 6, 8, 8]), (SinkPolicy(8, 2), [3, 6, 8, 8])):
        got = []
        for _ in range(4):
            L = pol.get_seq_length()
            cos, sin = mg.rope_table((L + torch.arange(3))[None], 4, 1e4, torch.float32)
            pol.update(torch.randn(1, 2, 3, 4), torch.randn(1, 2, 3, 4), 0, {"cos": cos, "sin": sin})
            got.append(pol.get_seq_length())
        assert got == want
    p = SinkPolicy(8, 2)
    for _ in range(5):
        cos, sin = mg.rope_table((p.get_seq_length() + torch.arange(3))[None], 4, 1e4, torch.float32)
        p.update(torch.randn(1, 2, 3, 4), torch.randn(1, 2, 3, 4), 0, {"cos": cos, "sin": sin})
    assert p._seen_tokens == 15


def test_make_policy_selection():
    assert isinstance(make_policy("default_sink"), SinkPolicy)
    assert isinstance(make_policy("sliding_window"), SlidingPolicy)
    assert isinstance(make_policy("static"), StaticPolicy)
    assert isinstance(make_policy(None), GrowingPolicy)
    with pytest.raises(ValueError):
        make_policy("bogus")
