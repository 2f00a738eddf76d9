"""CPU evidence that the ``regime="stable"`` weight set (aha_amd.synth.stable_regime_scale) makes the REFERENCE arithmetic
itself stable at full width and depth, so that a flat 1e-3 on free-running scores is a property an implementation can be
held to (tests/test_gpu_flat_parity.py holds the HIP path to it).

Runs the oracle three ways on the same frame sequence - bf16 sdpa, bf16 eager (two equally valid evaluations of the
reference: models/arguments_live.py:30 selects the attention implementation), float32 - and reports, per score column,
    |bf16 sdpa - bf16 eager|, |bf16 sdpa - fp32|, the spread (std) of the scores over the frames.
Pass criteria: both distances <= 5e-4, spread >= 0.02 on the informative / relevance columns.

    python tests/stable_regime_check.py [--frames 64] [--regime stable] [--policy none|default_sink] [--out profiles/...json]

Test infrastructure: imports oracle/.  Takes minutes at 7B dims (13 GB of bf16 weights + 26 GB fp32) - not a pytest case;
tests/test_oracle_models.py::test_stable_regime_is_stable_on_a_narrow_deep_model runs the same check on a narrow
28-layer model in seconds.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import aha_amd  # noqa: E402,F401
from aha_amd.config import preset  # noqa: E402
from aha_amd.synth import make_weights  # noqa: E402


def synth_embeds(n_frames, T, H, seed=7, prefix=20, scale=1.0):
    """Stand-in frame embeddings at the scale the stable regime's projector produces (unit rms); frame 0 carries a prefix."""
    g = torch.Generator().manual_seed(seed)
    out = []
    for i in range(n_frames):
        t = T + (prefix if i == 0 else 0)
        out.append((torch.randn(1, t, H, generator=g) * scale).bfloat16())
    return out


def run(olm, steps, policy):
    from oracle.cache_policies import make_policy
    from oracle.qwen2_live import frame_scores
    pol = make_policy(*policy)
    sc = []
    for x in steps:
        sc.append(frame_scores(olm.step(x.to(olm.dtype), pol))[0])
    return torch.stack(sc).double()


def stability_stats(cfg, w, steps, policy=(None, 0, 0), want_fp32=True, log=print):
    from oracle.qwen2_live import OracleLM
    t0 = time.time()
    a = run(OracleLM(cfg.lm, w, torch.bfloat16, attn_impl="sdpa"), steps, policy)
    log(f"  bf16 sdpa  {time.time() - t0:.1f}s")
    t0 = time.time()
    b = run(OracleLM(cfg.lm, w, torch.bfloat16, attn_impl="eager"), steps, policy)
    log(f"  bf16 eager {time.time() - t0:.1f}s")
    st = {"frames": len(steps), "policy": list(policy),
          "sdpa_vs_eager_max": (a - b).abs().max(0).values.tolist(), "sdpa_vs_eager_median": (a - b).abs().median(0).values.tolist(),
          "score_std": a.std(0).tolist(), "score_mean": a.mean(0).tolist(), "score_min": a.min(0).values.tolist(),
          "score_max": a.max(0).values.tolist()}
    if want_fp32:
        t0 = time.time()
        o32 = OracleLM(cfg.lm, w, torch.float32)
        c = run(o32, [x.float() for x in steps], policy)
        del o32
        log(f"  fp32       {time.time() - t0:.1f}s")
        st["bf16_vs_fp32_max"] = (a - c).abs().max(0).values.tolist()
        st["bf16_vs_fp32_median"] = (a - c).abs().median(0).values.tolist()
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=64)
    ap.add_argument("--regime", default="stable")
    ap.add_argument("--preset", default="bench")
    ap.add_argument("--policy", default="none")
    ap.add_argument("--window", type=int, default=1024)
    ap.add_argument("--sink", type=int, default=32)
    ap.add_argument("--no-fp32", action="store_true")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    torch.set_num_threads(os.cpu_count() or 8)
    cfg = preset(a.preset)
    w = make_weights(cfg, dtype=torch.bfloat16, skip_lm_head=True, regime=a.regime)
    w = {k: v for k, v in w.items() if not k.startswith(("vision.", "mm_projector"))}
    steps = synth_embeds(a.frames, cfg.frame_num_tokens, cfg.lm.hidden_size)
    policy = (None, 0, 0) if a.policy == "none" else (a.policy, a.window, a.sink)
    st = stability_stats(cfg, w, steps, policy, want_fp32=not a.no_fp32)
    st.update(regime=a.regime, preset=a.preset, threads=torch.get_num_threads())
    print(json.dumps(st, indent=1))
    if a.out:
        json.dump(st, open(os.path.join(ROOT, a.out), "w"), indent=1)


if __name__ == "__main__":
    main()
