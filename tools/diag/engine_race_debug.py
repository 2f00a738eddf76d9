#!/usr/bin/env python3
"""Debug: which step of the alternating-input loop first differs from the launches, and how."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime
lv = int(sys.argv[1]) if len(sys.argv) > 1 else 2
cfg = preset("bench"); tf, H = cfg.frame_num_tokens, cfg.lm.hidden_size
g = torch.Generator(device="cuda").manual_seed(1)
rt = Runtime(cfg, make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True), max_step_tokens=128, max_vit_frames=1)
st = rt.open_stream("static", 2048, 0)
rt.lm_step([st], (torch.randn(1, 20, H, generator=g, device="cuda") * 0.05).bfloat16())
Ts = [int(t) for t in sys.argv[2].split(",")] if len(sys.argv) > 2 else [tf, tf, 1, 48, tf]
xs = [(torch.randn(1, T, H, generator=g, device="cuda") * 0.05).bfloat16() for T in Ts]
rt.set_tuning("engine", 0)
ref = [tuple(t.clone() for t in rt.lm_step([st], x, want_hidden=True)) for x in xs]
reftaps = []
for i, x in enumerate(xs):
    rt.lm_step([st], x, want_hidden=True)
    reftaps.append({k: rt.debug_tap(k, 1, Ts[i]).clone() for k in ("h", "xn", "act", "attn_out")})
rt.set_tuning("engine", lv)
for kv in sys.argv[3:]:
    k, v = kv.split("="); rt.set_tuning(k, int(v))
ROUNDS = int(os.environ.get("ROUNDS", "6"))
for r in range(ROUNDS):
    for i, x in enumerate(xs):
        sc, hid = rt.lm_step([st], x, want_hidden=True)
        a = torch.equal(sc, ref[i][0]) and torch.equal(hid, ref[i][1])
        sc2 = rt.lm_step([st], x)
        b = torch.equal(sc2, ref[i][0])
        if not b:
            taps = {k: rt.debug_tap(k, 1, Ts[i]) for k in ("h", "xn", "act", "attn_out")}
            print("    taps of the LAST layer after the bad replay: " + "  ".join(f"{k}: {'same' if torch.equal(taps[k], reftaps[i][k]) else 'DIFF rows ' + str(sorted(set((taps[k] != reftaps[i][k]).any(1).nonzero().flatten().tolist()))[:8]) + ' n=' + str(int((taps[k] != reftaps[i][k]).sum()))}" for k in taps))
        print(f"round {r} input {i} T={Ts[i]}: direct {'ok' if a else 'DIFF'} (nan {bool(torch.isnan(sc).any())}, max|d| {(sc - ref[i][0]).abs().max().item():.3g}, hid d {(hid.float() - ref[i][1].float()).abs().max().item():.3g})  "
              f"replay {'ok' if b else 'DIFF'} (nan {bool(torch.isnan(sc2).any())}, max|d| {(sc2 - ref[i][0]).abs().max().item():.3g})", flush=True)
