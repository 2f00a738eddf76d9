#!/usr/bin/env python3
"""Mid-M GEMM (gemm_wl) with X row-major vs k-blocked ([K/32][M][32]: contiguous 64-byte-per-row panels per k-step), 7B shapes.
python tools/diag/wl_xkb.py"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd import lib as _l
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime, Linear, _cur_stream

cfg = preset("tiny")
rt = Runtime(cfg, make_weights(cfg, device="cuda", dtype=torch.bfloat16))
g = torch.Generator(device="cuda").manual_seed(1)
H, I = 3584, 18944


def rnd(*s):
    return (torch.randn(*s, generator=g, device="cuda") * 0.05).bfloat16()


def run(lin, x, ldx, M, epi, out, sk, n=40):
    def once(i):
        l = lin[i % len(lin)]
        rt._chk(rt.lib.aha_linear_forward(rt.ctx, l.handle, x.data_ptr(), ldx, M, epi, sk, None, out.data_ptr(), l.N, _cur_stream()))
    for i in range(6):
        once(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n):
        once(i)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


shapes = [("gate_up", I, H, True, _l.EPI_SWIGLU, 2, 1), ("down", H, I, False, _l.EPI_SPLITK_F32, 3, 8), ("qkv", 4608, H, False, _l.EPI_SPLITK_F32, 6, 7),
          ("o", H, H, False, _l.EPI_SPLITK_F32, 6, 8)]          # split-K factors as aha_lm_step picks them
for name, N, K, pair, epi, copies, SK in shapes:
    lins = [Linear(rt, rnd(N, K), rnd(N, K) if pair else None) for _ in range(copies)]
    for M in (288, 160, 320):
        x = rnd(M, K)
        xkb = x.view(M, K // 32, 32).permute(1, 0, 2).contiguous()
        S = rt.lib.aha_linear_split_k(rt.ctx, lins[0].handle, SK) if epi == _l.EPI_SPLITK_F32 else 1
        o0 = torch.zeros((S, M, N), dtype=torch.float32 if epi == _l.EPI_SPLITK_F32 else torch.bfloat16, device="cuda")
        o1 = torch.zeros_like(o0)
        rt.set_tuning("dev_xkb", 0); t0 = run(lins, x, K, M, epi, o0, SK)
        rt.set_tuning("dev_xkb", 1); t1 = run(lins, xkb, M, M, epi, o1, SK)
        rt.set_tuning("dev_xkb", 0)
        pads = []
        for pad in (32, 64, 128, 192):
            xp = torch.zeros(M, K + pad, dtype=torch.bfloat16, device="cuda"); xp[:, :K] = x
            o2 = torch.zeros_like(o0)
            pads.append((pad, run(lins, xp, K + pad, M, epi, o2, SK), torch.equal(o0, o2)))
        # same weight on the last call of both runs -> outputs must be bit-identical
        same = torch.equal(o0, o1)
        print(f"{name:8s} M={M}: row-major {t0:6.1f} us   k-blocked {t1:6.1f} us   bit-identical {same}   padded ld: " + "  ".join(f"+{p_}: {t_:.1f}{'' if ok_ else ' MISMATCH'}" for p_, t_, ok_ in pads), flush=True)
    for l in lins:
        l.close()
