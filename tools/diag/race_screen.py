#!/usr/bin/env python3
"""Race screen of the hand-synchronised round-3 kernels (cdna_hip_programming.md: a sync-structure edit makes a new template - screen
it over many runs at several sizes): every launch's output is compared bit for bit with the result of an independently synchronised
kernel of the same arithmetic.  python tools/diag/race_screen.py [rounds]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import aha_amd
from aha_amd.config import preset
from aha_amd.synth import make_weights
from aha_amd.runtime import Runtime
lib = ctypes.CDLL(os.path.join(ROOT, "aha-_amd", "libaha_amd.so"))
lib.aha_dev_gemm_tile.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
st = torch.cuda.current_stream().cuda_stream
bad = total = 0
g = torch.Generator(device="cuda").manual_seed(1)
# persistent 288x256 (variant 12) and the pipelined 64x64 (14), each against an independently synchronised kernel (register-staged or 32-deep DMA tile)
for M, N, K in [(18432, 3072, 1024), (18432, 1024, 4096), (9792, 4096, 1024), (18432 + 100, 1152, 640), (4608, 3584, 3584), (73728, 1024, 1024),
                (576, 1024, 4096), (576, 3072, 1024), (1731, 1024, 1024)]:
    A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
    W = (torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16()
    ref = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    out = torch.empty_like(ref)
    assert lib.aha_dev_gemm_tile(A.data_ptr(), W.data_ptr(), ref.data_ptr(), M, N, K, 0 if M < 4000 else 8, st) == 0
    for variant, piped in ((12, 1), (14, 1)):
        if variant == 14 and M > 2304: continue
        n_bad = 0
        for r in range(rounds):
            out.fill_(7.0)
            assert lib.aha_dev_gemm_tile(A.data_ptr(), W.data_ptr(), out.data_ptr(), M, N, K, variant, st) == 0
            n_bad += int(not torch.equal(out, ref))
        total += rounds; bad += n_bad
        print(f"gemm M={M:6d} N={N:5d} K={K:5d} variant {variant}{'s' if piped and variant == 12 else ''}: {n_bad} of {rounds} launches differ", flush=True)
# head-resident dense attention against the restaging kernel
cfg = preset("tiny")
rt = Runtime(cfg, make_weights(cfg, device="cuda", dtype=torch.bfloat16), max_step_tokens=64, max_vit_frames=1, max_positions=256)
for n, T in ((32, 576), (9, 500), (64, 576)):
    qkv = torch.randn(n, T, 3 * 1024, generator=g, device="cuda").bfloat16()
    rt.set_tuning("attn_head", 0); ref = rt.vit_attention(qkv, 16, 64).clone()
    rt.set_tuning("attn_head", 2)
    n_bad = sum(int(not torch.equal(rt.vit_attention(qkv, 16, 64), ref)) for _ in range(rounds))
    total += rounds; bad += n_bad
    print(f"attn_head64 n={n} T={T}: {n_bad} of {rounds} launches differ", flush=True)
# [r5] the latency path's row-group form of the head-resident kernel (4-wave workgroups, chosen automatically from one to seven frames)
for n, T in ((1, 576), (2, 577), (4, 576), (7, 500)):
    qkv = torch.randn(n, T, 3 * 1024, generator=g, device="cuda").bfloat16()
    rt.set_tuning("attn_head", 0); ref = rt.vit_attention(qkv, 16, 64).clone()
    rt.set_tuning("attn_head", 1)
    n_bad = sum(int(not torch.equal(rt.vit_attention(qkv, 16, 64), ref)) for _ in range(rounds))
    total += rounds; bad += n_bad
    print(f"attn_head64 row groups n={n} T={T}: {n_bad} of {rounds} launches differ", flush=True)
rt.close()
# [r5] whole encodes: k-blocked weight twins and activations through the persistent tile kernel, LayerNorm launches carrying prefetch riders,
# against the same tower with row-major operands, no riders and the restaging attention
from aha_amd.config import LiveConfig, LMConfig
from aha_amd.synth import make_frames
cfg = LiveConfig(lm=LMConfig(num_hidden_layers=1, vocab_size=1024), name="vit24")
rt = Runtime(cfg, make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True), max_step_tokens=64, max_vit_frames=32)
for n in (32, 8, 1, 3):
    fr = make_frames(n, cfg.vision.image_size, seed=n).cuda()
    for k, v in (("tile_wkb", 0), ("vit_akb", 0), ("vit_prefetch", 0), ("attn_head", 0)): rt.set_tuning(k, v)
    ref = rt.visual_embed(fr).clone()
    for k, v in (("tile_wkb", 1), ("vit_akb", 1), ("vit_prefetch", 2400), ("attn_head", 1)): rt.set_tuning(k, v)
    n_bad = sum(int(not torch.equal(rt.visual_embed(fr), ref)) for _ in range(rounds))
    total += rounds; bad += n_bad
    print(f"vision encode n={n} (k-blocked operands, riders, head-resident attention vs row-major, none, restaging): {n_bad} of {rounds} encodes differ", flush=True)
rt.close()
# [r6] the two single-launch MLP forms (tuning "engine": 1 = LDS-DMA loader ring with in-launch row phase, 2 = register-streaming gate/up -> down_proj):
# persistent workgroups that hand tiles to each other through write-through stores + counters - the most race-prone code in the library.  Every step's
# scores AND final hidden row against the launches', full model, direct launches (hidden requested) and graph replay, inputs alternating so that a stale
# hand-off buffer of the previous step would show; once idle and once with a second stream keeping the memory system busy (cdna_hip_programming.md G16:
# "test under uniform AND uneven load").
cfg = preset("bench"); tf, H = cfg.frame_num_tokens, cfg.lm.hidden_size
rt = Runtime(cfg, make_weights(cfg, device="cuda", dtype=torch.bfloat16, skip_lm_head=True), max_step_tokens=128, max_vit_frames=1)
st = rt.open_stream("static", 2048, 0)
rt.lm_step([st], (torch.randn(1, 20, H, generator=g, device="cuda") * 0.05).bfloat16())
xs = [(torch.randn(1, T, H, generator=g, device="cuda") * 0.05).bfloat16() for T in (tf, tf, 1, 48, tf)]
rt.set_tuning("engine", 0)
ref = [tuple(t.clone() for t in rt.lm_step([st], x, want_hidden=True)) for x in xs]
noise_src = torch.empty(64 << 20, dtype=torch.uint8, device="cuda"); noise_dst = torch.empty_like(noise_src)
side = torch.cuda.Stream()
for lv in (1, 2):
    rt.set_tuning("engine", lv)
    for load in (False, True):
        n_bad = 0
        for r in range(rounds):
            if load:
                with torch.cuda.stream(side):
                    for _ in range(4): noise_dst.copy_(noise_src)
            for i, x in enumerate(xs):
                sc, hid = rt.lm_step([st], x, want_hidden=True)              # direct launches
                n_bad += int(not (torch.equal(sc, ref[i][0]) and torch.equal(hid, ref[i][1])))
                sc2 = rt.lm_step([st], x)                                      # graph replay from the second time on
                n_bad += int(not torch.equal(sc2, ref[i][0]))
        torch.cuda.synchronize()
        total += rounds * len(xs) * 2; bad += n_bad
        print(f"layer engine {lv} ({'beside a copy stream' if load else 'alone'}): {n_bad} of {rounds * len(xs) * 2} steps differ from the launches", flush=True)
rt.set_tuning("engine", 0)
print(f"RACE SCREEN: {bad} differing launches of {total}")
sys.exit(1 if bad else 0)
