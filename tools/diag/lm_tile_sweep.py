#!/usr/bin/env python3
"""The batched LM step's QKV / O projections (M = 288 rows, K = 3584) WITHOUT split-K on the tiled-GEMM variants (rows split over
64- or 128-row tiles, bf16 output, no fp32 slabs): the first measured version of VERDICT r4 item 1(a).  Weights cycle over 28 buffers.
    python tools/diag/lm_tile_sweep.py [variants] [M,...]"""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "aha-_amd", "libaha_amd.so"))
lib.aha_dev_gemm_tile.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [5, 14, 21, 2, 8]        # variants 3 / 4 / 6 / 7 / 9 were removed in round 6 (never selected)
Ms = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [288]
shapes = [(4608, 3584), (3584, 3584)]
NW = 28
st = torch.cuda.current_stream().cuda_stream
for M in Ms:
    for N, K in shapes:
        g = torch.Generator(device="cuda").manual_seed(M + N + K)
        A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
        Ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16() for _ in range(NW)]
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        line = []
        for r in range(40):
            for i in range(NW):
                lib.aha_dev_gemm_tile(A.data_ptr(), Ws[i].data_ptr(), C.data_ptr(), M, N, K, variants[0], st)
        torch.cuda.synchronize()
        for v in variants:
            rc = lib.aha_dev_gemm_tile(A.data_ptr(), Ws[0].data_ptr(), C.data_ptr(), M, N, K, v, st)
            assert rc == 0, (v, rc)
            torch.cuda.synchronize()
            err = (C.float() - (A.float() @ Ws[0].float().T)).abs().max().item()
            for i in range(NW):
                lib.aha_dev_gemm_tile(A.data_ptr(), Ws[i].data_ptr(), C.data_ptr(), M, N, K, v, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(3):
                for i in range(NW):
                    lib.aha_dev_gemm_tile(A.data_ptr(), Ws[i].data_ptr(), C.data_ptr(), M, N, K, v, st)
            e1.record(); e1.synchronize()
            line.append(f"v{v}:{e0.elapsed_time(e1) * 1e3 / (3 * NW):6.1f} (err {err:.3f})")
        print(f"M={M} N={N} K={K} | " + " ".join(line), flush=True)
