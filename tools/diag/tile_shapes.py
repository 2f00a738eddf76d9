"""Tiled-GEMM variants on the tower's four shapes at 32 / 128 frames (dev hook aha_dev_gemm_tile): python tools/diag/tile_shapes.py"""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd import lib as L
lib = C.CDLL(L.LIB_PATH)
lib.aha_dev_gemm_tile.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
def bench(M, N, K, variant, n=12):
    A = [torch.randn(M, K, generator=g, device="cuda").bfloat16() for _ in range(2)]
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.03).bfloat16()
    Cc = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for i in range(2):
        if lib.aha_dev_gemm_tile(A[i % 2].data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, variant, st) != 0: return float("nan")
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): lib.aha_dev_gemm_tile(A[i % 2].data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, variant, st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
names = {1: "auto", 2: "256x128x64 ilv", 3: "128x128x64 4st", 5: "64x64", 8: "256x128x32 2/CU", 7: "256x256", 11: "288x128x32 2/CU"}
for frames in (8, 32, 64, 128):
    M = frames * 576
    for name, N, K in (("qkv", 3072, 1024), ("out", 1024, 1024), ("fc1", 4096, 1024), ("fc2", 1024, 4096)):
        row = []
        for v in (1, 2, 8, 11):
            t = bench(M, N, K, v)
            row.append(f"{names[v]} {t:7.1f} us ({2.0 * M * N * K / t / 1e6:5.0f} TF)")
        print(f"{frames:3d} frames {name}: " + " | ".join(row), flush=True)
