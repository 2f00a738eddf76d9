"""Fixed vs per-k cost of the tiled GEMM variants on the tower's shapes: time(K) for K = 128 .. 2048 (dev hook aha_dev_gemm_tile)."""
import os, sys, ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import aha_amd
from aha_amd import lib as L
lib = C.CDLL(L.LIB_PATH)
lib.aha_dev_gemm_tile.argtypes = [C.c_void_p] * 3 + [C.c_int] * 4 + [C.c_void_p]
g = torch.Generator(device="cuda").manual_seed(0)
def bench(M, N, K, variant, n=20):
    A = [torch.randn(M, K, generator=g, device="cuda").bfloat16() for _ in range(3)]
    W = (torch.randn(N, K, generator=g, device="cuda") * 0.03).bfloat16()
    Cc = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for i in range(3): assert lib.aha_dev_gemm_tile(A[i % 3].data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, variant, st) == 0
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): lib.aha_dev_gemm_tile(A[i % 3].data_ptr(), W.data_ptr(), Cc.data_ptr(), M, N, K, variant, st)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
M = 18432
for name, N in (("N=3072", 3072), ("N=1024", 1024)):
    for v, lab in ((8, "dma32 3st"), (2, "256x128x64 ilv"), (5, "64x64")):
        ts = [bench(M, N, K, v) for K in (128, 256, 512, 1024, 2048)]
        print(f"{name} {lab:16s} K=128..2048: " + "  ".join(f"{t:7.1f}" for t in ts) + f"   per 64-k: {(ts[4] - ts[3]) / 16:.2f} us, fixed ~{ts[0] - 2 * (ts[4] - ts[3]) / 16:.1f} us", flush=True)
