"""Per-shape sweep of the tiled-GEMM variants through the developer hook aha_dev_gemm_tile (variant 12 = persistent 288x256; 13 is
accepted as an alias from the time a plain-loop form of that kernel existed).
Weights cycle over 24 distinct buffers (a tower's worth), so they stream from HBM as in the real encode."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
lib = ctypes.CDLL(os.path.join(ROOT, "aha-_amd", os.environ.get("AHA_SWEEP_LIB", "libaha_amd.so")))
lib.aha_dev_gemm_tile.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 + [ctypes.c_void_p]
_raw = lib.aha_dev_gemm_tile
def dev_gemm(A, W, C, M, N, K, v, st):
    return _raw(A, W, C, M, N, K, 12 if v == 13 else v, st)
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [0, 1, 2, 5, 8, 14]
Ms = [int(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [576, 1152, 2304, 4608, 18432]
shapes = [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]
if len(sys.argv) > 3 and sys.argv[3] == "so400m":
    shapes = [(3456, 1152), (1152, 1152), (4304, 1152), (1152, 4352)]
NW = 24
st = torch.cuda.current_stream().cuda_stream
for M in Ms:
    for N, K in shapes:
        g = torch.Generator(device="cuda").manual_seed(M + N + K)
        A = (torch.randn(M, K, device="cuda", generator=g) * 0.5).bfloat16()
        Ws = [(torch.randn(N, K, device="cuda", generator=g) * 0.05).bfloat16() for _ in range(NW)]
        C = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
        ref, line = None, []
        # settle the shape first: the variant timed first used to read 15-25 % slow (131 vs 106 us on the 32-frame QKV shape) - cold
        # buffers and a clock that has not yet settled under load; ~0.3 s of launches of the first variant are thrown away
        for r in range(100):
            for i in range(NW):
                dev_gemm(A.data_ptr(), Ws[i].data_ptr(), C.data_ptr(), M, N, K, variants[0], st)
            if r % 10 == 9: torch.cuda.synchronize()
        for v in variants:
            C.zero_()
            rc = dev_gemm(A.data_ptr(), Ws[0].data_ptr(), C.data_ptr(), M, N, K, v, st)
            assert rc == 0, (v, rc)
            torch.cuda.synchronize()
            if ref is None:
                ref = C.clone()
                exact = (ref.float() - (A.float() @ Ws[0].float().T)).abs().max().item()
            same = torch.equal(ref, C)
            for i in range(NW):
                dev_gemm(A.data_ptr(), Ws[i].data_ptr(), C.data_ptr(), M, N, K, v, st)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for r in range(2):
                for i in range(NW):
                    dev_gemm(A.data_ptr(), Ws[i].data_ptr(), C.data_ptr(), M, N, K, v, st)
            e1.record(); e1.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / (2 * NW)
            line.append(f"v{v}:{us:6.1f}{'' if same else '!'}")
        best = min(line, key=lambda s: float(s.split(":")[1].rstrip("!")))
        print(f"M={M:6d} N={N:5d} K={K:5d} err {exact:.3f} | " + " ".join(line) + f" | best {best.split(':')[0]} {2*M*N*K/float(best.split(':')[1].rstrip('!'))/1e6:.0f} TF/s", flush=True)
