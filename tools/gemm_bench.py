"""Times the dense GEMM kernels on the engine's shapes (sbr_debug_gemm): exact-f32 MFMA vs bf16x6.
   python tools/gemm_bench.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbr_amd.engine import load_library
lib = load_library(os.environ.get("SBR_LIB"))
dev = torch.device("cuda")
SHAPES = [("c4 logits  NT", 256, 26744, 256, False, True), ("c4 dh      NN", 256, 256, 26744, False, False),
          ("c4 dW_out  TN", 26744, 256, 256, True, False), ("c4 wgrad   TN", 256, 1024, 51200, True, False),
          ("c2 logits  NT", 256, 3706, 128, False, True), ("c2 wgrad   TN", 128, 384, 51200, True, False),
          ("c2 dh      NN", 256, 128, 3712, False, False), ("c2 dW_out  TN", 3706, 128, 256, True, False),
          ("c3 l2 proj NN", 51200, 1024, 256, False, False)]
ws = torch.empty(1 << 26, device=dev)
for name, M, N, K, at, bt in SHAPES:
    A = torch.randn((K, M) if at else (M, K), device=dev); B = torch.randn((N, K) if bt else (K, N), device=dev)
    C = torch.empty((M, N), device=dev)
    sam, sak = (1, M) if at else (K, 1); sbk, sbn = (1, K) if bt else (N, 1)
    out = []
    for exact in (1, 0):
        def go():
            rc = lib.sbr_debug_gemm(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), A.data_ptr(), sam, sak, B.data_ptr(), sbk, sbn,
                                    C.data_ptr(), N, M, N, K, None, ws.data_ptr(), ws.numel(), exact)
            assert rc == 0
        for _ in range(3): go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): go()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100.0
        out.append("%s %7.1f us %6.1f TF" % ("f32" if exact else "x6 ", us, 2.0 * M * N * K / us / 1e6))
    print(name, "M=%d N=%d K=%d |" % (M, N, K), " | ".join(out))
