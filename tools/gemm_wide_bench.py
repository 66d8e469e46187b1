"""The wide-tile GEMM (gemm_x6w_kernel, 256 x 128 x 32) against the 128-wide gemm_x6_kernel on the layer GEMMs of C3 / C4 / C5, through
sbr_debug_gemm (mode 3 = fp16 split on whichever tile the library picks, 4 = the same arithmetic kept on the 128-wide tile; 2 / 5: plain
bf16 operands likewise): microseconds, f32-equivalent TFLOP/s, bitwise equality of the two results, error against float64 on sampled rows.
   python tools/gemm_wide_bench.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbr_amd.engine import load_library
lib = load_library(os.environ.get("SBR_LIB"))
dev = torch.device("cuda")
#        name                      M      N     K    A^T    B^T
SHAPES = [("c5 in-proj   NN", 51200, 2048, 512, False, False), ("c5 dx        NT", 51200, 512, 2048, False, True),
          ("c5 dW_in2    TN", 512, 2048, 51200, True, False), ("c5 dW_hid    TN", 512, 2048, 51200, True, False),
          ("c4 dW_hid    TN", 256, 1024, 51200, True, False), ("c3 l2 proj   NN", 51200, 1024, 256, False, False),
          ("c5 head-ish  NT", 256, 2048, 512, False, True)]
ws = torch.empty(1 << 27, device=dev)
g = torch.Generator(device="cuda").manual_seed(0)
for name, M, N, K, at, bt in SHAPES:
    A = torch.rand((K, M) if at else (M, K), device=dev, generator=g) * 2 - 1
    B = (torch.rand((N, K) if bt else (K, N), device=dev, generator=g) * 2 - 1) * 0.1
    sam, sak = (1, M) if at else (K, 1); sbk, sbn = (1, K) if bt else (N, 1)
    bias = torch.rand(N, device=dev, generator=g)
    res = {}
    line = []
    for mode in (4, 3, 5, 2):
        C = torch.zeros((M, N), device=dev)
        def go():
            rc = lib.sbr_debug_gemm(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), A.data_ptr(), sam, sak, B.data_ptr(), sbk, sbn,
                                    C.data_ptr(), N, M, N, K, ctypes.c_void_p(bias.data_ptr()), ws.data_ptr(), ws.numel(), mode)
            assert rc == 0
        for _ in range(2): go()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): go()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 200.0
        res[mode] = C
        rows = torch.arange(0, M, max(1, M // 64), device=dev)[:64]
        Ad = (A.t() if at else A)[rows].double(); Bd = (B.t() if bt else B).double()
        ref = Ad @ Bd + bias.double()
        err = ((C[rows].double() - ref).abs().max() / ref.abs().max()).item()
        line.append("%s %7.1f us %6.1f TF err %.1e" % ({4: "f16x3/128", 3: "f16x3/wide", 5: "bf16/128", 2: "bf16/wide"}[mode], us, 2.0 * M * N * K / us / 1e6, err))
    same = "bitwise %s / %s" % (torch.equal(res[3], res[4]), torch.equal(res[2], res[5]))
    print(name, "M=%d N=%d K=%d |" % (M, N, K), " | ".join(line), "|", same, flush=True)
