"""Phase counters of gemm_x6w_kernel (probe build -DX6W_DBG=9: SBR_LIB=tools/probes/variants/libsbr_x6w9.so): shader cycles per k step of
workgroup 8, by wave: fragments landed | first half (waves 0-3: split + LDS write, 4-7: MFMAs) | second half | barrier."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from sbr_amd.engine import load_library
lib = load_library(os.environ.get("SBR_LIB"))
dev = torch.device("cuda")
ws = torch.empty(1 << 26, device=dev)
for name, M, N, K, at, bt, mode in (("c5 in-proj f16x3", 51200, 2048, 512, False, False, 3), ("c5 in-proj bf16", 51200, 2048, 512, False, False, 2),
                                    ("c5 dW TN f16x3", 512, 2048, 51200, True, False, 3)):
    A = torch.rand((K, M) if at else (M, K), device=dev); B = torch.rand((N, K) if bt else (K, N), device=dev)
    sam, sak = (1, M) if at else (K, 1); sbk, sbn = (1, K) if bt else (N, 1)
    C = torch.zeros((M, N), device=dev)
    prof = torch.zeros(64, dtype=torch.int64, device=dev)
    for _ in range(2):
        rc = lib.sbr_debug_gemm(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), A.data_ptr(), sam, sak, B.data_ptr(), sbk, sbn,
                                C.data_ptr(), N, M, N, K, ctypes.c_void_p(prof.data_ptr()), ws.data_ptr(), ws.numel(), mode)
        assert rc == 0
    torch.cuda.synchronize()
    p = prof.cpu().numpy()[:32].reshape(8, 4)
    steps = K // 32 if mode == 2 or M > 1024 else None
    print(name, "(cycles over the whole K loop of a workgroup, per wave: frags | first half | second half | barrier)")
    for w in range(8):
        print("   wave %d" % w, p[w], "total", p[w].sum())
