"""The dense GEMM kernels of the hot path on their own (sbr_debug_gemm): exact-f32 MFMA kernel and the bf16x6 kernel
against float64, on the operand layouts and edge shapes the engine uses (output projection NT/NN/TN forms with
N = 3706 / 26744 item columns, weight gradients with K = T*B positions, unaligned leading dimensions, split-K)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def run(M, N, K, a_t, b_t, exact, bias=False, ws_floats=0, seed=0, lda_pad=0, ldb_pad=0):
    import torch
    from sbr_amd.engine import load_library
    lib = load_library()
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda")
    # A(m,k): stored [M][K+pad] (k fast) or, transposed, [K][M+pad] (m fast); same for B(k,n)
    if a_t:
        Ah = rng.standard_normal((K, M + lda_pad)).astype(np.float32); A64 = Ah[:, :M].T.astype(np.float64)
        sam, sak = 1, M + lda_pad
    else:
        Ah = rng.standard_normal((M, K + lda_pad)).astype(np.float32); A64 = Ah[:, :K].astype(np.float64)
        sam, sak = K + lda_pad, 1
    if b_t:
        Bh = rng.standard_normal((N, K + ldb_pad)).astype(np.float32); B64 = Bh[:, :K].T.astype(np.float64)
        sbk, sbn = 1, K + ldb_pad
    else:
        Bh = rng.standard_normal((K, N + ldb_pad)).astype(np.float32); B64 = Bh[:, :N].astype(np.float64)
        sbk, sbn = N + ldb_pad, 1
    A, B = torch.from_numpy(Ah).to(dev), torch.from_numpy(Bh).to(dev)
    C = torch.full((M, N), float("nan"), device=dev)
    bv = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).to(dev) if bias else None
    ws = torch.empty(ws_floats, device=dev) if ws_floats else None
    rc = lib.sbr_debug_gemm(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), A.data_ptr(), sam, sak, B.data_ptr(), sbk, sbn,
                            C.data_ptr(), N, M, N, K, bv.data_ptr() if bias else None, ws.data_ptr() if ws_floats else None,
                            ws_floats, int(exact))
    assert rc == 0, lib.sbr_last_error()
    torch.cuda.synchronize()
    ref = A64 @ B64 + (bv.cpu().numpy().astype(np.float64)[None, :] if bias else 0.0)
    scale = np.sqrt(K)                                      # entries ~ N(0, K)
    return float(np.abs(C.cpu().numpy() - ref).max() / scale)


SHAPES = [
    # (M, N, K, a_t, b_t, ws)   engine use
    (256, 3706, 128, False, True, 0),            # logits = h . W_out^T            (NT, N not a multiple of 4 or 128)
    (256, 128, 3706, False, False, 1 << 20),     # dh = dlogits . W_out            (NN, split-K, rows of 3706 floats: 8-byte aligned)
    (3706, 128, 256, True, False, 1 << 22),      # dW_out^T = dlogits^T . h        (TN)
    (128, 384, 51200, True, False, 1 << 22),     # dW_hid = hs^T . dhi             (TN, K = T*B positions)
    (200, 130, 97, False, False, 0),             # ragged everything, odd leading dimensions (4-byte loads)
    (131, 257, 64, True, True, 0),
    (96, 96, 32, False, True, 0),                # smallest shape the 128x128 bf16x6 tile takes
    (48, 48, 32, False, True, 0),                # smallest shape of the 64x64 tile
    (64, 200, 160, True, False, 0),              # 64x64 tiles, k tail of 32 inside a 64-wide step
    (256, 128, 3712, False, False, 1 << 20),     # dh with 16-byte aligned rows: 64x64 tiles + split-K
    (80, 3706, 128, False, True, 0),             # logits of a small batch
    (512, 1024, 256, False, False, 0),           # layer >= 2 input projection
]


@pytest.mark.parametrize("exact", [False, True])
@pytest.mark.parametrize("shape", SHAPES)
def test_gemm_kernels_against_float64(shape, exact):
    M, N, K, a_t, b_t, ws = shape
    err = run(M, N, K, a_t, b_t, exact, bias=(ws == 0), ws_floats=ws)
    assert err < 5e-6, (shape, exact, err)                  # f32-rounding class for both kernels


def test_gemm_unaligned_leading_dimensions():
    for pad in (1, 2, 3):
        assert run(160, 200, 128, False, False, False, lda_pad=pad, ldb_pad=pad) < 3e-6
        assert run(160, 200, 128, True, True, False, lda_pad=pad, ldb_pad=pad) < 3e-6


@pytest.mark.parametrize("shape", [(256, 3706, 128, False, True), (1, 5000, 512, False, True), (37, 100000, 256, False, True),
                                   (256, 26744, 256, False, True), (300, 132, 96, True, False)])
def test_plain_bf16_projection_kernel(shape):
    # SBR_FLAG_BF16_PROJECTION's kernel (mode 2): operands rounded to bf16 (2^-9 each), f32 accumulation -- the error of a
    # K-term dot product of N(0,1) entries is ~2^-8.5 * sqrt(K) * (a small factor for the worst of M*N entries)
    M, N, K, a_t, b_t = shape
    err = run(M, N, K, a_t, b_t, 2, bias=True)
    assert 1e-5 < err < 2.5e-2, (shape, err)          # bf16-class, not f32-class: the flag really selects the one-plane kernel


def test_plain_bf16_projection_rows_do_not_depend_on_the_batch():
    # a row's scores are the same bits whether 1 or 256 rows share the call (ranked ids must not depend on the batch size)
    import torch
    from sbr_amd.engine import load_library
    lib = load_library()
    rng = np.random.default_rng(3)
    K, N = 256, 7000
    A = torch.from_numpy(rng.standard_normal((256, K)).astype(np.float32)).cuda()
    B = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).cuda()
    outs = []
    for M in (256, 100, 1):
        C = torch.empty((M, N), device="cuda")
        rc = lib.sbr_debug_gemm(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), A.data_ptr(), K, 1, B.data_ptr(), 1, K,
                                C.data_ptr(), N, M, N, K, None, None, 0, 2)
        assert rc == 0, lib.sbr_last_error()
        outs.append(C.cpu().numpy())
    assert np.array_equal(outs[0][:100], outs[1]) and np.array_equal(outs[0][:1], outs[2])


# (M, N, K, b_t, ws): shapes the 256 x 128 tile of gemm_x6w_kernel takes (A k-contiguous, M % 256 == N % 128 == K % 32 == 0, >= 128 workgroups)
WIDE_SHAPES = [
    (8192, 2048, 512, False, 0),        # layer-2 input projection (NN), scaled down from 51 200 rows
    (8192, 512, 2048, True, 0),         # its backward, dx = dxt . W^T (NT)
    (4096, 1024, 256, False, 0),        # C3-sized projection
    (2048, 2048, 1024, True, 1 << 24),  # split-K slabs through the wide tile (K >= 512 and a workspace: launch_gemm's plan)
    (4096, 1024, 96, False, 0),         # three k steps: prologue, one full step, the two peeled ones
    (4096, 1024, 32, True, 0),          # a single k step
]


@pytest.mark.parametrize("planes", ["f16x3", "bf16"])
@pytest.mark.parametrize("shape", WIDE_SHAPES)
def test_wide_tile_equals_the_128_wide_kernel_bit_for_bit(shape, planes):
    """gemm_x6w_kernel against gemm_x6_kernel on the same operands (sbr_debug_gemm modes 3 / 4 and 2 / 5): the same products in the same
    order over K, so not a bit differs -- and both inside the f32-rounding class against float64 (fp16 split) or the bf16 bar."""
    import torch
    from sbr_amd.engine import load_library
    lib = load_library()
    M, N, K, b_t, wsf = shape
    if planes == "bf16" and wsf:
        pytest.skip("plain bf16 operands never take split-K")
    dev = torch.device("cuda")
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    A = torch.rand((M, K), device=dev, generator=g) * 2 - 1
    B = (torch.rand((N, K) if b_t else (K, N), device=dev, generator=g) * 2 - 1) * 0.25
    bias = torch.rand(N, device=dev, generator=g)
    ws = torch.empty(wsf, device=dev) if wsf else None
    sbk, sbn = (1, K) if b_t else (N, 1)
    out = {}
    for mode in ((3, 4) if planes == "f16x3" else (2, 5)):
        C = torch.full((M, N), float("nan"), device=dev)
        rc = lib.sbr_debug_gemm(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), A.data_ptr(), K, 1, B.data_ptr(), sbk, sbn,
                                C.data_ptr(), N, M, N, K, bias.data_ptr(), ws.data_ptr() if wsf else None, wsf, mode)
        assert rc == 0, lib.sbr_last_error()
        torch.cuda.synchronize()
        out[mode] = C
    wide, narrow = out.values()
    assert torch.equal(wide, narrow)
    rows = torch.arange(0, M, M // 32, device=dev)
    ref = A[rows].double() @ (B.t() if b_t else B).double() + bias.double()
    err = ((wide[rows].double() - ref).abs().max() / ref.abs().max()).item()
    assert err < (2e-6 if planes == "f16x3" else 1e-2), err
