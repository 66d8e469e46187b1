// Device-side cell math and bf16x6 helpers shared by the recurrent kernels (sbr_rec.hip: one workgroup per
// row tile; sbr_rec_cl.hip: a cluster of workgroups per row tile).
// Math follows the reference's scan step functions (neural_networks/sparse_lstm.py:377-425 LSTM,
// :764-805 GRU, :1120-1152 Vanilla); the backward is the hand-derived BPTT pinned by oracle/rnn_oracle.py.
#pragma once
#include "sbr_common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CELL_LSTM 0
#define CELL_GRU 1
#define CELL_VANILLA 2

__device__ __forceinline__ float sigm(float x) { return 1.0f / (1.0f + expf(-x)); }
// v_exp_f32 / v_rcp_f32 forms used by the MFMA kernels (1 ulp-class hardware transcendentals; the
// libm forms above cost ~25 VALU instructions each and sat on the per-step critical path)
__device__ __forceinline__ float sigm_fast(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanh_fast(float x) {   // 2*sigmoid(2x) - 1, abs error ~1 ulp(1.0)
    return fmaf(2.0f, __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-2.8853900817779268f * x)), -1.0f);
}
template <bool FAST> __device__ __forceinline__ float sg(float x) { return FAST ? sigm_fast(x) : sigm(x); }
template <bool FAST> __device__ __forceinline__ float th(float x) { return FAST ? tanh_fast(x) : tanhf(x); }
__device__ __forceinline__ float clipf(float x, float c) { return c > 0.0f ? fminf(fmaxf(x, -c), c) : x; }
// one v_med3_f32 per clip instead of max + min + select: ce = clip_bound(c), hoisted out of the step loops
__device__ __forceinline__ float clip_bound(float c) { return c > 0.0f ? c : 3.402823466e38f; }
template <bool FAST> __device__ __forceinline__ float clipb(float x, float c, float ce) {
    return FAST ? __builtin_amdgcn_fmed3f(x, -ce, ce) : clipf(x, c);
}
template <int CELL> struct Gates { static constexpr int G = CELL == CELL_LSTM ? 4 : (CELL == CELL_GRU ? 3 : 1); };

// ---------------------------------------------------------------------------------------
// Scalar cell math shared by the MFMA kernels and the triage ("simple") kernels
// ---------------------------------------------------------------------------------------
// Forward: a[g] = (h_prev . W_hid)[g], x[g] = xt[g].  Updates h/c in place (masked rows copy),
// writes the values saved for BPTT into sv[0..3].
// relu (Vanilla only): the layer is a stock lasagne RecurrentLayer (dense input: every Vanilla layer above layer 0 and
// layer 0 behind --r_emb, recurrent_layers.py:94-104), whose default nonlinearity is rectify [3P]; the reference's own
// index-input VanillaLayerOHEInput uses tanh (sparse_lstm.py:1015).
template <int CELL, bool FAST = false>
__device__ __forceinline__ void cell_forward(const float* x, const float* a, bool m, float& h, float& c,
                                             float pi, float pf, float po, float* sv, bool relu = false) {
    if (CELL == CELL_LSTM) {
        float i = sg<FAST>(x[0] + a[0] + c * pi);             // sparse_lstm.py:397-402
        float f = sg<FAST>(x[1] + a[1] + c * pf);
        float g = th<FAST>(x[2] + a[2]);
        float cn = f * c + i * g;                             // :407
        float o = sg<FAST>(x[3] + a[3] + cn * po);            // :409-411
        float hn = o * th<FAST>(cn);                          // :414
        sv[0] = i; sv[1] = f; sv[2] = g; sv[3] = o;
        c = m ? cn : c; h = m ? hn : h;                       // :422-423
    } else if (CELL == CELL_GRU) {
        float r = sg<FAST>(a[0] + x[0]);                      // :780-783
        float u = sg<FAST>(a[1] + x[1]);
        float cc = th<FAST>(x[2] + r * a[2]);                 // :786-792
        float hn = (1.0f - u) * h + u * cc;                   // :795
        sv[0] = r; sv[1] = u; sv[2] = cc; sv[3] = a[2];
        h = m ? hn : h;                                       // :803
    } else {
        const float pre = x[0] + a[0];
        float hn = relu ? fmaxf(pre, 0.0f) : th<FAST>(pre);   // :1133-1143
        h = m ? hn : h;                                       // :1150
    }
}

// Backward of one step for one (row, unit).  In: dh, dc = grads wrt h_t, c_t; saved values.
// Out: dxi[g], dhi[g] (grad wrt xt and wrt hid_input, both clipped), dh/dc updated to the part
// that flows to step t-1 WITHOUT the dhi.W^T term (added by the caller); peephole partials.
template <int CELL, bool FAST = false>
__device__ __forceinline__ void cell_backward(bool m, float clip, float& dh, float& dc, const float* sv, float hprev,
                                              float cprev, float cnew, float hnew, float pi, float pf, float po,
                                              float* dxi, float* dhi, float* dpeep, bool relu = false) {
    float dhn = m ? dh : 0.0f, dhp = m ? 0.0f : dh;
    const float ce = clip_bound(clip);
#define CLIP(X) clipb<FAST>(X, clip, ce)
    if (CELL == CELL_LSTM) {
        float i = sv[0], f = sv[1], g = sv[2], o = sv[3];
        float dcn = m ? dc : 0.0f, dcp = m ? 0.0f : dc;
        float tc = th<FAST>(cnew);
        float dzo = dhn * tc * o * (1.0f - o);
        dcn += dhn * o * (1.0f - tc * tc) + dzo * po;
        float dzi = dcn * g * i * (1.0f - i);
        float dzf = dcn * cprev * f * (1.0f - f);
        float dac = dcn * i * (1.0f - g * g);
        dpeep[0] = dzi * cprev; dpeep[1] = dzf * cprev; dpeep[2] = dzo * cnew;
        dxi[0] = dhi[0] = CLIP(dzi); dxi[1] = dhi[1] = CLIP(dzf);
        dxi[2] = dhi[2] = CLIP(dac); dxi[3] = dhi[3] = CLIP(dzo);
        dc = dcp + dcn * f + dzi * pi + dzf * pf;
        dh = dhp;
    } else if (CELL == CELL_GRU) {
        float r = sv[0], u = sv[1], cc = sv[2], hic = sv[3];
        float du = dhn * (cc - hprev);
        float dq = CLIP(dhn * u * (1.0f - cc * cc));
        float dzr = dq * hic * r * (1.0f - r);
        float dzu = du * u * (1.0f - u);
        dxi[0] = CLIP(dzr); dxi[1] = CLIP(dzu); dxi[2] = CLIP(dq);
        dhi[0] = dxi[0]; dhi[1] = dxi[1]; dhi[2] = CLIP(dq * r);
        dh = dhp + dhn * (1.0f - u);
    } else {
        float dq = CLIP(dhn * (relu ? (hnew > 0.0f ? 1.0f : 0.0f) : (1.0f - hnew * hnew)));
        dxi[0] = dhi[0] = CLIP(dq);
        dh = dhp;
    }
}
#undef CLIP

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split3(float v, __bf16& b1, __bf16& b2, __bf16& b3) {
    b1 = (__bf16)v; float r = v - (float)b1; b2 = (__bf16)r; r -= (float)b2; b3 = (__bf16)r;
}
// Truncating split for values published every step: hi 16 bits of v, of the remainder, of its remainder (2 VALU per
// level instead of 3-4 for the round-to-nearest form).  Still exact (8 + 8 + 8 mantissa bits); the terms are up to
// twice as large as the rounded ones, so against round-to-nearest weight planes the products the bf16x6 scheme
// drops stay <= 2^-23 of the result.  Returns the three planes in the HIGH halves of b1, b2, b3.
__device__ __forceinline__ void split3_trunc(float v, unsigned& b1, unsigned& b2, unsigned& b3) {
    b1 = __float_as_uint(v) & 0xFFFF0000u;
    const float r1 = v - __uint_as_float(b1);
    b2 = __float_as_uint(r1) & 0xFFFF0000u;
    b3 = __float_as_uint(r1 - __uint_as_float(b2));
}
__device__ __forceinline__ void split3x4(const f32x4 v, bf16x4& p1, bf16x4& p2, bf16x4& p3) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { __bf16 a, b, c; split3(v[e], a, b, c); p1[e] = a; p2[e] = b; p3[e] = c; }
}
#define MFMA_BF16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x32_bf16(A, B, C, 0, 0, 0)

