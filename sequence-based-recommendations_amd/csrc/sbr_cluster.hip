// The cluster head of RNNCluster (`train.py -m RNN --clusters C`, /root/reference/neural_networks/rnn_cluster.py:237-256, :275-300,
// :327-352): a second, small model that reads the user representation h (the recurrent stack's final state) and trains ONLY its own
// two arrays -- the cluster-selection weights Wc (H, C) (a DenseLayer without bias, :239) and the item / cluster repartition
// R (N, C) (:244) -- with its own call of the update manager (:282-285); nothing of it reaches the recurrent network, whose
// sampled head and loss stay in the engine (sbr_api.hip, losses SBR_LOSS_BLACKOUT .. SBR_LOSS_LIN).
//
//   z = h . Wc (+ noise);  p = softmax(scale z)                                          cluster selection        :240-243
//   M = f(R[targets ++ cluster_samples]),  f = softmax(scale .) | + sigmoid(scale .) | sigmoid(scale .)          :245-253
//   score = p . M^T;  cost_clusters = mean over rows of the sampled loss of `score` (row b's positive = column b) :254-256
//   hard clusters = f(100 R) (mix: clipped to [0, 1]);  a row's cluster = argmax z                                 :293-300, :334
//
// Sizes are tiny (B x (B + S') x C with C ~ 10 - 100): one workgroup per row / cell, f32 throughout, no MFMA.  The arrays are
// the object's own device allocations; h and the score matrix of the test function are the engine's buffers (device pointers).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <new>
#include "sbr_common.h"

struct sbr_cluster {
    sbr_cluster_config cfg;
    hipStream_t stream;
    float *R, *Wc, *dR, *dWc, *sR[2], *sWc[2];       // parameters, gradients, optimizer state
    float *z, *p, *M, *sm, *sg, *score, *dz, *rowcost, *ones, *cost, *hard, *nused;
    int* ids;
    long step;
    float scale;
    unsigned long long noise_ctr;
    int J;                                           // cells of the last forward (B + samples)
    int hard_valid;
    int dR_clean;                                    // dR is all zeros: the optimizer kernel clears every gradient it consumes (no N x C memset per step)
};

extern void sbr_set_error(const char* fmt, ...);
#define CL_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { sbr_set_error("%s: %s", #x, hipGetErrorString(e_)); return SBR_EHIP; } } while (0)
#define CL_ARG(c, ...) do { if (!(c)) { sbr_set_error(__VA_ARGS__); return SBR_EINVAL; } } while (0)

namespace {

// feature k of row b of the user representation: [0, H1) at column k, the rest at off2 + (k - H1) (--r_bi: two padded halves)
__device__ __forceinline__ float h_at(const float* __restrict__ h, int ld, int H1, int off2, int b, int k) {
    return h[(size_t)b * ld + (k < H1 ? k : off2 + (k - H1))];
}
__device__ __forceinline__ float sigm1(float x) { return 1.0f / (1.0f + expf(-x)); }

// counter-based normal draws (cluster_selection_noise, rnn_cluster.py:241-242: the reference draws from MRG_RandomStreams, so the
// two agree in law, not in stream)
__device__ __forceinline__ unsigned mix32(unsigned long long x) {
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (unsigned)x;
}
__device__ __forceinline__ float normal_draw(unsigned long long seed, unsigned long long idx) {
    const float u1 = ((float)mix32(seed + 2 * idx) + 1.0f) * (1.0f / 4294967808.0f);
    const float u2 = (float)mix32(seed + 2 * idx + 1) * (1.0f / 4294967296.0f);
    return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// one workgroup per row: z = h . Wc (+ noise), p = softmax(scale z)
__global__ void __launch_bounds__(256) cl_select_kernel(const float* __restrict__ h, int ld, int H1, int off2, int H,
                                                        const float* __restrict__ Wc, int C, float scale, float noise_std,
                                                        unsigned long long seed, float* __restrict__ z, float* __restrict__ p,
                                                        int* __restrict__ csel) {
    extern __shared__ float sh[];                    // [C] activations, then [256] reduction scratch
    float* red = sh + C;
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.0f;
        for (int k = 0; k < H; ++k) s = fmaf(h_at(h, ld, H1, off2, b, k), Wc[(size_t)k * C + c], s);
        if (noise_std > 0.0f) s += noise_std * normal_draw(seed, (unsigned long long)b * C + c);
        sh[c] = s;
        if (z) z[(size_t)b * C + c] = s;
    }
    __syncthreads();
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, scale * sh[c]);
    red[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float se = 0.0f;
    for (int c = threadIdx.x; c < C; c += 256) se += expf(scale * sh[c] - mx);
    red[threadIdx.x] = se;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    se = red[0];
    if (p) for (int c = threadIdx.x; c < C; c += 256) p[(size_t)b * C + c] = expf(scale * sh[c] - mx) / se;
    if (csel && threadIdx.x == 0) {                  // argmax, ties -> the lowest cluster (numpy argmax, rnn_cluster.py:334)
        int best = 0;
        for (int c = 1; c < C; ++c) if (sh[c] > sh[best]) best = c;
        csel[b] = best;
    }
}

// membership of one row r of R at scale `scale`: out = f(r); sm / sg (nullable) = its softmax / sigmoid parts
__device__ __forceinline__ void membership_row(const float* __restrict__ r, int C, float scale, int type, float* __restrict__ out,
                                               float* __restrict__ sm, float* __restrict__ sg, float* red, bool clip01) {
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 64) mx = fmaxf(mx, scale * r[c]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float se = 0.0f;
    for (int c = threadIdx.x; c < C; c += 64) se += expf(scale * r[c] - mx);
    for (int o = 32; o > 0; o >>= 1) se += __shfl_xor(se, o);
    for (int c = threadIdx.x; c < C; c += 64) {
        const float s = expf(scale * r[c] - mx) / se, g = sigm1(scale * r[c]);
        float v = type == SBR_CLUSTER_SOFTMAX ? s : type == SBR_CLUSTER_MIX ? s + g : g;
        if (clip01) v = fminf(fmaxf(v, 0.0f), 1.0f);
        out[c] = v;
        if (sm) sm[c] = s;
        if (sg) sg[c] = g;
    }
}
// one wave per cell j: M[j] = f(R[ids[j]])   /   one wave per item n: hard[n] = f(100 R[n]), clipped (mix)
__global__ void __launch_bounds__(64) cl_members_kernel(const float* __restrict__ R, const int* __restrict__ ids, int C, float scale, int type,
                                                        float* __restrict__ M, float* __restrict__ sm, float* __restrict__ sg, int clip01) {
    const size_t j = blockIdx.x;
    const size_t row = ids ? (size_t)ids[j] : j;
    membership_row(R + row * C, C, scale, type, M + j * C, sm ? sm + j * C : nullptr, sg ? sg + j * C : nullptr, nullptr, clip01 != 0);
}

// score[b][j] = sum_c p[b][c] M[j][c]
__global__ void __launch_bounds__(256) cl_score_kernel(const float* __restrict__ p, const float* __restrict__ M, int C, int J, float* __restrict__ score) {
    const int b = blockIdx.x;
    for (int j = threadIdx.x; j < J; j += 256) {
        float s = 0.0f;
        for (int c = 0; c < C; ++c) s = fmaf(p[(size_t)b * C + c], M[(size_t)j * C + c], s);
        score[(size_t)b * J + j] = s;
    }
}

// one workgroup per row b: dp = dscore[b] . M, dz = scale p (dp - <p, dp>)
__global__ void __launch_bounds__(256) cl_back_p_kernel(const float* __restrict__ ds, const float* __restrict__ M, const float* __restrict__ p,
                                                        int C, int J, float scale, float* __restrict__ dz) {
    extern __shared__ float sh[];                    // [C] dp, [256] scratch
    float* red = sh + C;
    const int b = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.0f;
        for (int j = 0; j < J; ++j) s = fmaf(ds[(size_t)b * J + j], M[(size_t)j * C + c], s);
        sh[c] = s;
    }
    __syncthreads();
    float dot = 0.0f;
    for (int c = threadIdx.x; c < C; c += 256) dot += sh[c] * p[(size_t)b * C + c];
    red[threadIdx.x] = dot;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    dot = red[0];
    for (int c = threadIdx.x; c < C; c += 256) dz[(size_t)b * C + c] = scale * p[(size_t)b * C + c] * (sh[c] - dot);
}

// dWc[k][c] = sum_b h[b][k] dz[b][c]   (one workgroup per k)
__global__ void __launch_bounds__(256) cl_back_wc_kernel(const float* __restrict__ h, int ld, int H1, int off2, const float* __restrict__ dz,
                                                         int B, int C, float* __restrict__ dWc) {
    const int k = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.0f;
        for (int b = 0; b < B; ++b) s = fmaf(h_at(h, ld, H1, off2, b, k), dz[(size_t)b * C + c], s);
        dWc[(size_t)k * C + c] = s;
    }
}

// one wave per cell j: dM = dscore[:, j]^T . p; back through f; dR[ids[j]] += dr (duplicate cells accumulate: AdvancedIncSubtensor [3P])
__global__ void __launch_bounds__(64) cl_back_m_kernel(const float* __restrict__ ds, const float* __restrict__ p, const float* __restrict__ sm,
                                                       const float* __restrict__ sg, const int* __restrict__ ids, int B, int C, int J,
                                                       float scale, int type, float* __restrict__ dR) {
    const int j = blockIdx.x;
    float dots = 0.0f;                               // <sm, dM> for the softmax part
    // two passes over c (C is small): first the dot product, then the row
    for (int pass = 0; pass < 2; ++pass) {
        for (int c = threadIdx.x; c < C; c += 64) {
            float dm = 0.0f;
            for (int b = 0; b < B; ++b) dm = fmaf(ds[(size_t)b * J + j], p[(size_t)b * C + c], dm);
            if (pass == 0) { if (type != SBR_CLUSTER_SIGMOID) dots += sm[(size_t)j * C + c] * dm; }
            else {
                float dr = 0.0f;
                if (type != SBR_CLUSTER_SIGMOID) { const float s = sm[(size_t)j * C + c]; dr += scale * s * (dm - dots); }
                if (type != SBR_CLUSTER_SOFTMAX) { const float g = sg[(size_t)j * C + c]; dr += scale * g * (1.0f - g) * dm; }
                atomicAdd(&dR[(size_t)ids[j] * C + c], dr);
            }
        }
        if (pass == 0) for (int o = 32; o > 0; o >>= 1) dots += __shfl_xor(dots, o);
    }
}

__global__ void cl_sum_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ float red[256];
    float s = 0.0f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) *out = red[0];
}
__global__ void cl_fill_kernel(float* x, int n, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}
__global__ void cl_concat_ids_kernel(const int* __restrict__ a, int na, const int* __restrict__ b, int nb, int* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < na) out[i] = a[i]; else if (i < na + nb) out[i] = b[i - na];
}
// scores[b][n] *= hard[n][csel[b]]   (rnn_cluster.py:335-336)
__global__ void cl_mask_kernel(float* __restrict__ scores, int ld, int N, int C, const float* __restrict__ hard, const int* __restrict__ csel) {
    const int b = blockIdx.y;
    const int c = csel[b];
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) scores[(size_t)b * ld + n] *= hard[(size_t)n * C + c];
}
// nused[b] = sum_n hard[n][csel[b]]
__global__ void __launch_bounds__(256) cl_nused_kernel(const float* __restrict__ hard, int N, int C, const int* __restrict__ csel, float* __restrict__ nused) {
    __shared__ float red[256];
    const int c = csel[blockIdx.x];
    float s = 0.0f;
    for (int n = threadIdx.x; n < N; n += 256) s += hard[(size_t)n * C + c];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) nused[blockIdx.x] = red[0];
}

}  // namespace

extern "C" sbr_cluster* sbr_cluster_create(const sbr_cluster_config* cfg, void* stream) {
    if (!cfg || cfg->abi_version != SBR_ABI_VERSION) { sbr_set_error("sbr_cluster_create: abi_version mismatch"); return nullptr; }
    const sbr_cluster_config& c = *cfg;
    if (c.n_items < 1 || c.n_hidden < 1 || c.n_clusters < 1 || c.batch_size < 1 || c.max_samples < 1 || c.hidden_split < 0 ||
        c.hidden_split > c.n_hidden || c.cluster_type < 0 || c.cluster_type > 2) { sbr_set_error("sbr_cluster_create: bad sizes"); return nullptr; }
    const bool sampled = c.loss == SBR_LOSS_BLACKOUT || c.loss == SBR_LOSS_BPR || c.loss == SBR_LOSS_TOP1 || c.loss >= SBR_LOSS_SCCE;
    if (!sampled || c.loss > SBR_LOSS_LIN) { sbr_set_error("Unknown cluster loss"); return nullptr; }      // rnn_cluster.py:101
    if (c.updater < 0 || c.updater > SBR_UPD_ADAM) { sbr_set_error("Unknown update option"); return nullptr; }
    sbr_cluster* k = new (std::nothrow) sbr_cluster();
    if (!k) return nullptr;
    k->cfg = c; k->stream = (hipStream_t)stream; k->step = 0; k->scale = c.scale; k->noise_ctr = 0; k->J = 0; k->hard_valid = 0; k->dR_clean = 0;
    const size_t N = c.n_items, C = c.n_clusters, H = c.n_hidden, B = c.batch_size, J = B + c.max_samples;
    const size_t nR = N * C, nW = H * C;
    float** f[] = {&k->R, &k->Wc, &k->dR, &k->dWc, &k->sR[0], &k->sR[1], &k->sWc[0], &k->sWc[1], &k->z, &k->p, &k->M, &k->sm, &k->sg,
                   &k->score, &k->dz, &k->rowcost, &k->ones, &k->cost, &k->hard, &k->nused};
    const size_t n[] = {nR, nW, nR, nW, nR, nR, nW, nW, B * C, B * C, J * C, J * C, J * C, B * J, B * C, B, B, 4, nR, B};
    for (size_t i = 0; i < sizeof(n) / sizeof(n[0]); ++i) {
        *f[i] = nullptr;
        if (hipMalloc((void**)f[i], n[i] * sizeof(float)) != hipSuccess || hipMemsetAsync(*f[i], 0, n[i] * sizeof(float), k->stream) != hipSuccess) {
            sbr_set_error("sbr_cluster_create: out of device memory"); sbr_cluster_destroy(k); return nullptr;
        }
    }
    k->ids = nullptr;
    if (hipMalloc((void**)&k->ids, J * sizeof(int)) != hipSuccess) { sbr_set_error("sbr_cluster_create: out of device memory"); sbr_cluster_destroy(k); return nullptr; }
    cl_fill_kernel<<<(unsigned)((B + 255) / 256), 256, 0, k->stream>>>(k->ones, (int)B, 1.0f);
    return k;
}

extern "C" void sbr_cluster_destroy(sbr_cluster* k) {
    if (!k) return;
    float* f[] = {k->R, k->Wc, k->dR, k->dWc, k->sR[0], k->sR[1], k->sWc[0], k->sWc[1], k->z, k->p, k->M, k->sm, k->sg, k->score, k->dz,
                  k->rowcost, k->ones, k->cost, k->hard, k->nused};
    (void)hipStreamSynchronize(k->stream);
    for (float* q : f) if (q) (void)hipFree(q);
    if (k->ids) (void)hipFree(k->ids);
    delete k;
}

extern "C" int sbr_cluster_set_params(sbr_cluster* k, const float* R, const float* Wc) {
    CL_ARG(k && R && Wc, "null argument");
    CL_HIP(hipMemcpyAsync(k->R, R, (size_t)k->cfg.n_items * k->cfg.n_clusters * sizeof(float), hipMemcpyHostToDevice, k->stream));
    CL_HIP(hipMemcpyAsync(k->Wc, Wc, (size_t)k->cfg.n_hidden * k->cfg.n_clusters * sizeof(float), hipMemcpyHostToDevice, k->stream));
    CL_HIP(hipStreamSynchronize(k->stream));
    k->hard_valid = 0;
    return SBR_OK;
}
static int cl_get(sbr_cluster* k, float* R, float* Wc, const float* dR, const float* dW) {
    if (R) CL_HIP(hipMemcpyAsync(R, dR, (size_t)k->cfg.n_items * k->cfg.n_clusters * sizeof(float), hipMemcpyDeviceToHost, k->stream));
    if (Wc) CL_HIP(hipMemcpyAsync(Wc, dW, (size_t)k->cfg.n_hidden * k->cfg.n_clusters * sizeof(float), hipMemcpyDeviceToHost, k->stream));
    CL_HIP(hipStreamSynchronize(k->stream));
    return SBR_OK;
}
extern "C" int sbr_cluster_get_params(sbr_cluster* k, float* R, float* Wc) { CL_ARG(k, "null handle"); return cl_get(k, R, Wc, k->R, k->Wc); }
extern "C" int sbr_cluster_get_grads(sbr_cluster* k, float* dR, float* dWc) { CL_ARG(k, "null handle"); return cl_get(k, dR, dWc, k->dR, k->dWc); }
extern "C" int sbr_cluster_set_scale(sbr_cluster* k, float scale) { CL_ARG(k && scale > 0.0f, "bad scale"); k->scale = scale; return SBR_OK; }

extern "C" int sbr_cluster_forward_backward(sbr_cluster* k, const float* h_dev, int ld_h, int off2, const int32_t* targets_dev,
                                            const int32_t* samples_dev, int n_samples, float* cost_host) {
    CL_ARG(k && h_dev && targets_dev && samples_dev, "null argument");
    const sbr_cluster_config& c = k->cfg;
    CL_ARG(n_samples >= 1 && n_samples <= c.max_samples, "%d cluster samples outside [1, %d]", n_samples, c.max_samples);
    const int B = c.batch_size, C = c.n_clusters, H = c.n_hidden, J = B + n_samples;
    hipStream_t s = k->stream;
    k->J = J;
    cl_concat_ids_kernel<<<(J + 255) / 256, 256, 0, s>>>(targets_dev, B, samples_dev, n_samples, k->ids);
    const size_t lds = (size_t)(C + 256) * sizeof(float);
    cl_select_kernel<<<B, 256, lds, s>>>(h_dev, ld_h, c.hidden_split, off2, H, k->Wc, C, k->scale, c.noise_std,
                                         0x9E3779B97F4A7C15ull * (++k->noise_ctr) + c.seed, k->z, k->p, nullptr);
    cl_members_kernel<<<J, 64, 0, s>>>(k->R, k->ids, C, k->scale, c.cluster_type, k->M, k->sm, k->sg, 0);
    cl_score_kernel<<<B, 256, 0, s>>>(k->p, k->M, C, J, k->score);
    CL_HIP(launch_sampled_loss(s, k->score, nullptr, k->ones, k->rowcost, B, B, n_samples, 0, c.loss, B));      // score -> d cost / d score
    cl_sum_kernel<<<1, 256, 0, s>>>(k->rowcost, B, k->cost);
    if (!k->dR_clean) CL_HIP(hipMemsetAsync(k->dR, 0, (size_t)c.n_items * C * sizeof(float), s));      // (only where no update ran in between)
    k->dR_clean = 0;
    cl_back_p_kernel<<<B, 256, lds, s>>>(k->score, k->M, k->p, C, J, k->scale, k->dz);
    cl_back_wc_kernel<<<H, 256, 0, s>>>(h_dev, ld_h, c.hidden_split, off2, k->dz, B, C, k->dWc);
    cl_back_m_kernel<<<J, 64, 0, s>>>(k->score, k->p, k->sm, k->sg, k->ids, B, C, J, k->scale, c.cluster_type, k->dR);
    CL_HIP(hipGetLastError());
    if (cost_host) {
        CL_HIP(hipMemcpyAsync(cost_host, k->cost, sizeof(float), hipMemcpyDeviceToHost, s));
        CL_HIP(hipStreamSynchronize(s));
    }
    return SBR_OK;
}

extern "C" int sbr_cluster_apply_update(sbr_cluster* k) {      // self.updater(cost_clusters, [Wc, R]): dense, its own step count
    CL_ARG(k, "null handle");
    const sbr_cluster_config& c = k->cfg;
    k->step += 1;
    const bool two = c.updater == SBR_UPD_ADAM || c.updater == SBR_UPD_ADADELTA;
    CL_HIP(launch_update(k->stream, c.updater, k->Wc, k->dWc, k->sWc[0], two ? k->sWc[1] : nullptr, (size_t)c.n_hidden * c.n_clusters,
                         c.learning_rate, c.rho, c.beta1, c.beta2, k->step));
    CL_HIP(launch_update(k->stream, c.updater, k->R, k->dR, k->sR[0], two ? k->sR[1] : nullptr, (size_t)c.n_items * c.n_clusters,
                         c.learning_rate, c.rho, c.beta1, c.beta2, k->step));
    k->hard_valid = 0;
    k->dR_clean = 1;                                 // update_kernel has cleared dR (and dWc)
    return SBR_OK;
}

extern "C" int sbr_cluster_select(sbr_cluster* k, const float* h_dev, int ld_h, int off2, int rows, int32_t* csel_dev, float* z_dev) {
    CL_ARG(k && h_dev && csel_dev && rows >= 1, "bad argument");
    const sbr_cluster_config& c = k->cfg;
    const size_t lds = (size_t)(c.n_clusters + 256) * sizeof(float);
    cl_select_kernel<<<rows, 256, lds, k->stream>>>(h_dev, ld_h, c.hidden_split, off2, c.n_hidden, k->Wc, c.n_clusters, 1.0f, 0.0f, 0, z_dev,
                                                    nullptr, csel_dev);      // deterministic=True: no noise
    CL_HIP(hipGetLastError());
    return SBR_OK;
}

extern "C" int sbr_cluster_mask_scores(sbr_cluster* k, float* scores_dev, int ld, int rows, const int32_t* csel_dev, float* n_used_dev) {
    CL_ARG(k && csel_dev && rows >= 1 && (scores_dev || n_used_dev), "bad argument");
    const sbr_cluster_config& c = k->cfg;
    if (!k->hard_valid) {        // _get_hard_clusters: f(100 R), the mix clipped to [0, 1]
        cl_members_kernel<<<c.n_items, 64, 0, k->stream>>>(k->R, nullptr, c.n_clusters, 100.0f, c.cluster_type, k->hard, nullptr, nullptr, 1);
        k->hard_valid = 1;
    }
    if (scores_dev) {
        CL_ARG(ld >= c.n_items, "row stride %d < %d items", ld, c.n_items);
        const dim3 grid((unsigned)std::min(256, (c.n_items + 255) / 256), (unsigned)rows);
        cl_mask_kernel<<<grid, 256, 0, k->stream>>>(scores_dev, ld, c.n_items, c.n_clusters, k->hard, csel_dev);
    }
    if (n_used_dev) cl_nused_kernel<<<rows, 256, 0, k->stream>>>(k->hard, c.n_items, c.n_clusters, csel_dev, n_used_dev);
    CL_HIP(hipGetLastError());
    return SBR_OK;
}

extern "C" int sbr_cluster_hard(sbr_cluster* k, float* hard_host) {      // [N][C]: what prepare_tests / the metrics read (host copy)
    CL_ARG(k && hard_host, "null argument");
    const sbr_cluster_config& c = k->cfg;
    cl_members_kernel<<<c.n_items, 64, 0, k->stream>>>(k->R, nullptr, c.n_clusters, 100.0f, c.cluster_type, k->hard, nullptr, nullptr, 1);
    k->hard_valid = 1;
    CL_HIP(hipMemcpyAsync(hard_host, k->hard, (size_t)c.n_items * c.n_clusters * sizeof(float), hipMemcpyDeviceToHost, k->stream));
    CL_HIP(hipStreamSynchronize(k->stream));
    return SBR_OK;
}
