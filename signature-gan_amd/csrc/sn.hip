// sn.hip -- spectral normalisation of the Discriminator's weights (torch.nn.utils.spectral_norm on every Conv2d and on the
// classifier, discriminator_vanilla_gan.py:60-62,200-202), gfx950.  Byte-moving / reduction work, HBM-bound, tiny.
//
// torch's hook, per forward pass of a layer with weight matrix W = weight_orig.view(Cout, -1):
//     training mode:  v <- normalize(W^T u);  u <- normalize(W v)      (one power iteration, in place, no grad)
//     every mode:     sigma = u . (W v);      weight = weight_orig / sigma
// and autograd differentiates weight w.r.t. weight_orig THROUGH sigma (u, v constants):
//     dL/dW_orig = (1/sigma) * (G - <G, W/sigma> u v^T),   G = dL/d(weight).
// Here every layer of one pass is handled by the same launches (job table, like k_prepare):
//   k_sn_wtu   t = W^T u                 (one thread per column, rows streamed coalesced)
//   k_sn_v     v = t / max(|t|, eps)     (one workgroup per layer)
//   k_sn_wv    w = W v                   (one wave per row)
//   k_sn_u     u = w / max(|w|, eps), sigma = u . w  (training)  |  sigma = u . w with the stored u (eval);
//              sigma, 1/sigma and copies of (u, v) are kept per PASS: the real and the fake pass of one D step see
//              different effective weights, and each pass's backward needs its own (sigma, u, v)
//   k_sn_dots / k_sn_combine   the gradient through sigma, summed over the passes of the step
#include "ops.h"

namespace siggan {

__device__ __forceinline__ float block_sum1024(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += sh[k];
    return s;
}

__global__ __launch_bounds__(256) void k_sn_wtu(const SnTable t) {
    int j = 0;
    while ((int)blockIdx.x >= t.pre_k[j + 1]) ++j;
    const SnLayer& L = t.layer[j];
    const int k = (blockIdx.x - t.pre_k[j]) * 256 + threadIdx.x;
    if (k >= L.K) return;
    const float* W = L.W + k;
    const float* u = t.u + L.u_off;
    float acc = 0.f;
#pragma unroll 8
    for (int i = 0; i < L.rows; ++i) acc = fmaf(W[(size_t)i * L.K], u[i], acc);
    t.tbuf[L.v_off + k] = acc;
}

__global__ __launch_bounds__(1024) void k_sn_v(const SnTable t, float eps) {
    __shared__ float sh[16];
    const SnLayer& L = t.layer[blockIdx.x];
    const float* tb = t.tbuf + L.v_off;
    float q = 0.f;
    for (int k = threadIdx.x; k < L.K; k += 1024) q = fmaf(tb[k], tb[k], q);
    const float nrm = sqrtf(block_sum1024(q, sh));
    const float inv = 1.0f / fmaxf(nrm, eps);
    float* v = t.v + L.v_off;
    for (int k = threadIdx.x; k < L.K; k += 1024) v[k] = tb[k] * inv;
}

__global__ __launch_bounds__(256) void k_sn_wv(const SnTable t) {
    int j = 0;
    while ((int)blockIdx.x >= t.pre_r[j + 1]) ++j;
    const SnLayer& L = t.layer[j];
    const int row = (blockIdx.x - t.pre_r[j]) * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= L.rows) return;
    const float* W = L.W + (size_t)row * L.K;
    const float* v = t.v + L.v_off;
    float acc = 0.f;
    for (int k = lane; k < L.K; k += 64) acc = fmaf(W[k], v[k], acc);
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if (lane == 0) t.wbuf[L.u_off + row] = acc;
}

// slot: where this pass's sigma / 1/sigma / (u, v) copies go
__global__ __launch_bounds__(1024) void k_sn_u(const SnTable t, float eps, int training, int slot) {
    __shared__ float sh[16];
    const SnLayer& L = t.layer[blockIdx.x];
    const float* w = t.wbuf + L.u_off;
    float* u = t.u + L.u_off;
    float sigma;
    if (training) {
        float q = 0.f;
        for (int i = threadIdx.x; i < L.rows; i += 1024) q = fmaf(w[i], w[i], q);
        const float nrm = sqrtf(block_sum1024(q, sh));
        const float inv = 1.0f / fmaxf(nrm, eps);
        float d = 0.f;
        for (int i = threadIdx.x; i < L.rows; i += 1024) { const float ui = w[i] * inv; u[i] = ui; d = fmaf(ui, w[i], d); }
        sigma = block_sum1024(d, sh);
    } else {
        float d = 0.f;
        for (int i = threadIdx.x; i < L.rows; i += 1024) d = fmaf(u[i], w[i], d);
        sigma = block_sum1024(d, sh);
    }
    if (threadIdx.x == 0) {
        t.sig[(slot * 2 + 0) * SnTable::MAXS + blockIdx.x] = sigma;
        t.sig[(slot * 2 + 1) * SnTable::MAXS + blockIdx.x] = 1.0f / sigma;
    }
    __syncthreads();
    float* us = t.u_saved + (size_t)slot * t.u_total + L.u_off;
    float* vs = t.v_saved + (size_t)slot * t.v_total + L.v_off;
    const float* v = t.v + L.v_off;
    for (int i = threadIdx.x; i < L.rows; i += 1024) us[i] = u[i];
    for (int k = threadIdx.x; k < L.K; k += 1024) vs[k] = v[k];
}

void launch_sn_sigma(const SnTable& t, int training, int slot, float eps, hipStream_t s) {
    if (training) {
        hipLaunchKernelGGL(k_sn_wtu, dim3(t.pre_k[t.n]), dim3(256), 0, s, t);
        hipLaunchKernelGGL(k_sn_v, dim3(t.n), dim3(1024), 0, s, t, eps);
    }
    hipLaunchKernelGGL(k_sn_wv, dim3(t.pre_r[t.n]), dim3(256), 0, s, t);
    hipLaunchKernelGGL(k_sn_u, dim3(t.n), dim3(1024), 0, s, t, eps, training, slot);
}

// ---- the gradient through sigma ------------------------------------------------------------------------------------
// G_p = dL/d(W / sigma_p) of pass p (temp arena p), P passes.  dots: d[p][layer] = <G_p, W> (block partials, 64 per layer and pass)
static constexpr int SN_CH = 64;
__global__ __launch_bounds__(256) void k_sn_dots(const SnTable t, const float* __restrict__ g0, const float* __restrict__ g1, int npass) {
    __shared__ float sh[4];
    const int layer = blockIdx.x / SN_CH, ch = blockIdx.x % SN_CH;
    const SnLayer& L = t.layer[layer];
    const int64_t n = (int64_t)L.rows * L.K;
    for (int p = 0; p < npass; ++p) {
        const float* g = (p == 0 ? g0 : g1) + L.w_off;
        float acc = 0.f;
        for (int64_t i = (int64_t)ch * 256 + threadIdx.x; i < n; i += (int64_t)SN_CH * 256) acc = fmaf(g[i], L.W[i], acc);
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) t.dots[((size_t)p * SnTable::MAXS + layer) * SN_CH + ch] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
    }
}
// out = sum_p [ G_p / sigma_p - (d_p / sigma_p^2) u_p v_p^T ] over the weight tensors; plain sum over everything else (biases)
__global__ __launch_bounds__(256) void k_sn_combine(const SnTable t, const float* __restrict__ g0, const float* __restrict__ g1,
                                                    float* __restrict__ out, int64_t total, int npass) {
    __shared__ float coef[2][SnTable::MAXS];        // d_p / sigma_p^2 per layer
    __shared__ float isg[2][SnTable::MAXS];
    if (threadIdx.x < 2 * SnTable::MAXS) {
        const int p = threadIdx.x / SnTable::MAXS, l = threadIdx.x % SnTable::MAXS;
        float c = 0.f, is = 0.f;
        if (p < npass && l < t.n) {
            float d = 0.f;
            for (int k = 0; k < SN_CH; ++k) d += t.dots[((size_t)p * SnTable::MAXS + l) * SN_CH + k];
            is = t.sig[(t.slot[p] * 2 + 1) * SnTable::MAXS + l];
            c = d * is * is;
        }
        coef[p][l] = c; isg[p][l] = is;
    }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    int layer = -1;
    for (int l = 0; l < t.n; ++l)
        if (i >= t.layer[l].w_off && i < t.layer[l].w_off + (int64_t)t.layer[l].rows * t.layer[l].K) layer = l;
    float v = 0.f;
    if (layer < 0) {
        v = g0[i] + (npass > 1 ? g1[i] : 0.f);
    } else {
        const SnLayer& L = t.layer[layer];
        const int64_t e = i - L.w_off;
        const int row = (int)(e / L.K), k = (int)(e - (int64_t)row * L.K);
        for (int p = 0; p < npass; ++p) {
            const float* g = p == 0 ? g0 : g1;
            const float* us = t.u_saved + (size_t)t.slot[p] * t.u_total + L.u_off;
            const float* vs = t.v_saved + (size_t)t.slot[p] * t.v_total + L.v_off;
            v += g[i] * isg[p][layer] - coef[p][layer] * us[row] * vs[k];
        }
    }
    out[i] = v;
}
void launch_sn_combine(const SnTable& t, const float* g0, const float* g1, float* out, int64_t total, int npass, hipStream_t s) {
    hipLaunchKernelGGL(k_sn_dots, dim3(t.n * SN_CH), dim3(256), 0, s, t, g0, g1, npass);
    hipLaunchKernelGGL(k_sn_combine, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, t, g0, g1, out, total, npass);
}

}  // namespace siggan
