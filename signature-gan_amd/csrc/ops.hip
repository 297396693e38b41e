// ops.hip -- bandwidth-bound kernels of the signature-GAN step (gfx950): RNG, the C=1 end layers,
// BatchNorm statistics / apply / backward, classifier + BCE, bias reductions, clip + Adam.
// Every activation is NHWC fp32; consecutive lanes walk the channel axis (coalesced 128-256 B
// segments) and per-channel reductions are two-stage (per-chunk partials, then a finalize
// kernel that adds the partials in chunk order), so results are bitwise reproducible.
#include "ops.h"
#include "rng.h"

namespace siggan {

static inline int cdiv(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// =========================================================================================
// RNG (rng.h): Philox4x32-10
// =========================================================================================
__device__ __forceinline__ uint4 draw(const DevState* st, uint64_t idx, uint32_t stream_id, uint32_t ctr_add = 0) {
    return draw_raw(st->seed, st->rng_ctr + ctr_add, idx, stream_id);
}

__global__ void k_randn(float* __restrict__ out, int64_t n, const DevState* __restrict__ st, uint32_t sid) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t * 4 >= n) return;
    const f32x4 v = normal4(draw(st, (uint64_t)t, sid));
    for (int j = 0; j < 4 && t * 4 + j < n; ++j) out[t * 4 + j] = v[j];
}
__global__ void k_mask_to_noise(const float* __restrict__ m, float* __restrict__ out, int64_t n, float inv) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = m[i] * inv;
}
__global__ void k_tick(DevState* st) { st->rng_ctr += 1; }

void launch_randn(float* out, int64_t n, const DevState* st, uint32_t sid, hipStream_t s) {
    hipLaunchKernelGGL(k_randn, dim3(cdiv((n + 3) / 4, 256)), dim3(256), 0, s, out, n, st, sid);
}
// all layers' tables in one launch: table t covers threads [pre[t], pre[t+1])
struct NoiseTable { float* out[8]; int64_t n[8]; int64_t elem0[8]; uint32_t sid[8]; int64_t pre[9]; int nt; };
__global__ void k_dropnoise_multi(const NoiseTable tb, float keep, float inv, const DevState* __restrict__ st, uint32_t ctr_add) {
    const int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= tb.pre[tb.nt]) return;
    int t = 0;
    while (g >= tb.pre[t + 1]) ++t;
    const int64_t q = g - tb.pre[t];
    const uint4 r = draw(st, (uint64_t)(tb.elem0[t] / 4 + q), tb.sid[t], ctr_add);
    const uint32_t x[4] = {r.x, r.y, r.z, r.w};
    float* o = tb.out[t];
    for (int j = 0; j < 4 && q * 4 + j < tb.n[t]; ++j) o[q * 4 + j] = u01(x[j]) < keep ? inv : 0.f;
}
void launch_dropnoise_multi(int nt, float* const* out, const int64_t* n, const int64_t* elem0, const uint32_t* sid, float keep,
                            const DevState* st, hipStream_t s, uint32_t ctr_add) {
    NoiseTable tb; tb.nt = nt; tb.pre[0] = 0;
    for (int t = 0; t < nt; ++t) { tb.out[t] = out[t]; tb.n[t] = n[t]; tb.elem0[t] = elem0[t]; tb.sid[t] = sid[t]; tb.pre[t + 1] = tb.pre[t] + (n[t] + 3) / 4; }
    hipLaunchKernelGGL(k_dropnoise_multi, dim3(cdiv(tb.pre[nt], 256)), dim3(256), 0, s, tb, keep, 1.0f / keep, st, ctr_add);
}
void launch_mask_to_noise(const float* mask, float* out, int64_t n, float keep, hipStream_t s) {
    hipLaunchKernelGGL(k_mask_to_noise, dim3(cdiv(n, 256)), dim3(256), 0, s, mask, out, n, 1.0f / keep);
}
void launch_tick(DevState* st, hipStream_t s) { hipLaunchKernelGGL(k_tick, dim3(1), dim3(1), 0, s, st); }

__device__ __forceinline__ int perm16(int c, int perm_c0) { return perm_c0 > 0 ? (c % perm_c0) * 16 + c / perm_c0 : c; }

// the optimiser's per-element arithmetic (k_adam, k_adam_pack and the riders of k_adam_pack share it: bitwise the same values)
struct AdamK { float mul, ss, bc2, w1, beta2, w2, eps; };     // w1 = (float)(1 - beta1), w2 = (float)(1 - beta2): formed in double on the host, as torch does
__device__ __forceinline__ void adam_upd(const AdamK& k, float& pp, float& gg, float& mm, float& vv) {
    const float gr = gg * k.mul;
    gg = gr;
    // exp_avg.lerp_(grad, 1 - beta1)
    mm = k.w1 < 0.5f ? mm + k.w1 * (gr - mm) : gr - (gr - mm) * (1.0f - k.w1);
    // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    vv = vv * k.beta2 + k.w2 * gr * gr;
    const float denom = sqrtf(vv) / k.bc2 + k.eps;
    pp = pp + (k.ss * mm) / denom;
}

// =========================================================================================
// fused prepare pass (weight packs, fc / classifier permutes, BatchNorm eval folding)
// =========================================================================================
// One workgroup = one unit of a job; every unit is a small transpose staged through LDS so that both
// the global reads and the global writes are contiguous runs:
//   PACK_DOWN  unit = output channel o : w[o][i][tap]     -> dst[o][tap*I + i]          (I*16 floats)
//   PACK_UP    unit = output channel o : w[i][o][kh][kw]  -> dst[cls][o][t*I + i]       (I*16 floats)
//   FC_T       unit = 64 features x <=128 latent dims    : W[f][k] -> Wt[k][f']
//   TAPS       unit = 256 elements of a one-channel conv's weight (final conv, first D block): w[c][tap] -> dst[tap][c]
//   BN_EVAL    unit = 256 channels
static long long prep_units(const PrepJob& j) {
    switch (j.type) {
        case PREP_PACK_DOWN: case PREP_CLS: return j.type == PREP_CLS ? 1 : j.O;
        case PREP_PACK_UP: return j.O;
        case PREP_FC_T: return (long long)(j.I * 16 / 64) * ((j.O + 127) / 128);
        case PREP_SCALE: return (j.O + 1023) / 1024;
        case PREP_TAPS: return (j.O * j.I + 255) / 256;
        default: return (j.O + 255) / 256;
    }
}
void prep_add(PrepTable& t, const PrepJob& j, long long /*count*/) {
    if (t.njobs >= PrepTable::MAXJ) { t.overflow = 1; return; }   // launch_prepare refuses an incomplete table
    if (t.njobs == 0) t.prefix[0] = 0;
    t.job[t.njobs] = j;
    t.prefix[t.njobs + 1] = t.prefix[t.njobs] + prep_units(j);
    ++t.njobs;
}
// a packed weight goes out in the element type the consuming kernel reads (q.dt: fp32, or bf16 / f16 rounded to nearest)
__device__ __forceinline__ void put_w(const PrepJob& q, size_t idx, float v) {
    if (q.mul) v *= q.mul[0];
    if (q.dt == DT_F32) q.dst[idx] = v;
    else if (q.dt == DT_BF16) reinterpret_cast<bf16_t*>(q.dst)[idx] = (bf16_t)v;
    else reinterpret_cast<f16_t*>(q.dst)[idx] = (f16_t)v;
}
__device__ __forceinline__ void prepare_unit(const PrepTable& t, float eps, unsigned bid) {
    extern __shared__ float tile[];
    int j = 0;
    while ((long long)bid >= t.prefix[j + 1]) ++j;
    const PrepJob& q = t.job[j];
    const int u = (int)(bid - t.prefix[j]);
    const int tid = threadIdx.x;
    if (q.type == PREP_PACK_DOWN || q.type == PREP_CLS) {
        const int I = q.type == PREP_CLS ? q.O : q.I;
        const float* src = q.src + (size_t)u * I * 16;
#pragma unroll 4
        for (int e = tid; e < I * 16; e += 256) tile[(e >> 4) * 17 + (e & 15)] = src[e];
        __syncthreads();
        const size_t d0 = (size_t)u * I * 16;
        if ((I & (I - 1)) == 0) {                  // channel counts are powers of two: shifts instead of a division per element
            const int lgI = 31 - __builtin_clz(I);
#pragma unroll 4
            for (int e = tid; e < I * 16; e += 256) { const int tap = e >> lgI, i = e & (I - 1); put_w(q, d0 + e, tile[i * 17 + tap]); }
        } else {
            for (int e = tid; e < I * 16; e += 256) { const int tap = e / I, i = e - tap * I; put_w(q, d0 + e, tile[i * 17 + tap]); }
        }
    } else if (q.type == PREP_PACK_UP) {
        const int I = q.I, O = q.O;
#pragma unroll 4
        for (int e = tid; e < I * 16; e += 256) tile[(e >> 4) * 17 + (e & 15)] = q.src[((size_t)(e >> 4) * O + u) * 16 + (e & 15)];
        __syncthreads();
        if ((I & (I - 1)) == 0) {
            const int lgI = 31 - __builtin_clz(I);
#pragma unroll 4
            for (int e = tid; e < I * 16; e += 256) {
                const int i = e & (I - 1), tt = (e >> lgI) & 3, cls = e >> (lgI + 2);
                const int kh = 1 - (cls >> 1) + 2 * (tt >> 1), kw = 1 - (cls & 1) + 2 * (tt & 1);
                put_w(q, ((size_t)cls * O + u) * 4 * I + tt * I + i, tile[i * 17 + kh * 4 + kw]);
            }
        } else {
            for (int e = tid; e < I * 16; e += 256) {
                const int i = e % I, tt = (e / I) & 3, cls = e / (4 * I);
                const int kh = 1 - (cls >> 1) + 2 * (tt >> 1), kw = 1 - (cls & 1) + 2 * (tt & 1);
                put_w(q, ((size_t)cls * O + u) * 4 * I + tt * I + i, tile[i * 17 + kh * 4 + kw]);
            }
        }
    } else if (q.type == PREP_FC_T) {            // K = q.O, C0 = q.I
        const int K = q.O, C0 = q.I, F = C0 * 16, kchunks = (K + 127) / 128;
        const int f0 = (u / kchunks) * 64, k0 = (u % kchunks) * 128, kn = min(128, K - k0);
        for (int e = tid; e < 64 * kn; e += 256) {
            const int r = e / kn, k = e - r * kn, fp = f0 + r;
            tile[r * 129 + k] = q.src[(size_t)((fp % C0) * 16 + fp / C0) * K + k0 + k];
        }
        __syncthreads();
        for (int e = tid; e < 64 * kn; e += 256) { const int k = e >> 6, r = e & 63; q.dst[(size_t)(k0 + k) * F + f0 + r] = tile[r * 129 + k]; }
    } else if (q.type == PREP_TAPS) {            // one-channel convs: w[c][tap] -> dst[tap][c] (I taps, O channels), 256 elements per unit
        const int e = u * 256 + tid;
        if (e < q.O * q.I) q.dst[(e % q.I) * q.O + e / q.I] = q.mul ? q.src[e] * q.mul[0] : q.src[e];
    } else if (q.type == PREP_SCALE) {           // dst = src * mul (fp32), 1024 elements per unit
        const float m = q.mul ? q.mul[0] : 1.0f;
        for (int e = u * 1024 + tid; e < min(q.O, (u + 1) * 1024); e += 256) q.dst[e] = q.src[e] * m;
    } else {                                     // BN eval: [scale | shift | mean | rstd]
        const int C = q.O, c = u * 256 + tid;
        if (c < C) {
            const int tix = perm16(c, q.perm);
            const float rstd = 1.0f / sqrtf(q.src4[tix] + eps);
            const float sc = q.src[tix] * rstd;
            q.dst[c] = sc; q.dst[C + c] = q.src2[tix] - q.src3[tix] * sc; q.dst[2 * C + c] = q.src3[tix]; q.dst[3 * C + c] = rstd;
        }
    }
}
__global__ __launch_bounds__(256) void k_prepare(const PrepTable t, float eps) { prepare_unit(t, eps, blockIdx.x); }
bool launch_prepare(const PrepTable& t, float bn_eps, hipStream_t s) {
    if (t.overflow) return false;
    if (t.njobs == 0) return true;
    // LDS: 512 input channels x 17 floats (conv packs) or 64 x 129 (fc) -- 34.8 KB
    hipLaunchKernelGGL(k_prepare, dim3((unsigned)t.prefix[t.njobs]), dim3(256), 512 * 17 * sizeof(float), s, t, bn_eps);
    return true;
}

// =========================================================================================
// generic two-stage column reduction over an [R][C] fp32 matrix (C % 4 == 0).
// stage 1: a 256-thread block covers cg float4 column groups x (256/cg) row lanes of one row chunk
//          (16-byte loads, lanes walk the channel axis), LDS-combines its row lanes and writes one
//          partial row per chunk;  stage 2 (gather2): 64 columns x 16 lanes add the chunk partials.
// =========================================================================================
struct ColPlan { int cg, cbx, nch, rows; };
static ColPlan col_plan(int64_t R, int C) {
    const int C4 = C / 4;
    ColPlan p;
    p.cg = C4 < 64 ? C4 : 64;              // C4 is a power of two for every layer
    p.cbx = cdiv(C4, p.cg);
    const int rl = 256 / p.cg;
    int nch = 1024 / p.cbx;
    if (nch > 256) nch = 256;
    const int64_t max_ch = (R + 2 * rl - 1) / (2 * rl);
    if (nch > max_ch) nch = (int)max_ch;
    if (nch < 1) nch = 1;
    int rows = (int)((R + nch - 1) / nch);
    rows = ((rows + rl - 1) / rl) * rl;
    p.nch = (int)((R + rows - 1) / rows);
    p.rows = rows;
    return p;
}

__device__ __forceinline__ void add4(float4& a, const float4 b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

template <class F>
__global__ __launch_bounds__(256) void k_colreduce(F f, int64_t R, int C, int cg, int rows, float* __restrict__ p0,
                                                   float* __restrict__ p1) {
    __shared__ float4 sh[2][256];
    const int C4 = C / 4, rl = 256 / cg;
    const int cl = threadIdx.x % cg, lane = threadIdx.x / cg;
    const int c4 = blockIdx.x * cg + cl;
    const int64_t r0 = (int64_t)blockIdx.y * rows;
    const int64_t r1 = r0 + rows < R ? r0 + rows : R;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (c4 < C4) {
#pragma unroll 8
        for (int64_t r = r0 + lane; r < r1; r += rl) f(r, c4, C4, s0, s1);     // (unrolled: eight rows' loads in flight per lane)
    }
    sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = s1;
    __syncthreads();
    if (lane == 0 && c4 < C4) {
        for (int k = 1; k < rl; ++k) { add4(s0, sh[0][k * cg + cl]); add4(s1, sh[1][k * cg + cl]); }
        reinterpret_cast<float4*>(p0 + (size_t)blockIdx.y * C)[c4] = s0;
        reinterpret_cast<float4*>(p1 + (size_t)blockIdx.y * C)[c4] = s1;
    }
}

// stage 2: a 1024-thread block finalizes 64 columns; 16 lanes add the chunk partials, LDS combines
// the lanes in a fixed order (bitwise reproducible).
template <int W = 64>
__device__ __forceinline__ void gather2(const float* __restrict__ p0, const float* __restrict__ p1, int nch, int C,
                                        float& s, float& q, float (*sh)[16][64], int bx = -1) {
    // W column lanes x (blockDim / W) row lanes; the LDS block is used as [2][blockDim / W][W]
    const int cl = threadIdx.x & (W - 1), rl = threadIdx.x / W, nl = blockDim.x / W;
    const int c = (bx < 0 ? (int)blockIdx.x : bx) * W + cl;
    float* const s0 = &sh[0][0][0];
    float* const s1 = &sh[1][0][0];
    float a = 0.f, b = 0.f;
    if (c < C) {            // (eight partial rows' loads in flight per lane; the sums keep their order)
        if (p1) {
#pragma unroll 8
            for (int k = rl; k < nch; k += nl) { a += p0[(size_t)k * C + c]; b += p1[(size_t)k * C + c]; }
        } else {
#pragma unroll 8
            for (int k = rl; k < nch; k += nl) a += p0[(size_t)k * C + c];
        }
    }
    s0[rl * W + cl] = a; s1[rl * W + cl] = b;
    __syncthreads();
    s = 0.f; q = 0.f;
    for (int k = 0; k < nl; ++k) { s += s0[k * W + cl]; q += s1[k * W + cl]; }
}

__device__ __forceinline__ float4 f4(const f32x4 v) { return make_float4(v[0], v[1], v[2], v[3]); }
template <class T>
struct FStats {   // shifted sums around the first row: robust single-pass variance
    const T* y;
    __device__ void operator()(int64_t r, int c4, int C4, float4& s0, float4& s1) const {
        const float4 v = f4(ld4<T>(y + (r * C4 + c4) * 4)), p = f4(ld4<T>(y + c4 * 4));
        const float4 d = make_float4(v.x - p.x, v.y - p.y, v.z - p.z, v.w - p.w);
        add4(s0, d);
        s1.x = fmaf(d.x, d.x, s1.x); s1.y = fmaf(d.y, d.y, s1.y); s1.z = fmaf(d.z, d.z, s1.z); s1.w = fmaf(d.w, d.w, s1.w);
    }
};
template <class T>
struct FBnBwd {   // relu mask re-derived from y (a > 0 <=> fma(y, scale, shift) > 0, k_bn_relu's own expression): a is not read
    const T* da; const T* y; const float4* bn;
    __device__ void operator()(int64_t r, int c4, int C4, float4& s0, float4& s1) const {
        const size_t i = (size_t)r * C4 + c4;
        const float4 g = f4(ld4<T>(da + i * 4)), yy = f4(ld4<T>(y + i * 4)), sc = bn[c4], sf = bn[C4 + c4], mu = bn[2 * C4 + c4], rs = bn[3 * C4 + c4];
        const float4 d = make_float4(fmaf(yy.x, sc.x, sf.x) > 0.f ? g.x : 0.f, fmaf(yy.y, sc.y, sf.y) > 0.f ? g.y : 0.f,
                                     fmaf(yy.z, sc.z, sf.z) > 0.f ? g.z : 0.f, fmaf(yy.w, sc.w, sf.w) > 0.f ? g.w : 0.f);
        add4(s0, d);
        s1.x = fmaf(d.x, (yy.x - mu.x) * rs.x, s1.x); s1.y = fmaf(d.y, (yy.y - mu.y) * rs.y, s1.y);
        s1.z = fmaf(d.z, (yy.z - mu.z) * rs.z, s1.z); s1.w = fmaf(d.w, (yy.w - mu.w) * rs.w, s1.w);
    }
};


// =========================================================================================
// BatchNorm
// =========================================================================================
template <int W, class T>
__global__ __launch_bounds__(1024) void k_bn_train_fin(const float* __restrict__ p0, const float* __restrict__ p1, int nch, int64_t R, int C,
                               const T* __restrict__ y, const float* __restrict__ gamma,
                               const float* __restrict__ beta, float* __restrict__ rmean, float* __restrict__ rvar,
                               int64_t* __restrict__ batches, float* __restrict__ bn, int perm_c0, float momentum,
                               float eps) {
    __shared__ float sh[2][16][64];
    float s, q;
    gather2<W>(p0, p1, nch, C, s, q, sh);
    const int c = blockIdx.x * W + (threadIdx.x & (W - 1));
    if (blockIdx.x == 0 && threadIdx.x == 0 && batches) batches[0] += 1;
    if (threadIdx.x >= W || c >= C) return;
    const float invR = 1.0f / (float)R;
    const float d = s * invR;
    const float mean = ld1<T>(y + c) + d;
    float var = q * invR - d * d;
    var = var > 0.f ? var : 0.f;
    const float rstd = 1.0f / sqrtf(var + eps);
    const int t = perm16(c, perm_c0);
    const float sc = gamma[t] * rstd;
    bn[c] = sc; bn[C + c] = beta[t] - mean * sc; bn[2 * C + c] = mean; bn[3 * C + c] = rstd;
    const float unb = R > 1 ? var * ((float)R / (float)(R - 1)) : var;
    rmean[t] = momentum * mean + (1.0f - momentum) * rmean[t];
    rvar[t] = momentum * unb + (1.0f - momentum) * rvar[t];
}
void launch_bn_train_stats(int dt, const void* yv, int64_t R, int C, const float* gamma, const float* beta, float* rmean,
                           float* rvar, int64_t* batches, float* bn, float* partial, int perm_c0, float momentum,
                           float eps, hipStream_t s) {
    const ColPlan pl = col_plan(R, C);
    float* p0 = partial; float* p1 = partial + (size_t)pl.nch * C;
    SIGGAN_DT_SWITCH(dt, T, {
        const T* y = (const T*)yv;
        hipLaunchKernelGGL((k_colreduce<FStats<T>>), dim3(pl.cbx, pl.nch), dim3(256), 0, s, FStats<T>{y}, R, C, pl.cg, pl.rows, p0, p1);
        if (C <= 32)
            hipLaunchKernelGGL((k_bn_train_fin<32, T>), dim3(cdiv(C, 32)), dim3(1024), 0, s, p0, p1, pl.nch, R, C, y, gamma, beta, rmean,
                               rvar, batches, bn, perm_c0, momentum, eps);
        else
            hipLaunchKernelGGL((k_bn_train_fin<64, T>), dim3(cdiv(C, 64)), dim3(1024), 0, s, p0, p1, pl.nch, R, C, y, gamma, beta, rmean,
                               rvar, batches, bn, perm_c0, momentum, eps);
    });
}

template <class T>
__global__ void k_bn_relu(const T* __restrict__ y, T* __restrict__ a, int64_t n4, int C4,
                          const float4* __restrict__ bn) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (C4 & (C4 - 1)) == 0 ? (int)(i & (C4 - 1)) : (int)(i % C4);      // (channel counts are powers of two)
    const float4 v = f4(ld4<T>(y + i * 4)), sc = bn[c], sh = bn[C4 + c];
    st4<T>(a + i * 4, f32x4{fmaxf(fmaf(v.x, sc.x, sh.x), 0.f), fmaxf(fmaf(v.y, sc.y, sh.y), 0.f),
                            fmaxf(fmaf(v.z, sc.z, sh.z), 0.f), fmaxf(fmaf(v.w, sc.w, sh.w), 0.f)});
}
void launch_bn_relu(int dt, const void* y, void* a, int64_t R, int C, const float* bn, hipStream_t s) {
    const int64_t n4 = R * C / 4;
    SIGGAN_DT_SWITCH(dt, T, hipLaunchKernelGGL(k_bn_relu<T>, dim3(cdiv(n4, 256)), dim3(256), 0, s, (const T*)y, (T*)a, n4, C / 4,
                                                (const float4*)bn));
}

template <int W>
__global__ __launch_bounds__(1024) void k_bn_bwd_fin(const float* __restrict__ p0, const float* __restrict__ p1, int nch, int64_t R, int C,
                             float* __restrict__ bn, float* __restrict__ dgamma, float* __restrict__ dbeta, int perm_c0) {
    __shared__ float sh[2][16][64];
    float s, q;
    gather2<W>(p0, p1, nch, C, s, q, sh);
    const int c = blockIdx.x * W + (threadIdx.x & (W - 1));
    if (threadIdx.x >= W || c >= C) return;
    const int t = perm16(c, perm_c0);
    dbeta[t] = s; dgamma[t] = q;
    const float invR = 1.0f / (float)R;
    bn[4 * C + c] = s * invR; bn[5 * C + c] = q * invR;
}
static void launch_bn_bwd_fin(const float* p0, const float* p1, int nch, int64_t R, int C, float* bn, float* dgamma,
                              float* dbeta, int perm_c0, hipStream_t s) {
    if (C <= 32) hipLaunchKernelGGL(k_bn_bwd_fin<32>, dim3(cdiv(C, 32)), dim3(1024), 0, s, p0, p1, nch, R, C, bn, dgamma, dbeta, perm_c0);
    else hipLaunchKernelGGL(k_bn_bwd_fin<64>, dim3(cdiv(C, 64)), dim3(1024), 0, s, p0, p1, nch, R, C, bn, dgamma, dbeta, perm_c0);
}
template <class T>
__global__ void k_bn_bwd_apply(T* __restrict__ da, const T* __restrict__ y, int64_t n4, int C4,
                               const float4* __restrict__ bn) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (C4 & (C4 - 1)) == 0 ? (int)(i & (C4 - 1)) : (int)(i % C4);      // (channel counts are powers of two)
    const float4 g = f4(ld4<T>(da + i * 4)), yy = f4(ld4<T>(y + i * 4));
    const float4 sc = bn[c], sf = bn[C4 + c], mu = bn[2 * C4 + c], rs = bn[3 * C4 + c], c1 = bn[4 * C4 + c], c2 = bn[5 * C4 + c];
    float4 o;
    o.x = sc.x * ((fmaf(yy.x, sc.x, sf.x) > 0.f ? g.x : 0.f) - c1.x - (yy.x - mu.x) * rs.x * c2.x);
    o.y = sc.y * ((fmaf(yy.y, sc.y, sf.y) > 0.f ? g.y : 0.f) - c1.y - (yy.y - mu.y) * rs.y * c2.y);
    o.z = sc.z * ((fmaf(yy.z, sc.z, sf.z) > 0.f ? g.z : 0.f) - c1.z - (yy.z - mu.z) * rs.z * c2.z);
    o.w = sc.w * ((fmaf(yy.w, sc.w, sf.w) > 0.f ? g.w : 0.f) - c1.w - (yy.w - mu.w) * rs.w * c2.w);
    st4<T>(da + i * 4, f32x4{o.x, o.y, o.z, o.w});
}
// k_bn_bwd_fin folded into k_bn_bwd_apply (round 4; used when the producing GEMM left partial rows): a workgroup owns a
// 32-channel slice x a row chunk, and its prologue adds the producer's nrows partial rows for ITS 32 channels (row lane rl takes
// rows rl, rl + 32, ...; the 32 lanes are then added in order: a fixed order, so every workgroup of a slice forms the same
// bits).  64-128 KB of L2-resident rows per workgroup instead of a 1-4 workgroup finalize launch on the Generator backward's
// critical lane: same-box A/B 1.4192 -> 1.4130 ms at fp32, 0.6400 -> 0.6304 at bf16, three launches fewer.
template <class T>
__global__ __launch_bounds__(256) void k_bn_bwd_fin_apply(T* __restrict__ da, const T* __restrict__ y, int64_t R, int C, float* __restrict__ bn,
                                                          const float* __restrict__ p0, const float* __restrict__ p1, int nrows,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta, int rows_per_chunk) {
    __shared__ f32x4 sh[2][32][8];
    const int c4 = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int ch = blockIdx.x * 32 + c4 * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = s;
#pragma unroll 4
    for (int k = rl; k < nrows; k += 32) {
        s += *reinterpret_cast<const f32x4*>(p0 + (size_t)k * C + ch);
        q += *reinterpret_cast<const f32x4*>(p1 + (size_t)k * C + ch);
    }
    sh[0][rl][c4] = s; sh[1][rl][c4] = q;
    __syncthreads();
    s = sh[0][0][c4]; q = sh[1][0][c4];
#pragma unroll
    for (int k = 1; k < 32; ++k) { s += sh[0][k][c4]; q += sh[1][k][c4]; }
    const float invR = 1.0f / (float)R;
    const f32x4 c1 = s * invR, c2 = q * invR;
    if (blockIdx.y == 0 && rl == 0) {
        *reinterpret_cast<f32x4*>(dbeta + ch) = s; *reinterpret_cast<f32x4*>(dgamma + ch) = q;
        *reinterpret_cast<f32x4*>(bn + 4 * C + ch) = c1; *reinterpret_cast<f32x4*>(bn + 5 * C + ch) = c2;
    }
    const f32x4 sc = *reinterpret_cast<const f32x4*>(bn + ch), sf = *reinterpret_cast<const f32x4*>(bn + C + ch);
    const f32x4 mu = *reinterpret_cast<const f32x4*>(bn + 2 * C + ch), rs = *reinterpret_cast<const f32x4*>(bn + 3 * C + ch);
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = r0 + rows_per_chunk < R ? r0 + rows_per_chunk : R;
#pragma unroll 4
    for (int64_t r = r0 + rl; r < r1; r += 32) {
        const size_t i = (size_t)r * C + ch;
        const f32x4 g = ld4<T>(da + i), yy = ld4<T>(y + i);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            o[e] = sc[e] * ((fmaf(yy[e], sc[e], sf[e]) > 0.f ? g[e] : 0.f) - c1[e] - (yy[e] - mu[e]) * rs[e] * c2[e]);
        st4<T>(da + i, o);
    }
}
void launch_bn_bwd(int dt, void* dav, const void* yv, int64_t R, int C, float* bn, float* partial,
                   float* dgamma, float* dbeta, int perm_c0, hipStream_t s, int pre_rows, hipEvent_t done) {
    // pre_rows > 0: the kernel that produced da left that many partial rows of both sums in `partial` (gconv's
    // EPI_BN_BWD_STATS): no reduction pass over da and y
    const ColPlan pl = col_plan(R, C);
    const int nch = pre_rows > 0 ? pre_rows : pl.nch;
    float* p0 = partial; float* p1 = partial + (size_t)nch * C;
    const int64_t n4 = R * C / 4;
    if (pre_rows > 0 && perm_c0 == 0 && (C % 32) == 0) {
        const int slices = C / 32;
        int64_t chunks = 256 / slices; if (chunks < 1) chunks = 1;
        if (chunks > (R + 63) / 64) chunks = (R + 63) / 64;
        int rpc = (int)((R + chunks - 1) / chunks); rpc = ((rpc + 31) / 32) * 32;
        chunks = (R + rpc - 1) / rpc;
        SIGGAN_DT_SWITCH(dt, T, SIGGAN_LAUNCH_EV(done, k_bn_bwd_fin_apply<T>, dim3(slices, (unsigned)chunks), dim3(256), 0, s, (T*)dav, (const T*)yv, R, C,
                                                  bn, p0, p1, nch, dgamma, dbeta, rpc));
        return;
    }
    SIGGAN_DT_SWITCH(dt, T, {
        T* da = (T*)dav; const T* y = (const T*)yv;
        if (pre_rows <= 0)
            hipLaunchKernelGGL((k_colreduce<FBnBwd<T>>), dim3(pl.cbx, pl.nch), dim3(256), 0, s,
                               FBnBwd<T>{da, y, (const float4*)bn}, R, C, pl.cg, pl.rows, p0, p1);
        launch_bn_bwd_fin(p0, p1, nch, R, C, bn, dgamma, dbeta, perm_c0, s);
        SIGGAN_LAUNCH_EV(done, k_bn_bwd_apply<T>, dim3(cdiv(n4, 256)), dim3(256), 0, s, da, y, n4, C / 4, (const float4*)bn);
    });
}

// =========================================================================================
// Generator fc (latent x weight) -- K = latent_dim is tiny; one thread per output
// =========================================================================================
// (Wt[k][f'] = W[f][k] is the k-major copy k_prepare keeps in the NHWC feature order, so lanes walk f' coalesced)
// one thread = one feature f' x 8 batch rows; z rows broadcast from LDS; 50 weight loads in flight
template <class T>
__global__ __launch_bounds__(256) void k_fc_fwd(const float* __restrict__ z, const float* __restrict__ Wt,
                                                const float* __restrict__ b, T* __restrict__ y, int B, int K, int C0,
                                                const float* __restrict__ bn, const DevState* __restrict__ st, uint32_t sid,
                                                float* __restrict__ z_out) {
    extern __shared__ float sz[];   // [8][K]
    const int F = C0 * 16;
    const int fp = blockIdx.x * 256 + threadIdx.x, nb = blockIdx.y * 8;
    if (z) {
        for (int i = threadIdx.x; i < 8 * K; i += 256) {
            const int n = nb + i / K;
            sz[i] = n < B ? z[(size_t)n * K + i % K] : 0.f;
        }
    } else {
        // z ~ N(0,1) drawn here: the block's 8 rows are elements [nb*K, nb*K + cnt) of the (B, K) tensor k_randn would
        // fill (same Philox draw per group of four elements, same Box-Muller), also written out for the backward pass
        const int64_t e0 = (int64_t)nb * K;                 // nb % 8 == 0: a multiple of 4
        const int rows = B - nb < 8 ? B - nb : 8, cnt = rows * K;
        for (int i = threadIdx.x; i < 2 * K; i += 256) {    // 8*K elements = 2*K groups of four
            const int q = i * 4;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (q < cnt) {
                const f32x4 nv = normal4(draw(st, (uint64_t)(e0 / 4 + i), sid));
                v[0] = nv[0]; v[1] = nv[1]; v[2] = nv[2]; v[3] = nv[3];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool in = q + j < cnt;
                sz[q + j] = in ? v[j] : 0.f;
                if (in && z_out && blockIdx.x == 0) z_out[e0 + q + j] = v[j];
            }
        }
    }
    __syncthreads();
    if (fp >= F) return;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.f;
    const float* wp = Wt + fp;
    int k = 0;
    for (; k + 50 <= K; k += 50) {                   // 50 weight loads in flight (the latent size is 100 or 128)
        float w[50];
#pragma unroll
        for (int u = 0; u < 50; ++u) w[u] = wp[(size_t)(k + u) * F];
#pragma unroll
        for (int u = 0; u < 50; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(sz[j * K + k + u], w[u], acc[j]);
    }
    for (; k + 14 <= K; k += 14) {
        float w[14];
#pragma unroll
        for (int u = 0; u < 14; ++u) w[u] = wp[(size_t)(k + u) * F];
#pragma unroll
        for (int u = 0; u < 14; ++u)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = fmaf(sz[j * K + k + u], w[u], acc[j]);
    }
    for (; k < K; ++k) {
        const float w = wp[(size_t)k * F];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = fmaf(sz[j * K + k], w, acc[j]);
    }
    const float bias = b[(fp % C0) * 16 + fp / C0];
    if (bn) {            // eval mode: BatchNorm1d folded to scale/shift + ReLU, the pre-BN tensor is not kept
        const float sc = bn[fp], sf = bn[F + fp];
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (nb + j < B) st1<T>(y + (size_t)(nb + j) * F + fp, fmaxf(fmaf(acc[j] + bias, sc, sf), 0.f));
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (nb + j < B) st1<T>(y + (size_t)(nb + j) * F + fp, acc[j] + bias);
}
void launch_fc_fwd(int dt, const float* z, const float* Wt, const float* b, void* y, int B, int K, int C0, hipStream_t s,
                   const float* bn_affine_relu, const DevState* st, uint32_t stream_id, float* z_out) {
    SIGGAN_DT_SWITCH(dt, T, hipLaunchKernelGGL(k_fc_fwd<T>, dim3(cdiv(C0 * 16, 256), cdiv(B, 8)), dim3(256), 8 * K * sizeof(float), s, z, Wt,
                                                b, (T*)y, B, K, C0, bn_affine_relu, st, stream_id, z_out));
}
// dW[f][k] = sum_n dy[n][f'] * z[n][k],  db[f] = sum_n dy[n][f'].  A thread owns feature f' and a group of
// FK latent columns: every dy value it loads feeds FK FMAs (z rows broadcast from LDS, 64 batch rows per
// pass), instead of one load per FMA.
static constexpr int FK = 8;
template <class T>
__global__ __launch_bounds__(256) void k_fc_wgrad(const T* __restrict__ dy, const float* __restrict__ z, float* __restrict__ dW,
                           float* __restrict__ db, int B, int K, int C0) {
    __shared__ float sz[64][FK];
    const int F = C0 * 16;
    const int fp = blockIdx.x * 256 + threadIdx.x, k0 = blockIdx.y * FK;
    float acc[FK], sb = 0.f;
#pragma unroll
    for (int u = 0; u < FK; ++u) acc[u] = 0.f;
    for (int n0 = 0; n0 < B; n0 += 64) {
        __syncthreads();
        for (int i = threadIdx.x; i < 64 * FK; i += 256) {
            const int n = n0 + i / FK, k = k0 + i % FK;
            sz[i / FK][i % FK] = (n < B && k < K) ? z[(size_t)n * K + k] : 0.f;
        }
        __syncthreads();
        if (fp < F) {
            const int nn = B - n0 < 64 ? B - n0 : 64;
            int n = 0;
            for (; n + 8 <= nn; n += 8) {
                float g[8];
#pragma unroll
                for (int v = 0; v < 8; ++v) g[v] = ld1<T>(dy + (size_t)(n0 + n + v) * F + fp);
#pragma unroll
                for (int v = 0; v < 8; ++v) {
                    sb += g[v];
#pragma unroll
                    for (int u = 0; u < FK; ++u) acc[u] = fmaf(g[v], sz[n + v][u], acc[u]);
                }
            }
            for (; n < nn; ++n) {
                const float g = ld1<T>(dy + (size_t)(n0 + n) * F + fp);
                sb += g;
#pragma unroll
                for (int u = 0; u < FK; ++u) acc[u] = fmaf(g, sz[n][u], acc[u]);
            }
        }
    }
    if (fp >= F) return;
    const int f = (fp % C0) * 16 + fp / C0;
#pragma unroll
    for (int u = 0; u < FK; ++u)
        if (k0 + u < K) dW[(size_t)f * K + k0 + u] = acc[u];
    if (blockIdx.y == 0) db[f] = sb;
}
void launch_fc_wgrad(int dt, const void* dy, const float* z, float* dW, float* db, int B, int K, int C0, hipStream_t s) {
    SIGGAN_DT_SWITCH(dt, T, hipLaunchKernelGGL(k_fc_wgrad<T>, dim3(cdiv(C0 * 16, 256), cdiv(K, FK)), dim3(256), 0, s, (const T*)dy, z, dW, db,
                                                B, K, C0));
}

// =========================================================================================
// Generator final 3x3 conv (32 -> 1) + tanh, and its backward
// =========================================================================================
// All three kernels: 8 lanes per pixel (4 channels each, weights in registers) x 32 pixels along a
// row; a block owns an RY-row strip of one image and reads the (RY+2) x 3 neighbourhood it needs
// with unconditional loads (clamped address, value selected to 0 outside the image), so every load
// of the strip is in flight at once.
typedef f32x4 f4v;
__device__ __forceinline__ f4v ldg4(const float* p) { return *reinterpret_cast<const f4v*>(p); }
__device__ __forceinline__ int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

struct StripId { int n, y0, x; };
template <int RY>
__device__ __forceinline__ StripId strip_of(int sid, int S, int xi) {
    const int nbx = S >> 5, nby = S / RY;
    StripId r;
    r.x = (sid % nbx) * 32 + xi; sid /= nbx;
    r.y0 = (sid % nby) * RY; r.n = sid / nby;
    return r;
}

// BN: `act` is the last block's PRE-BatchNorm tensor y and bn = [scale | shift]: the activation relu(fma(y, scale, shift))
// (k_bn_relu's own expression) is formed on load and never stored -- in training mode nothing else reads it in the
// forward pass, and the backward pass re-derives it from y as well (k_final_bwd_reduce).
template <class T, bool BN>
__global__ __launch_bounds__(256) void k_final_fwd(const T* __restrict__ act, const float* __restrict__ Wt,
                                                   const float* __restrict__ b, float* __restrict__ img, int S,
                                                   const float* __restrict__ bn) {
    constexpr int RY = 4, C = 32;
    const int c4 = threadIdx.x & 7;
    const StripId t = strip_of<RY>(blockIdx.x, S, threadIdx.x >> 3);
    f4v w[9];                         // Wt = the weight as [tap][c] (k_prepare's PREP_TAPS copy): no LDS transpose, no barrier in front of the strip
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = ldg4(Wt + k * 32 + c4 * 4);
    const float bias = b[0];
    f4v sc = {1.f, 1.f, 1.f, 1.f}, sf = {0.f, 0.f, 0.f, 0.f};
    if (BN) { sc = ldg4(bn + c4 * 4); sf = ldg4(bn + C + c4 * 4); }
    const T* base = act + (size_t)t.n * S * S * C + c4 * 4;
    f4v v[RY + 2][3];
#pragma unroll
    for (int r = 0; r < RY + 2; ++r) {
        const int yy = t.y0 + r - 1, yc = clampi(yy, S - 1);
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int xx = t.x + d - 1, xc = clampi(xx, S - 1);
            f4v q = ld4<T>(base + ((size_t)yc * S + xc) * C);
            if (BN) q = f4v{fmaxf(fmaf(q.x, sc.x, sf.x), 0.f), fmaxf(fmaf(q.y, sc.y, sf.y), 0.f),
                            fmaxf(fmaf(q.z, sc.z, sf.z), 0.f), fmaxf(fmaf(q.w, sc.w, sf.w), 0.f)};
            v[r][d] = (yy == yc && xx == xc) ? q : f4v{0.f, 0.f, 0.f, 0.f};
        }
    }
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const f4v a = v[r + kh][kw], ww = w[kh * 3 + kw];
                acc = fmaf(a.x, ww.x, acc); acc = fmaf(a.y, ww.y, acc); acc = fmaf(a.z, ww.z, acc); acc = fmaf(a.w, ww.w, acc);
            }
        acc += __shfl_xor(acc, 4, 8); acc += __shfl_xor(acc, 2, 8); acc += __shfl_xor(acc, 1, 8);
        if (c4 == 0) img[((size_t)t.n * S + t.y0 + r) * S + t.x] = tanhf(acc + bias);
    }
}
void launch_final_fwd(int dt, const void* act, const float* Wt, const float* b, float* img, int B, int S, int C, hipStream_t s,
                      const float* bn, hipEvent_t done) {
    (void)C;                                            // host checks C == 32, S % 32 == 0
    const dim3 grid(B * (S / 4) * (S / 32));
    SIGGAN_DT_SWITCH(dt, T, {
        if (bn) SIGGAN_LAUNCH_EV(done, (k_final_fwd<T, true>), grid, dim3(256), 0, s, (const T*)act, Wt, b, img, S, bn);
        else SIGGAN_LAUNCH_EV(done, (k_final_fwd<T, false>), grid, dim3(256), 0, s, (const T*)act, Wt, b, img, S, bn);
    });
}

// d(act)[y][x][c] = sum_{kh,kw} dpre[y + 1 - kh][x + 1 - kw] * W[c][kh][kw], for the 4 channels of a lane.
// The activation gradient is never stored: BatchNorm's backward of the last Generator block needs it
// twice (statistics, then apply) and recomputing it from the 1-channel dpre is 36 FMAs against a
// 33 MB round trip.  d[r][k] = dpre row (y0 - 1 + r), column (x - 1 + k), zero outside the image.
// The strip's dpre neighbourhood -- (RY + 2) rows x 34 columns of a one-channel image, zero outside it -- goes through LDS:
// one element per thread, one coalesced load each (the eight channel lanes of a pixel used to load the same 18 scalars, and
// hipcc turned the row-validity tests -- uniform per strip -- into branches with a vmcnt(0) behind each: twelve dependent
// round trips in front of the strip's arithmetic).
template <int RY>
__device__ __forceinline__ float dpre_patch_elem(const float* __restrict__ dpre, int n, int y0, int xb, int S, int idx) {
    constexpr int PW = 34;
    const int r = idx / PW, k = idx - r * PW;
    const int yy = y0 + r - 1, xx = xb + k - 1, yc = clampi(yy, S - 1), xc = clampi(xx, S - 1);
    const float q = dpre[((size_t)n * S + yc) * S + xc];
    return (yy == yc && xx == xc && idx < (RY + 2) * PW) ? q : 0.f;
}
template <int RY>
__device__ __forceinline__ void read_dpre_patch(const float* sp, int xi, float (&d)[RY + 2][3]) {
#pragma unroll
    for (int r = 0; r < RY + 2; ++r)
#pragma unroll
        for (int k = 0; k < 3; ++k) d[r][k] = sp[r * 34 + xi + k];
}
template <int RY>
__device__ __forceinline__ f4v final_dact(const float (&d)[RY + 2][3], const f4v (&w)[9], int r) {
    f4v acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const float g = d[r + 2 - kh][2 - kw];
            const f4v ww = w[kh * 3 + kw];
            acc.x = fmaf(g, ww.x, acc.x); acc.y = fmaf(g, ww.y, acc.y); acc.z = fmaf(g, ww.z, acc.z); acc.w = fmaf(g, ww.w, acc.w);
        }
    return acc;
}

// Backward through [final conv] <- relu <- BatchNorm of the last Generator block, stage 1 -- ONE read of the pre-BatchNorm
// tensor y serves both consumers of the block's output:
//   * BatchNorm backward: per-channel sums of dy_relu and dy_relu * xhat, dy_relu = relu'(.) * d(act) with d(act) of the final
//     conv recomputed from the 1-channel dpre (never stored) and the relu mask re-derived from y (a > 0 <=> fma(y, scale,
//     shift) > 0, the forward's own expression);
//   * the final conv's weight gradient dW[c][kh][kw] = sum act[n][y][x][c] * dpre[n][y - kh + 1][x - kw + 1], db = sum dpre,
//     with act = relu(fma(y, scale, shift)) re-derived the same way (the activation tensor is not materialised in training).
// A block walks 8-row strips (grid-stride), folds the 32 pixel lanes (shuffles inside a wave, LDS across the 4 waves, fixed
// order) and writes one partial row per output family; k_bn_bwd_fin / k_rows_sum add the rows.
template <class T, int RY>
__global__ __launch_bounds__(256) void k_final_bwd_reduce(const float* __restrict__ dpre, const float* __restrict__ Wt,
                                                          const T* __restrict__ y, const float* __restrict__ bn,
                                                          float* __restrict__ p0, float* __restrict__ p1,
                                                          float* __restrict__ pw, int S, int nstrips) {
    constexpr int C = 32;
    __shared__ f4v sh[2][4][8];
    __shared__ float shw[4][8][37];
    const int c4 = threadIdx.x & 7, wave = threadIdx.x >> 6;
    f4v w[9];                         // Wt = the weight as [tap][c] (k_prepare's PREP_TAPS copy): no LDS transpose, no barrier in front of the strip
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = ldg4(Wt + k * 32 + c4 * 4);
    const f4v sc = ldg4(bn + c4 * 4), sf = ldg4(bn + C + c4 * 4), mu = ldg4(bn + 2 * C + c4 * 4), rs = ldg4(bn + 3 * C + c4 * 4);
    f4v s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    f4v acc[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = f4v{0.f, 0.f, 0.f, 0.f};
    float sdb = 0.f;
    // the y rows of the NEXT strip are requested before the current strip is reduced (a block walks several strips when the
    // launcher caps the grid): the loads of strip i+1 fly under the ~300 FMAs per row of strip i
    __shared__ float sp[2][(RY + 2) * 34];
    const int xi = threadIdx.x >> 3;
    f4v yn[RY];
    float pn;                                  // this thread's element of the next strip's dpre patch
    {
        const StripId t0 = strip_of<RY>(blockIdx.x < (unsigned)nstrips ? blockIdx.x : 0, S, xi);
        const T* yb = y + (((size_t)t0.n * S + t0.y0) * S + t0.x) * C + c4 * 4;
#pragma unroll
        for (int r = 0; r < RY; ++r) yn[r] = ld4<T>(yb + (size_t)r * S * C);
        pn = dpre_patch_elem<RY>(dpre, t0.n, t0.y0, t0.x - xi, S, threadIdx.x);
    }
    int it = 0;
    for (int sid = blockIdx.x; sid < nstrips; sid += gridDim.x, it ^= 1) {
        if (threadIdx.x < (RY + 2) * 34) sp[it][threadIdx.x] = pn;
        __syncthreads();                       // (buffers alternate: the strip before last is read out by every thread by now)
        f4v yv[RY];
#pragma unroll
        for (int r = 0; r < RY; ++r) yv[r] = yn[r];
        {
            const int nx = sid + gridDim.x < nstrips ? sid + gridDim.x : sid;
            const StripId tn = strip_of<RY>(nx, S, xi);
            const T* yb = y + (((size_t)tn.n * S + tn.y0) * S + tn.x) * C + c4 * 4;
#pragma unroll
            for (int r = 0; r < RY; ++r) yn[r] = ld4<T>(yb + (size_t)r * S * C);
            pn = dpre_patch_elem<RY>(dpre, tn.n, tn.y0, tn.x - xi, S, threadIdx.x);
        }
        float d[RY + 2][3];
        read_dpre_patch<RY>(sp[it], xi, d);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const f4v g = final_dact<RY>(d, w, r);
            const f4v v = yv[r];
            const f4v pre = {fmaf(v.x, sc.x, sf.x), fmaf(v.y, sc.y, sf.y), fmaf(v.z, sc.z, sf.z), fmaf(v.w, sc.w, sf.w)};
            const f4v m = {pre.x > 0.f ? g.x : 0.f, pre.y > 0.f ? g.y : 0.f, pre.z > 0.f ? g.z : 0.f, pre.w > 0.f ? g.w : 0.f};
            s0 += m;
            s1.x = fmaf(m.x, (v.x - mu.x) * rs.x, s1.x); s1.y = fmaf(m.y, (v.y - mu.y) * rs.y, s1.y);
            s1.z = fmaf(m.z, (v.z - mu.z) * rs.z, s1.z); s1.w = fmaf(m.w, (v.w - mu.w) * rs.w, s1.w);
            const f4v a = {fmaxf(pre.x, 0.f), fmaxf(pre.y, 0.f), fmaxf(pre.z, 0.f), fmaxf(pre.w, 0.f)};
            sdb += d[r + 1][1];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh)
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) {
                    const float gg = d[r + 2 - kh][2 - kw];
                    f4v& q = acc[kh * 3 + kw];
                    q.x = fmaf(a.x, gg, q.x); q.y = fmaf(a.y, gg, q.y); q.z = fmaf(a.z, gg, q.z); q.w = fmaf(a.w, gg, q.w);
                }
        }
    }
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
        s0.x += __shfl_xor(s0.x, o); s0.y += __shfl_xor(s0.y, o); s0.z += __shfl_xor(s0.z, o); s0.w += __shfl_xor(s0.w, o);
        s1.x += __shfl_xor(s1.x, o); s1.y += __shfl_xor(s1.y, o); s1.z += __shfl_xor(s1.z, o); s1.w += __shfl_xor(s1.w, o);
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            acc[k].x += __shfl_xor(acc[k].x, o); acc[k].y += __shfl_xor(acc[k].y, o);
            acc[k].z += __shfl_xor(acc[k].z, o); acc[k].w += __shfl_xor(acc[k].w, o);
        }
        sdb += __shfl_xor(sdb, o);
    }
    if ((threadIdx.x & 63) < 8) {
        sh[0][wave][c4] = s0; sh[1][wave][c4] = s1;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            shw[wave][c4][0 * 9 + k] = acc[k].x; shw[wave][c4][1 * 9 + k] = acc[k].y;
            shw[wave][c4][2 * 9 + k] = acc[k].z; shw[wave][c4][3 * 9 + k] = acc[k].w;
        }
        shw[wave][c4][36] = sdb;
    }
    __syncthreads();
    if (threadIdx.x < 8) {
        const f4v a = ((sh[0][0][c4] + sh[0][1][c4]) + sh[0][2][c4]) + sh[0][3][c4];
        const f4v b = ((sh[1][0][c4] + sh[1][1][c4]) + sh[1][2][c4]) + sh[1][3][c4];
        *reinterpret_cast<f4v*>(p0 + (size_t)blockIdx.x * C + c4 * 4) = a;
        *reinterpret_cast<f4v*>(p1 + (size_t)blockIdx.x * C + c4 * 4) = b;
    }
    float* out = pw + (size_t)blockIdx.x * (C * 9 + 1);
    for (int o = threadIdx.x; o < C * 9 + 1; o += 256) {
        const int g = o < C * 9 ? o / 36 : 0, j = o < C * 9 ? o % 36 : 36;     // channel c = 4*g + j/9, tap j%9
        out[o] = ((shw[0][g][j] + shw[1][g][j]) + shw[2][g][j]) + shw[3][g][j];
    }
}
// stage 2 (after k_bn_bwd_fin): dy = scale * (dy_relu - c1 - xhat * c2), written to dy[B][S][S][C]
template <class T>
__global__ __launch_bounds__(256) void k_final_bnbwd_apply(const float* __restrict__ dpre, const float* __restrict__ Wt,
                                                           const T* __restrict__ y, const float* __restrict__ bn,
                                                           T* __restrict__ dy, int S) {
    constexpr int RY = 4, C = 32;
    const int c4 = threadIdx.x & 7;
    const StripId t = strip_of<RY>(blockIdx.x, S, threadIdx.x >> 3);
    f4v w[9];                         // Wt = the weight as [tap][c] (k_prepare's PREP_TAPS copy): no LDS transpose, no barrier in front of the strip
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = ldg4(Wt + k * 32 + c4 * 4);
    const f4v sc = ldg4(bn + c4 * 4), sf = ldg4(bn + C + c4 * 4), mu = ldg4(bn + 2 * C + c4 * 4), rs = ldg4(bn + 3 * C + c4 * 4);
    const f4v c1 = ldg4(bn + 4 * C + c4 * 4), c2 = ldg4(bn + 5 * C + c4 * 4);
    __shared__ float sp[(RY + 2) * 34];
    const int xi = threadIdx.x >> 3;
    const float pe = dpre_patch_elem<RY>(dpre, t.n, t.y0, t.x - xi, S, threadIdx.x);
    const size_t o0 = (((size_t)t.n * S + t.y0) * S + t.x) * C + c4 * 4;
    f4v yv[RY];
#pragma unroll
    for (int r = 0; r < RY; ++r) yv[r] = ld4<T>(y + o0 + (size_t)r * S * C);
    if (threadIdx.x < (RY + 2) * 34) sp[threadIdx.x] = pe;
    __syncthreads();
    float d[RY + 2][3];
    read_dpre_patch<RY>(sp, xi, d);
#pragma unroll
    for (int r = 0; r < RY; ++r) {
        const f4v g = final_dact<RY>(d, w, r);
        const f4v v = yv[r];
        f4v o;
        o.x = sc.x * ((fmaf(v.x, sc.x, sf.x) > 0.f ? g.x : 0.f) - c1.x - (v.x - mu.x) * rs.x * c2.x);
        o.y = sc.y * ((fmaf(v.y, sc.y, sf.y) > 0.f ? g.y : 0.f) - c1.y - (v.y - mu.y) * rs.y * c2.y);
        o.z = sc.z * ((fmaf(v.z, sc.z, sf.z) > 0.f ? g.z : 0.f) - c1.z - (v.z - mu.z) * rs.z * c2.z);
        o.w = sc.w * ((fmaf(v.w, sc.w, sf.w) > 0.f ? g.w : 0.f) - c1.w - (v.w - mu.w) * rs.w * c2.w);
        st4<T>(dy + o0 + (size_t)r * S * C, o);
    }
}
__global__ __launch_bounds__(1024) void k_rows_sum(const float* __restrict__ partial, int nch, int width, float* __restrict__ o0, int n0,
                           float* __restrict__ o1) {
    // out[j] = sum_k partial[k][j];  j < n0 -> o0[j], else o1[j - n0]
    __shared__ float sh[2][16][64];
    float s, q;
    gather2(partial, nullptr, nch, width, s, q, sh);
    const int j = blockIdx.x * 64 + (threadIdx.x & 63);
    if (threadIdx.x >= 64 || j >= width) return;
    if (j < n0) o0[j] = s; else o1[j - n0] = s;
}
// Both finalizers behind k_final_bwd_reduce in ONE launch (two independent 5 us kernels on the Generator backward's
// critical lane): blocks [0, nbw) add the partial rows of the final conv's weight / bias gradient (k_rows_sum), the block
// behind them finalizes the last block's BatchNorm-backward sums (k_bn_bwd_fin<32>).  Same sums, same order.
__global__ __launch_bounds__(1024) void k_final_fin(const float* __restrict__ partial_w, int nch_w, int width, float* __restrict__ dW,
                                                    int n0, float* __restrict__ db, int nbw, const float* __restrict__ p0,
                                                    const float* __restrict__ p1, int nch, int64_t R, int C, float* __restrict__ bn,
                                                    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ float sh[2][16][64];
    float s, q;
    if ((int)blockIdx.x < nbw) {
        gather2(partial_w, nullptr, nch_w, width, s, q, sh);
        const int j = blockIdx.x * 64 + (threadIdx.x & 63);
        if (threadIdx.x >= 64 || j >= width) return;
        if (j < n0) dW[j] = s; else db[j - n0] = s;
        return;
    }
    const int bx = blockIdx.x - nbw;
    gather2<32>(p0, p1, nch, C, s, q, sh, bx);
    const int c = bx * 32 + (threadIdx.x & 31);
    if (threadIdx.x >= 32 || c >= C) return;
    dbeta[c] = s; dgamma[c] = q;                       // (perm_c0 == 0: identity)
    const float invR = 1.0f / (float)R;
    bn[4 * C + c] = s * invR; bn[5 * C + c] = q * invR;
}
// rows per strip of k_final_bwd_reduce: 4 (8-row strips need 256 registers and measured 36 vs 26 us)
constexpr int FINAL_RY = 4;
// two workgroups of this kernel per CU (188-204 registers): 512 of them walk the strips, each requesting strip i+1's rows while
// it reduces strip i (26.5 -> 23.0 us at batch 64; 1024 blocks x 2 strips: 26.5; one strip per block: slower still)
static int final_reduce_rows(int B, int S) {
    const int n = B * (S / FINAL_RY) * (S / 32); return n < 512 ? n : 512;
}
void launch_final_bwd_reduce(int dt, const float* dpre, const float* Wt, const void* y, int B, int S, int C, const float* bn,
                             float* partial, float* partial_w, hipStream_t s) {
    const int nstrips = B * (S / FINAL_RY) * (S / 32), nch = final_reduce_rows(B, S);
    float* p0 = partial; float* p1 = partial + (size_t)nch * C;
    SIGGAN_DT_SWITCH(dt, T, hipLaunchKernelGGL((k_final_bwd_reduce<T, FINAL_RY>), dim3(nch), dim3(256), 0, s, dpre, Wt, (const T*)y, bn, p0, p1,
                                                partial_w, S, nstrips));
}
void launch_final_bn_bwd_apply(int dt, const float* dpre, const float* Wt, const void* y, void* dy, int B, int S, int C, float* bn,
                               const float* partial, const float* partial_w, float* dW, float* db, float* dgamma, float* dbeta,
                               hipStream_t s, hipEvent_t done) {
    const int nstrips = B * (S / 4) * (S / 32), nch = final_reduce_rows(B, S);
    const float* p0 = partial; const float* p1 = partial + (size_t)nch * C;
    const int nbw = cdiv(C * 9 + 1, 64);
    hipLaunchKernelGGL(k_final_fin, dim3(nbw + cdiv(C, 32)), dim3(1024), 0, s, partial_w, nch, C * 9 + 1, dW, C * 9, db, nbw, p0, p1, nch,
                       (int64_t)B * S * S, C, bn, dgamma, dbeta);
    SIGGAN_DT_SWITCH(dt, T, SIGGAN_LAUNCH_EV(done, k_final_bnbwd_apply<T>, dim3(nstrips), dim3(256), 0, s, dpre, Wt, (const T*)y, bn, (T*)dy, S));
}

// =========================================================================================
// Discriminator first block (1 -> 64, 4x4 stride 2 pad 1) and its backward
// =========================================================================================
__device__ __forceinline__ const float* seg_ptr(const float* x0, int n0, const float* x1, int n, int S) {
    return n < n0 ? x0 + (size_t)n * S * S : x1 + (size_t)(n - n0) * S * S;
}
// zero-padded input rows [2*oh0 - 1, 2*oh0 + 2*RY] x cols [-1, S] of one image into LDS
template <int RY>
__device__ __forceinline__ void stage_x(float* sx, const float* xp, int oh0, int S) {
    const int Wp = S + 2;
    for (int i = threadIdx.x; i < (2 * RY + 2) * Wp; i += 256) {
        const int r = i / Wp, cc = i - r * Wp, ih = 2 * oh0 - 1 + r, iw = cc - 1;
        sx[i] = ((unsigned)ih < (unsigned)S && (unsigned)iw < (unsigned)S) ? xp[ih * S + iw] : 0.f;
    }
}

// thread = 4 output channels (16 taps x 4 weights in registers); 16 channel lanes x 16 pixel lanes;
// a block produces RY output rows of one image from an LDS copy of the input rows (broadcast reads)
// and writes 256 contiguous bytes per pixel.
// RIDE (k_adam_pack's riders): W and b are the arena's values BEFORE this launch's update; the block forms the updated ones
// itself (same arithmetic as the workgroup that owns them) and reports that it has read them
struct Conv1Ride { const float *pw, *gw, *mw, *vw, *pb, *gb, *mb, *vb; unsigned* counter; AdamK k; };
template <class T, bool RIDE = false>
__device__ __forceinline__ void conv1_fwd_block(const float* __restrict__ x0, int n0, const float* __restrict__ x1,
                                                const float* __restrict__ W, const float* __restrict__ b,
                                                const float* __restrict__ noise, float slope,
                                                T* __restrict__ out, int S, unsigned bid, const Conv1Ride* rd = nullptr) {
    constexpr int RY = 2, C = 64;
    __shared__ float sx[(2 * RY + 2) * 130];
    __shared__ __attribute__((aligned(16))) float sw[16 * (C + 4)];  // weights transposed to [tap][co] (row stride C + 4: conflict-free both ways)
    __shared__ __attribute__((aligned(16))) float sb[RIDE ? C : 4];
    const int Ho = S >> 1, nby = Ho / RY, Wp = S + 2;
    const int n = bid / nby, oh0 = (bid % nby) * RY;
    const int q = threadIdx.x & 15, pl = threadIdx.x >> 4;
    stage_x<RY>(sx, seg_ptr(x0, n0, x1, n, S), oh0, S);
    f4v bias;
    if (RIDE) {
        for (int i = threadIdx.x; i < 16 * C; i += 256) {
            float pp = rd->pw[i], gg = rd->gw[i], mm = rd->mw[i], vv = rd->vw[i];
            adam_upd(rd->k, pp, gg, mm, vv);
            sw[(i & 15) * (C + 4) + (i >> 4)] = pp;
        }
        if (threadIdx.x < C) {
            float pp = rd->pb[threadIdx.x], gg = rd->gb[threadIdx.x], mm = rd->mb[threadIdx.x], vv = rd->vb[threadIdx.x];
            adam_upd(rd->k, pp, gg, mm, vv);
            sb[threadIdx.x] = pp;
        }
    } else {
        for (int i = threadIdx.x; i < 16 * C; i += 256) sw[(i & 15) * (C + 4) + (i >> 4)] = W[i];
        bias = ldg4(b + q * 4);
    }
    f4v nz = {1.f, 1.f, 1.f, 1.f};
    if (noise) nz = ldg4(noise + (size_t)n * C + q * 4);
    __syncthreads();
    if (RIDE) {
        // every value this block needs of the ranges is in LDS: the owner may overwrite them (relaxed: the loads have completed)
        if (threadIdx.x == 0) __hip_atomic_fetch_add(rd->counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        bias = *reinterpret_cast<const f4v*>(sb + q * 4);
    }
    f4v w[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) w[t] = *reinterpret_cast<const f4v*>(sw + t * (C + 4) + q * 4);
    for (int p = pl; p < RY * Ho; p += 16) {
        const int r = p / Ho, ow = p - r * Ho;
        const float* xr = sx + (2 * r) * Wp + 2 * ow;
        f4v acc = bias;
#pragma unroll
        for (int kh = 0; kh < 4; ++kh)
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) {
                const float xv = xr[kh * Wp + kw];
                const f4v wv = w[kh * 4 + kw];
                acc.x = fmaf(xv, wv.x, acc.x); acc.y = fmaf(xv, wv.y, acc.y);
                acc.z = fmaf(xv, wv.z, acc.z); acc.w = fmaf(xv, wv.w, acc.w);
            }
        acc.x = acc.x > 0.f ? acc.x : acc.x * slope; acc.y = acc.y > 0.f ? acc.y : acc.y * slope;
        acc.z = acc.z > 0.f ? acc.z : acc.z * slope; acc.w = acc.w > 0.f ? acc.w : acc.w * slope;
        if (noise) { acc.x *= nz.x; acc.y *= nz.y; acc.z *= nz.z; acc.w *= nz.w; }
        st4<T>(out + (((size_t)n * Ho + oh0 + r) * Ho + ow) * C + q * 4, acc);
    }
}
template <class T>
__global__ __launch_bounds__(256) void k_conv1_fwd(const float* __restrict__ x0, int n0, const float* __restrict__ x1,
                                                   const float* __restrict__ W, const float* __restrict__ b,
                                                   const float* __restrict__ noise, float slope,
                                                   T* __restrict__ out, int S) {
    conv1_fwd_block<T>(x0, n0, x1, W, b, noise, slope, out, S, blockIdx.x);
}
// The Discriminator's weight re-packs and a first-block forward that does not read them, in ONE launch: in the G step the two
// stand back to back on the main lane (packs of the just-updated D, then D(fake) of the new images) and share nothing --
// blocks [0, nprep) are k_prepare's units, the blocks behind them k_conv1_fwd's.
template <class T>
__global__ __launch_bounds__(256) void k_prepare_conv1(const PrepTable t, float eps, unsigned nprep, const float* __restrict__ x,
                                                       int nB, const float* __restrict__ W, const float* __restrict__ b,
                                                       float slope, T* __restrict__ out, int S) {
    if (blockIdx.x < nprep) prepare_unit(t, eps, blockIdx.x);
    else conv1_fwd_block<T>(x, nB, x, W, b, nullptr, slope, out, S, blockIdx.x - nprep);
}
bool launch_prepare_conv1(const PrepTable& t, float bn_eps, int dt, const float* x, const float* W, const float* b, float slope,
                          void* out, int B, int S, hipStream_t s) {
    if (t.overflow) return false;
    const unsigned nprep = (unsigned)t.prefix[t.njobs];
    SIGGAN_DT_SWITCH(dt, T, hipLaunchKernelGGL(k_prepare_conv1<T>, dim3(nprep + (unsigned)(B * (S / 4))), dim3(256), 512 * 17 * sizeof(float), s,
                                                t, bn_eps, nprep, x, B, W, b, slope, (T*)out, S));
    return true;
}
void launch_conv1_fwd(int dt, const float* x0, int n0, const float* x1, const float* W, const float* b, const float* noise,
                      float slope, void* out, int B, int S, int C, hipStream_t s) {
    (void)C;                                            // host checks C == 64, S in {64, 128}
    SIGGAN_DT_SWITCH(dt, T, hipLaunchKernelGGL(k_conv1_fwd<T>, dim3(B * (S / 4)), dim3(256), 0, s, x0, n0, x1, W, b, noise, slope, (T*)out, S));
}

// dW[co][kh][kw] = sum dv[n][oh][ow][co] * x[n][2oh-1+kh][2ow-1+kw];  db[co] = sum dv.
// Same thread map as the forward; 17 x 4 sums per thread, folded over the 16 pixel lanes at the end
// (shuffles, then LDS across the waves); partial row = [C*16 weights (co*16 + tap)] [C biases].
template <class T>
__global__ __launch_bounds__(256) void k_conv1_wgrad(const T* __restrict__ dv, const float* __restrict__ x0, int n0,
                                                     const float* __restrict__ x1, float* __restrict__ partial, int S,
                                                     int nstrips) {
    constexpr int RY = 8, C = 64;
    __shared__ float sx[(2 * RY + 2) * 130];
    __shared__ float sh[4][16][69];
    const int Ho = S >> 1, nby = Ho / RY, Wp = S + 2;
    const int q = threadIdx.x & 15, pl = threadIdx.x >> 4, wave = threadIdx.x >> 6;
    f4v acc[17];
#pragma unroll
    for (int t = 0; t < 17; ++t) acc[t] = f4v{0.f, 0.f, 0.f, 0.f};
    for (int sid = blockIdx.x; sid < nstrips; sid += gridDim.x) {
        const int n = sid / nby, oh0 = (sid % nby) * RY;
        __syncthreads();
        stage_x<RY>(sx, seg_ptr(x0, n0, x1, n, S), oh0, S);
        __syncthreads();
        const T* gbase = dv + ((size_t)n * Ho + oh0) * Ho * C + q * 4;
        for (int p0 = pl; p0 < RY * Ho; p0 += 64) {
            f4v g[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) g[u] = ld4<T>(gbase + (size_t)(p0 + 16 * u) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = p0 + 16 * u, r = p / Ho, ow = p - r * Ho;
                const float* xr = sx + (2 * r) * Wp + 2 * ow;
                acc[16] += g[u];
#pragma unroll
                for (int kh = 0; kh < 4; ++kh)
#pragma unroll
                    for (int kw = 0; kw < 4; ++kw) {
                        const float xv = xr[kh * Wp + kw];
                        f4v& a = acc[kh * 4 + kw];
                        a.x = fmaf(g[u].x, xv, a.x); a.y = fmaf(g[u].y, xv, a.y); a.z = fmaf(g[u].z, xv, a.z); a.w = fmaf(g[u].w, xv, a.w);
                    }
            }
        }
    }
#pragma unroll
    for (int o = 16; o < 64; o <<= 1)
#pragma unroll
        for (int t = 0; t < 17; ++t) {
            acc[t].x += __shfl_xor(acc[t].x, o); acc[t].y += __shfl_xor(acc[t].y, o);
            acc[t].z += __shfl_xor(acc[t].z, o); acc[t].w += __shfl_xor(acc[t].w, o);
        }
    if ((threadIdx.x & 63) < 16) {
#pragma unroll
        for (int t = 0; t < 17; ++t) {
            sh[wave][q][0 * 17 + t] = acc[t].x; sh[wave][q][1 * 17 + t] = acc[t].y;
            sh[wave][q][2 * 17 + t] = acc[t].z; sh[wave][q][3 * 17 + t] = acc[t].w;
        }
    }
    __syncthreads();
    float* out = partial + (size_t)blockIdx.x * (C * 17);
    for (int o = threadIdx.x; o < C * 17; o += 256) {
        int c, t;
        if (o < C * 16) { c = o >> 4; t = o & 15; } else { c = o - C * 16; t = 16; }
        const int g = c >> 2, j = (c & 3) * 17 + t;
        out[o] = ((sh[0][g][j] + sh[1][g][j]) + sh[2][g][j]) + sh[3][g][j];
    }
}
void launch_conv1_wgrad(int dt, const void* dv, const float* x0, int n0, const float* x1, float* dW, float* db, float* partial,
                        int B, int S, int C, hipStream_t s) {
    const int nstrips = B * (S / 2 / 8);
    const int nch = nstrips < 1024 ? nstrips : 1024;
    // (all 16 loads of a strip issued up front instead of four at a time measured slower: 16.1 vs 13.1 us)
    SIGGAN_DT_SWITCH(dt, T, hipLaunchKernelGGL(k_conv1_wgrad<T>, dim3(nch), dim3(256), 0, s, (const T*)dv, x0, n0, x1, partial, S, nstrips));
    hipLaunchKernelGGL(k_rows_sum, dim3(cdiv(C * 17, 64)), dim3(1024), 0, s, partial, nch, C * 17, dW, C * 16, db);
}

// d(image) from dv, times tanh' -> d(pre-tanh).  A thread owns 4 channels of a column of 2x2 image
// blocks: block (a, b) = image pixels (2a..2a+1, 2b..2b+1) needs dv rows a-1..a+1, cols b-1..b+1 and
// every tap exactly once (ih = 2a: kh 1 -> oh a, kh 3 -> oh a-1; ih = 2a+1: kh 0 -> oh a+1, kh 2 -> oh a).
// 16 channel lanes x 16 columns per block, RA block rows per thread (sliding 3-row window).
template <class T, int RA>
__global__ __launch_bounds__(256) void k_conv1_dgrad_tanh(const T* __restrict__ dv, const float* __restrict__ Wt,
                                                          const float* __restrict__ img, float* __restrict__ dpre, int S) {
    constexpr int C = 64;
    const int Ho = S >> 1, nbb = Ho >> 4, nba = Ho / RA;
    const int q = threadIdx.x & 15, bl = threadIdx.x >> 4;
    int bid = blockIdx.x;
    const int b = (bid % nbb) * 16 + bl; bid /= nbb;
    const int a0 = (bid % nba) * RA, n = bid / nba;
    f4v w[16];                        // Wt = the weight as [tap][co] (k_prepare's PREP_TAPS copy): no LDS transpose, no barrier
#pragma unroll
    for (int t = 0; t < 16; ++t) w[t] = ldg4(Wt + t * C + q * 4);
    const T* base = dv + (size_t)n * Ho * Ho * C + q * 4;
    f4v g[RA + 2][3];
#pragma unroll
    for (int r = 0; r < RA + 2; ++r) {
        const int oh = a0 + r - 1, oc = clampi(oh, Ho - 1);
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int ow = b + k - 1, wc = clampi(ow, Ho - 1);
            const f4v t = ld4<T>(base + ((size_t)oc * Ho + wc) * C);
            g[r][k] = (oh == oc && ow == wc) ? t : f4v{0.f, 0.f, 0.f, 0.f};
        }
    }
#define DOT4(A, Wt, acc) acc = fmaf((A).x, (Wt).x, acc); acc = fmaf((A).y, (Wt).y, acc); acc = fmaf((A).z, (Wt).z, acc); acc = fmaf((A).w, (Wt).w, acc)
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        // g[r] = row a-1, g[r+1] = row a, g[r+2] = row a+1; columns 0,1,2 = b-1, b, b+1
        float o00 = 0.f, o01 = 0.f, o10 = 0.f, o11 = 0.f;
        // (2a, 2b): kh in {1: a, 3: a-1}, kw in {1: b, 3: b-1}
        DOT4(g[r + 1][1], w[1 * 4 + 1], o00); DOT4(g[r + 1][0], w[1 * 4 + 3], o00);
        DOT4(g[r][1], w[3 * 4 + 1], o00);     DOT4(g[r][0], w[3 * 4 + 3], o00);
        // (2a, 2b+1): kw in {0: b+1, 2: b}
        DOT4(g[r + 1][2], w[1 * 4 + 0], o01); DOT4(g[r + 1][1], w[1 * 4 + 2], o01);
        DOT4(g[r][2], w[3 * 4 + 0], o01);     DOT4(g[r][1], w[3 * 4 + 2], o01);
        // (2a+1, 2b): kh in {0: a+1, 2: a}
        DOT4(g[r + 2][1], w[0 * 4 + 1], o10); DOT4(g[r + 2][0], w[0 * 4 + 3], o10);
        DOT4(g[r + 1][1], w[2 * 4 + 1], o10); DOT4(g[r + 1][0], w[2 * 4 + 3], o10);
        // (2a+1, 2b+1)
        DOT4(g[r + 2][2], w[0 * 4 + 0], o11); DOT4(g[r + 2][1], w[0 * 4 + 2], o11);
        DOT4(g[r + 1][2], w[2 * 4 + 0], o11); DOT4(g[r + 1][1], w[2 * 4 + 2], o11);
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) {
            o00 += __shfl_xor(o00, o, 16); o01 += __shfl_xor(o01, o, 16);
            o10 += __shfl_xor(o10, o, 16); o11 += __shfl_xor(o11, o, 16);
        }
        if (q < 4) {
            const float v = q == 0 ? o00 : (q == 1 ? o01 : (q == 2 ? o10 : o11));
            const size_t pix = ((size_t)n * S + 2 * (a0 + r) + (q >> 1)) * S + 2 * b + (q & 1);
            const float t = img[pix];
            dpre[pix] = v * (1.0f - t * t);
        }
    }
#undef DOT4
}
void launch_conv1_dgrad_tanh(int dt, const void* dv, const float* Wt, const float* img, float* dpre, int B, int S, int C,
                             hipStream_t s) {
    (void)C;
    const int Ho = S / 2;
    SIGGAN_DT_SWITCH(dt, T, {
        // two block rows per thread: 118 registers, four waves per SIMD (four rows: 142 / three; 13.2 -> 12.0 us)
        hipLaunchKernelGGL((k_conv1_dgrad_tanh<T, 2>), dim3(B * (Ho / 2) * (Ho / 16)), dim3(256), 0, s, (const T*)dv, Wt, img, dpre, S);
    });
}

// =========================================================================================
// classifier, BCE
// =========================================================================================
__device__ __forceinline__ float block_sum(float v, float* sh) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) s += sh[k];
    return s;
}
template <class T>
__global__ __launch_bounds__(256) void k_cls_fwd(const T* __restrict__ act, const float* __restrict__ wcp,
                                                 const float* __restrict__ bc, float* __restrict__ logits, int F) {
    __shared__ float sh[4];
    const T* a = act + (size_t)blockIdx.x * F;
    const float4* w = (const float4*)wcp;
    float acc = 0.f;
    for (int i = threadIdx.x; i < F / 4; i += 256) {
        const float4 x = f4(ld4<T>(a + i * 4)), y = w[i];
        acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
    }
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) logits[blockIdx.x] = s + bc[0];
}
void launch_cls_fwd(int dt, const void* act, const float* wcp, const float* bc, float* logits, int B, int F, hipStream_t s) {
    SIGGAN_DT_SWITCH(dt, T, hipLaunchKernelGGL(k_cls_fwd<T>, dim3(B), dim3(256), 0, s, (const T*)act, wcp, bc, logits, F));
}
template <class T>
__global__ void k_cls_features(const T* __restrict__ act, float* __restrict__ feat, int64_t total, int C) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // i over [n][c][hw] (torch order)
    if (i >= total) return;
    const int hw = (int)(i % 16), c = (int)((i / 16) % C);
    const int64_t n = i / (16 * (int64_t)C);
    feat[i] = ld1<T>(act + (n * 16 + hw) * C + c);
}
void launch_cls_features(int dt, const void* act, float* feat, int B, int C, hipStream_t s) {
    const int64_t total = (int64_t)B * 16 * C;
    SIGGAN_DT_SWITCH(dt, T, hipLaunchKernelGGL(k_cls_features<T>, dim3(cdiv(total, 256)), dim3(256), 0, s, (const T*)act, feat, total, C));
}

__device__ __forceinline__ float bce_dlogit(float x, float y, float inv_count);
// nn.Sigmoid + nn.BCELoss(mean) per segment and its gradient w.r.t. the logit (torch formulas:
// log clamped at -100; grad_p = (p - y) / max((1 - p) * p, 1e-12) / count; dlogit = grad_p * p * (1 - p))
// The logit of row n: stored (k_cls_fwd), or -- when the last block's split-K epilogue left P partial dot products per image
// (GConvArgs::cls_part) -- their sum in order plus the bias.  k_bce and k_cls_bwd use this one expression: they agree bit for bit.
struct LogitSrc { const float* logits; const float* parts; int P; const float* bc; };
__device__ __forceinline__ float logit_of(const LogitSrc& q, int64_t n) {
    if (q.P == 0) return q.logits[n];
    float v[16];                                  // P <= 16: every load in flight before the first add (a run-time loop
#pragma unroll                                    // compiles to load, wait, add per partial: 8 dependent round trips)
    for (int p = 0; p < 16; ++p) v[p] = q.parts[n * q.P + (p < q.P ? p : 0)];
    float s = v[0];
#pragma unroll
    for (int p = 1; p < 16; ++p) s += p < q.P ? v[p] : 0.f;
    return s + q.bc[0];
}
struct BceArgs { LogitSrc src; float* logits_out; int B, n0; float y0, y1; float* probs; float* dlogit; float* metrics; int is_g; float gscale; };
__device__ __forceinline__ void bce_block(const LogitSrc src, float* __restrict__ logits_out, int B, int n0, float y0, float y1,
                                          float* __restrict__ probs, float* __restrict__ dlogit,
                                          float* __restrict__ metrics, int is_g, float gscale, float* sh) {
    float l0 = 0.f, l1 = 0.f, p0 = 0.f, p1 = 0.f, a0 = 0.f, a1 = 0.f;
    const float c0 = 1.0f / (float)(n0 > 0 ? n0 : 1), c1 = 1.0f / (float)(B - n0 > 0 ? B - n0 : 1);
    for (int n = threadIdx.x; n < B; n += 256) {
        const float x = logit_of(src, n);
        if (src.P && logits_out) logits_out[n] = x;
        const float p = 1.0f / (1.0f + expf(-x));
        const bool s0 = n < n0;
        const float y = s0 ? y0 : y1;
        const float lp = fmaxf(logf(p), -100.f), lq = fmaxf(logf(1.0f - p), -100.f);
        const float loss = -(y * lp + (1.0f - y) * lq);
        if (probs) probs[n] = p;
        if (dlogit) dlogit[n] = bce_dlogit(x, y, s0 ? c0 : c1) * gscale;
        if (s0) { l0 += loss; p0 += p; a0 += p > 0.5f ? 1.f : 0.f; }
        else    { l1 += loss; p1 += p; a1 += p < 0.5f ? 1.f : 0.f; }
    }
    // the six sums in one pass (block_sum's order per value: wave tree, then the waves in order; one barrier pair, not six)
    float v6[6] = {l0, l1, p0, p1, a0, a1};
#pragma unroll
    for (int q = 0; q < 6; ++q)
        for (int o = 32; o > 0; o >>= 1) v6[q] += __shfl_down(v6[q], o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) sh[q * 4 + (threadIdx.x >> 6)] = v6[q];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int q = 0; q < 6; ++q) { float t = 0.f; for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += sh[q * 4 + k]; v6[q] = t; }
        l0 = v6[0]; l1 = v6[1]; p0 = v6[2]; p1 = v6[3]; a0 = v6[4]; a1 = v6[5];
    }
    if (threadIdx.x == 0 && metrics) {
        if (is_g) {
            metrics[8] = l0 * c0;        // g_loss
            metrics[9] = p0 * c0;        // g_fake_mean
        } else {
            metrics[1] = l0 * c0; metrics[2] = l1 * c1; metrics[0] = l0 * c0 + l1 * c1;
            metrics[3] = p0 * c0; metrics[4] = p1 * c1; metrics[5] = a0 * c0; metrics[6] = a1 * c1;
        }
    }
}
__global__ __launch_bounds__(256) void k_bce(const BceArgs b) {
    __shared__ float sh[24];
    bce_block(b.src, b.logits_out, b.B, b.n0, b.y0, b.y1, b.probs, b.dlogit, b.metrics, b.is_g, b.gscale, sh);
}
void launch_bce(const float* logits, int B, int n0, float y0, float y1, float* probs, float* dlogit, float* metrics,
                int is_g_step, hipStream_t s, float gscale, const float* parts, int P, const float* bc) {
    hipLaunchKernelGGL(k_bce, dim3(1), dim3(256), 0, s,
                       BceArgs{LogitSrc{logits, parts, P, bc}, const_cast<float*>(logits), B, n0, y0, y1, probs, dlogit, metrics, is_g_step, gscale});
}

// d(logit) of sigmoid + BCE(mean) for row n, k_bce's own expression (so both kernels agree bit for bit)
__device__ __forceinline__ float bce_dlogit(float x, float y, float inv_count) {
    const float p = 1.0f / (1.0f + expf(-x));
    const float gp = (p - y) / fmaxf((1.0f - p) * p, 1e-12f) * inv_count;
    return gp * ((1.0f - p) * p);
}
// d(classifier input) * leaky'(a) * dropout.  Takes the logits, not d(logit): the backward chain then does not
// wait for k_bce (metrics + d(logit) for the classifier's weight gradient), which runs beside it.
// bce.logits != nullptr: one more block (the last) does k_bce's work -- the G step has the loss kernel and this one back to
// back on its only lane, and neither needs the other
template <class T>
__global__ __launch_bounds__(256) void k_cls_bwd(const float* __restrict__ logits, int B, int n0, float y0, float y1, const float* __restrict__ wcp,
                          const T* __restrict__ act, const float* __restrict__ noise, float slope, T* __restrict__ dv,
                          int64_t total, int C, float gscale, const BceArgs bce, int with_bce) {
    __shared__ float sh[24];
    if (with_bce && blockIdx.x == gridDim.x - 1) {
        bce_block(bce.src, bce.logits_out, bce.B, bce.n0, bce.y0, bce.y1, bce.probs, bce.dlogit, bce.metrics, bce.is_g, bce.gscale, sh);
        return;
    }
    // four consecutive channels per lane (C % 4 == 0: one image, one pixel): 16-byte loads and one store of four
    const int64_t i = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    const int F = 16 * C;
    if (i >= total) return;
    const float x = logit_of(bce.src, i / F);     // (every lane for itself: the partials are L1 hits beside the act / weight loads;
                                                  //  one lane + LDS broadcast + barrier doubled the kernel's time)
    const int j = (int)(i % F), c = j % C;
    const int64_t n = i / F;
    const bool s0 = n < n0;
    const float dl = bce_dlogit(x, s0 ? y0 : y1, 1.0f / (float)(s0 ? (n0 > 0 ? n0 : 1) : (B - n0 > 0 ? B - n0 : 1))) * gscale;
    const f32x4 w4 = *reinterpret_cast<const f32x4*>(wcp + j), a4 = ld4<T>(act + i);
    f32x4 g;
#pragma unroll
    for (int e = 0; e < 4; ++e) g[e] = dl * w4[e] * (a4[e] > 0.f ? 1.f : slope);
    if (noise) {
        const f32x4 nz = *reinterpret_cast<const f32x4*>(noise + n * C + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) g[e] *= nz[e];
    }
    st4<T>(dv + i, g);
}
void launch_cls_bwd(int dt, const float* logits, int n0, float y0, float y1, const float* wcp, const void* act, const float* noise,
                    float slope, void* dv, int B, int C, hipStream_t s, float gscale, float* bce_probs, float* bce_dlogit,
                    float* bce_metrics, int bce_is_g, bool with_bce, const float* parts, int P, const float* bc) {
    const int64_t total = (int64_t)B * 16 * C;
    const BceArgs b = BceArgs{LogitSrc{logits, parts, P, bc}, const_cast<float*>(logits), B, n0, y0, y1, bce_probs, bce_dlogit, bce_metrics,
                              bce_is_g, gscale};
    SIGGAN_DT_SWITCH(dt, T, hipLaunchKernelGGL(k_cls_bwd<T>, dim3(cdiv(total, 1024) + (with_bce ? 1 : 0)), dim3(256), 0, s, logits, B, n0, y0,
                                                y1, wcp, (const T*)act, noise, slope, (T*)dv, total, C, gscale, b, with_bce ? 1 : 0));
}
// dWc[f] = sum_n dlogit[n] * act[n][f'], dbc = sum_n dlogit[n]: 64 features x 4 row lanes per block (rows n = lane, lane + 4,
// ...), the four partial sums are added in lane order through LDS
template <class T>
__global__ __launch_bounds__(256) void k_cls_wgrad(const float* __restrict__ dlogit, const T* __restrict__ act, float* __restrict__ dWc,
                            float* __restrict__ dbc, int B, int C) {
    __shared__ float sh[4][64];
    const int F = 16 * C;
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    float acc = 0.f;
    if (j < F) {
#pragma unroll 8
        for (int n = rl; n < B; n += 4) acc = fmaf(dlogit[n], ld1<T>(act + (size_t)n * F + j), acc);
    } else if (j == F) {
        for (int n = rl; n < B; n += 4) acc += dlogit[n];
    }
    sh[rl][threadIdx.x & 63] = acc;
    __syncthreads();
    if (rl != 0 || j > F) return;
    const float t = ((sh[0][threadIdx.x] + sh[1][threadIdx.x]) + sh[2][threadIdx.x]) + sh[3][threadIdx.x];
    if (j < F) dWc[(j % C) * 16 + j / C] = t; else dbc[0] = t;
}
void launch_cls_wgrad(int dt, const float* dlogit, const void* act, float* dWc, float* dbc, int B, int C, hipStream_t s) {
    SIGGAN_DT_SWITCH(dt, T, hipLaunchKernelGGL(k_cls_wgrad<T>, dim3(cdiv(16 * C + 1, 64)), dim3(256), 0, s, dlogit, (const T*)act, dWc, dbc, B, C));
}

// test hook / operator entries: element-type conversion of a whole tensor
template <class T>
__global__ void k_to_f32(const T* __restrict__ src, float* __restrict__ dst, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = ld1<T>(src + i);
}
void launch_to_f32(int dt, const void* src, float* dst, int64_t n, hipStream_t s) {
    SIGGAN_DT_SWITCH(dt, T, hipLaunchKernelGGL(k_to_f32<T>, dim3(cdiv(n, 256)), dim3(256), 0, s, (const T*)src, dst, n));
}

// =========================================================================================
// input pipeline: per-sample nearest-neighbour rotation + scale of cached 8-bit images (byte work, HBM-bound)
// =========================================================================================
// One thread per output pixel.  The two resampling stages of the reference's transform chain
// (RandomRotation, then RandomAffine(scale), data_loader_signatures.py:176-193, both Pillow nearest-neighbour
// transforms with fill) are composed backwards: the scale stage's per-axis source tables (Pillow's
// ImagingScaleAffine, running double sums -- tabulated on the host) give the pixel of the rotated image, the
// rotation's 16.16 fixed-point map (Pillow's affine_fixed) gives the pixel of the cached image.  ToTensor +
// Normalize is a 256-entry table computed by torch itself.
//   prm[b] = {mode, a0, a1, a2, a3, a4, a5, flags}: mode 0 copy, 1 fixed point, 2 per-axis tables (tab rows 0, 1);
//   flags bit 0 horizontal flip (last), bit 1 scale stage present (tab rows 2, 3).  Table value -1: outside.
__global__ __launch_bounds__(256) void k_augment(const uint8_t* __restrict__ cache, const int32_t* __restrict__ index,
                                                 const int32_t* __restrict__ prm, const int16_t* __restrict__ tabs,
                                                 const float* __restrict__ lut, float* __restrict__ out, int S, int augment,
                                                 int fill, int64_t n_images) {
    const int b = blockIdx.y, p = blockIdx.x * 256 + threadIdx.x;
    if (p >= S * S) return;
    const int y = p / S, x = p - y * S;
    int64_t img = index[b];
    img = img < 0 ? 0 : (img >= n_images ? n_images - 1 : img);      // never read outside the cache
    const uint8_t* src = cache + (size_t)img * S * S;
    int v = fill;
    if (!augment) {
        v = src[p];
    } else {
        const int32_t* q = prm + b * 8;
        const int16_t* t = tabs + (size_t)b * 4 * S;
        const int flags = q[7];
        int xs = (flags & 1) ? S - 1 - x : x, ys = y;
        bool ok = true;
        if (flags & 2) { xs = t[2 * S + xs]; ys = t[3 * S + ys]; ok = xs >= 0 && ys >= 0; }
        if (ok) {
            const int mode = q[0];
            if (mode == 0) {
                v = src[ys * S + xs];
            } else if (mode == 1) {
                const int xin = (q[3] + ys * q[2] + xs * q[1]) >> 16, yin = (q[6] + ys * q[5] + xs * q[4]) >> 16;
                if (xin >= 0 && xin < S && yin >= 0 && yin < S) v = src[yin * S + xin];
            } else {
                const int xin = t[xs], yin = t[S + ys];
                if (xin >= 0 && yin >= 0) v = src[yin * S + xin];
            }
        }
    }
    out[(size_t)b * S * S + p] = lut[v & 255];
}
void launch_augment(const uint8_t* cache, int64_t n_images, const int32_t* index, const int32_t* prm, const int16_t* tabs,
                    const float* lut, float* out, int B, int S, int augment, int fill, hipStream_t s) {
    hipLaunchKernelGGL(k_augment, dim3(cdiv((int64_t)S * S, 256), B), dim3(256), 0, s, cache, index, prm, tabs, lut, out, S,
                       augment, fill, n_images);
}

// =========================================================================================
// clip + Adam
// =========================================================================================
__global__ __launch_bounds__(256) void k_sumsq(const float* __restrict__ g, int64_t n, float* __restrict__ partial) {
    __shared__ float sh[4];
    float acc = 0.f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) acc = fmaf(g[i], g[i], acc);
    const float s = block_sum(acc, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}
// Sum of the nb (<= 512) block partials of k_sumsq by ONE wave in a fixed order (lane l adds partial[l], partial[l + 64], ...,
// then a butterfly): the consumers of the norm do this themselves -- k_adam_prepare, or wave 0 of every k_adam<FUSED> block --
// so no finalize kernel sits between k_sumsq and the optimiser, and both paths form bit-identical sums.
__device__ __forceinline__ float sum_partials_wave(const float* __restrict__ partial, int nb, int lane) {
    float acc = 0.f;
    for (int i = lane; i < nb; i += 64) acc += partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    return acc;
}
static constexpr int SUMSQ_BLOCKS = 512;
void launch_grad_sumsq(const float* g, int64_t n, DevState* st, float* partial, hipStream_t s) {
    (void)st;
    hipLaunchKernelGGL(k_sumsq, dim3(SUMSQ_BLOCKS), dim3(256), 0, s, g, n, partial);
}

__global__ void k_adam_prepare(DevState* st, float* __restrict__ steps, int ntensors, double lr, double beta1, double beta2,
                               float grad_scale, float clip_max_norm, float* __restrict__ metric_norm, int check_finite,
                               float* __restrict__ metric_skipped, const float* __restrict__ sumsq_partial) {
    if (sumsq_partial) {                                               // (one wave: blockDim.x == 64)
        const float ssq = sum_partials_wave(sumsq_partial, SUMSQ_BLOCKS, threadIdx.x);
        if (threadIdx.x == 0) st->sumsq = ssq;
        __syncthreads();
    }
    // fp16 chains carry a static gradient scale: an activation gradient that overflowed arrives here as inf / NaN in the sum
    // of squares -- the update is skipped (parameters, moments, step counts untouched) instead of poisoning the fp32 masters
    const bool skip = check_finite && !isfinite(st->sumsq);
    // torch.optim.Adam: step += 1; bias corrections as Python doubles (1 - beta**step)
    const float t = steps[0] + 1.0f;
    __syncthreads();
    if (!skip) for (int i = threadIdx.x; i < ntensors; i += blockDim.x) steps[i] = t;
    if (threadIdx.x != 0) return;
    st->rng_ctr += 1;                       // every optimiser update starts a new RNG epoch (z, dropout tables)
    st->skip = skip ? 1 : 0;
    if (metric_skipped) *metric_skipped = skip ? 1.0f : 0.0f;
    const double bc1 = 1.0 - pow_step(beta1, (double)t);
    const double bc2 = 1.0 - pow_step(beta2, (double)t);
    st->step_size = (float)(lr / bc1);
    st->bc2_sqrt = (float)sqrt(bc2);
    float mul = grad_scale;
    if (clip_max_norm > 0.f) {
        const float norm = sqrtf(st->sumsq) * grad_scale;      // nn.utils.clip_grad_norm_ (norm_type 2)
        const float coef = fminf(clip_max_norm / (norm + 1e-6f), 1.0f);
        mul = grad_scale * coef;
        st->grad_norm = norm;
        if (metric_norm) *metric_norm = norm;
    }
    st->grad_mul = mul;
}
void launch_adam_prepare(DevState* st, float* steps, int ntensors, double lr, double beta1, double beta2, float grad_scale,
                         float clip_max_norm, float* metric_norm, hipStream_t s, int check_finite, float* metric_skipped,
                         const float* sumsq_partial) {
    hipLaunchKernelGGL(k_adam_prepare, dim3(1), dim3(64), 0, s, st, steps, ntensors, lr, beta1, beta2, grad_scale,
                       clip_max_norm, metric_norm, check_finite, metric_skipped, sumsq_partial);
}

struct AdamHost {          // scalars of a fused update, formed on the host (FUSED: no k_adam_prepare ran)
    float step_size, bc2_sqrt, grad_scale, clip_max_norm, t;
    int ntensors;
    float* steps;
    float* metric_norm;
    float* metric_skipped;        // the step's "update skipped" flag: a fused update never skips, it writes 0 (include/siggan.h)
    const float* sumsq_partial;   // k_sumsq's block partials (clip): every block's wave 0 adds them itself
};
// multiplier / step size / bias correction of a fused update (every workgroup of the launch; contains a barrier when clipping),
// and what k_adam_prepare does besides the scalars (workgroup 0)
__device__ __forceinline__ void adam_fused_scalars(const AdamHost& h, DevState* st, AdamK& k) {
    k.mul = h.grad_scale; k.ss = -h.step_size; k.bc2 = h.bc2_sqrt;
    float norm = 0.f;
    if (h.clip_max_norm > 0.f) {                                   // nn.utils.clip_grad_norm_ (norm_type 2), k_adam_prepare's expression
        __shared__ float s_ssq;
        if (threadIdx.x < 64) {
            const float ssq = sum_partials_wave(h.sumsq_partial, SUMSQ_BLOCKS, threadIdx.x);
            if (threadIdx.x == 0) s_ssq = ssq;
        }
        __syncthreads();
        norm = sqrtf(s_ssq) * h.grad_scale;
        k.mul = h.grad_scale * fminf(h.clip_max_norm / (norm + 1e-6f), 1.0f);
    }
    if (blockIdx.x == 0) {
        for (int i = threadIdx.x; i < h.ntensors; i += 256) h.steps[i] = h.t;
        if (threadIdx.x == 0) {
            st->rng_ctr += 1;                                      // every optimiser update starts a new RNG epoch
            st->skip = 0;
            if (h.metric_skipped) *h.metric_skipped = 0.0f;
            if (h.clip_max_norm > 0.f) { st->grad_norm = norm; if (h.metric_norm) *h.metric_norm = norm; }
        }
    }
}
template <bool FUSED>
__global__ __launch_bounds__(256) void k_adam(float4* __restrict__ p, float4* __restrict__ g, float4* __restrict__ m,
                                              float4* __restrict__ v, int64_t n4, float* __restrict__ pt,
                                              float* __restrict__ gt, float* __restrict__ mt, float* __restrict__ vt,
                                              int tail, DevState* __restrict__ st, float w1, float beta2,
                                              float w2, float eps, int wb, const AdamHost h) {
    AdamK k; k.w1 = w1; k.beta2 = beta2; k.w2 = w2; k.eps = eps;
    if (FUSED) {
        adam_fused_scalars(h, st, k);
    } else {
        if (st->skip) return;               // (uniform: k_adam_prepare found a non-finite fp16 gradient)
        k.mul = st->grad_mul; k.ss = -st->step_size; k.bc2 = st->bc2_sqrt;
    }
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) {
        float4 pp = p[i], gg = g[i], mm = m[i], vv = v[i];
        adam_upd(k, pp.x, gg.x, mm.x, vv.x); adam_upd(k, pp.y, gg.y, mm.y, vv.y);
        adam_upd(k, pp.z, gg.z, mm.z, vv.z); adam_upd(k, pp.w, gg.w, mm.w, vv.w);
        p[i] = pp; m[i] = mm; v[i] = vv;
        if (wb) g[i] = gg;
    } else if (i - n4 < tail) {
        const int e = (int)(i - n4);
        float pp = pt[e], gg = gt[e], mm = mt[e], vv = vt[e];
        adam_upd(k, pp, gg, mm, vv);
        pt[e] = pp; mt[e] = mm; vt[e] = vv;
        if (wb) gt[e] = gg;
    }
}
static AdamHost adam_host(double t, double lr, double beta1, double beta2, float grad_scale, float clip_max_norm, float* steps,
                          int ntensors, float* metric_norm, float* metric_skipped, const float* sumsq_partial) {
    AdamHost h;
    h.step_size = (float)(lr / (1.0 - pow_step(beta1, t)));            // torch.optim.Adam: lr / (1 - beta1**step), in double
    h.bc2_sqrt = (float)sqrt(1.0 - pow_step(beta2, t));
    h.grad_scale = grad_scale; h.clip_max_norm = clip_max_norm; h.t = (float)t; h.ntensors = ntensors; h.steps = steps;
    h.metric_norm = metric_norm; h.metric_skipped = metric_skipped; h.sumsq_partial = sumsq_partial;
    return h;
}
void launch_adam_fused(float* p, float* g, float* m, float* v, int64_t n, DevState* st, float* steps, int ntensors, double t,
                       double lr, double beta1, double beta2, double eps, float grad_scale, float clip_max_norm,
                       float* metric_norm, const float* sumsq_partial, hipStream_t s, float* metric_skipped) {
    const int64_t n4 = n / 4;
    const int tail = (int)(n - n4 * 4);
    const AdamHost h = adam_host(t, lr, beta1, beta2, grad_scale, clip_max_norm, steps, ntensors, metric_norm, metric_skipped, sumsq_partial);
    const int wb = (clip_max_norm > 0.f || grad_scale != 1.0f) ? 1 : 0;
    hipLaunchKernelGGL(k_adam<true>, dim3(cdiv(n4 + tail, 256)), dim3(256), 0, s, (float4*)p, (float4*)g, (float4*)m, (float4*)v,
                       n4, p + n4 * 4, g + n4 * 4, m + n4 * 4, v + n4 * 4, tail, st, (float)(1.0 - beta1), (float)beta2,
                       (float)(1.0 - beta2), (float)eps, wb, h);
}
void launch_adam(float* p, float* g, float* m, float* v, int64_t n, const DevState* st, double beta1, double beta2,
                 double eps, int write_back_grad, hipStream_t s) {
    const int64_t n4 = n / 4;
    const int tail = (int)(n - n4 * 4);
    hipLaunchKernelGGL(k_adam<false>, dim3(cdiv(n4 + tail, 256)), dim3(256), 0, s, (float4*)p, (float4*)g, (float4*)m, (float4*)v,
                       n4, p + n4 * 4, g + n4 * 4, m + n4 * 4, v + n4 * 4, tail, const_cast<DevState*>(st), (float)(1.0 - beta1),
                       (float)beta2, (float)(1.0 - beta2), (float)eps, write_back_grad, AdamHost{});
}


// =========================================================================================
// k_adam_pack: the optimiser update and everything launch_prepare derives from the arena, one launch
// =========================================================================================
static long long ap_units(const ApJob& j) {
    switch (j.type) {
        case AP_FLAT: return (j.n + 1023) / 1024;
        case AP_CONV: return (long long)(j.A / 16) * (j.Bc / 16);
        case AP_TAPS: return 1;
        default: return (j.A + 255) / 256;       // T16 / BN: 256 rows / channels per workgroup
    }
}
void ap_add(ApTable& t, const ApJob& j) {
    if (t.njobs >= ApTable::MAXJ) { t.overflow = 1; return; }
    if (t.njobs == 0) t.prefix[0] = 0;
    t.job[t.njobs] = j;
    t.prefix[t.njobs + 1] = t.prefix[t.njobs] + ap_units(j);
    ++t.njobs;
}
struct ApArena { float *p, *g, *m, *v; int wb; };
__device__ __forceinline__ void ap_put(float* dst, int dt, size_t idx, float v) {
    if (dt == DT_F32) dst[idx] = v;
    else if (dt == DT_BF16) reinterpret_cast<bf16_t*>(dst)[idx] = (bf16_t)v;
    else reinterpret_cast<f16_t*>(dst)[idx] = (f16_t)v;
}
// one element / one aligned group of four: update in place, return the new parameter(s)
__device__ __forceinline__ float ap_upd1(const ApArena& a, const AdamK& k, size_t e) {
    float pp = a.p[e], gg = a.g[e], mm = a.m[e], vv = a.v[e];
    adam_upd(k, pp, gg, mm, vv);
    a.p[e] = pp; a.m[e] = mm; a.v[e] = vv;
    if (a.wb) a.g[e] = gg;
    return pp;
}
__device__ __forceinline__ float4 ap_upd4(const ApArena& a, const AdamK& k, size_t e) {
    float4 pp = *reinterpret_cast<const float4*>(a.p + e), gg = *reinterpret_cast<const float4*>(a.g + e);
    float4 mm = *reinterpret_cast<const float4*>(a.m + e), vv = *reinterpret_cast<const float4*>(a.v + e);
    adam_upd(k, pp.x, gg.x, mm.x, vv.x); adam_upd(k, pp.y, gg.y, mm.y, vv.y);
    adam_upd(k, pp.z, gg.z, mm.z, vv.z); adam_upd(k, pp.w, gg.w, mm.w, vv.w);
    *reinterpret_cast<float4*>(a.p + e) = pp; *reinterpret_cast<float4*>(a.m + e) = mm; *reinterpret_cast<float4*>(a.v + e) = vv;
    if (a.wb) *reinterpret_cast<float4*>(a.g + e) = gg;
    return pp;
}
static constexpr int AP_TS = 273;          // LDS tile of a CONV unit: [a][b][tap] at a * 273 + b * 17 + tap (odd strides: conflict-free both ways)
__device__ __forceinline__ void ap_unit(const ApTable& t, const ApArena& a, const AdamK& k, float bn_eps, unsigned bid, unsigned* counter) {
    extern __shared__ float tile[];
    int j = 0;
    while ((long long)bid >= t.prefix[j + 1]) ++j;
    const ApJob& q = t.job[j];
    const int u = (int)(bid - t.prefix[j]);
    const int tid = threadIdx.x;
    if (q.type == AP_CONV) {
        const int ntb = q.Bc >> 4, a0 = (u / ntb) * 16, b0 = (u % ntb) * 16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f = tid + 256 * r, al = f >> 6, rr = f & 63;        // float4 f of the tile: row a0 + al, 64 float4 of 16 b x 16 taps
            const float4 w = ap_upd4(a, k, (size_t)q.off + ((size_t)(a0 + al) * q.Bc + b0) * 16 + rr * 4);
            float* d = tile + al * AP_TS + (rr >> 2) * 17 + (rr & 3) * 4;
            d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
        }
        __syncthreads();
        // unit = dim 0 (PREP_PACK_DOWN, I = Bc): dst[a][tap * Bc + b]
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
            const int e = tid + 256 * r, bl = e & 15, tap = (e >> 4) & 15, al = e >> 8;
            ap_put(q.dst, q.dt, ((size_t)(a0 + al) * 16 + tap) * q.Bc + b0 + bl, tile[al * AP_TS + bl * 17 + tap]);
        }
        // unit = dim 1 (PREP_PACK_UP, I = A, O = Bc): dst[cls][b][t * A + a], the four taps of each output-parity class
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
            const int e = tid + 256 * r, al = e & 15, ct = (e >> 4) & 15, bl = e >> 8;
            const int cls = ct >> 2, tt = ct & 3;
            const int kh = 1 - (cls >> 1) + 2 * (tt >> 1), kw = 1 - (cls & 1) + 2 * (tt & 1);
            ap_put(q.dst2, q.dt, (((size_t)cls * q.Bc + b0 + bl) * 4 + tt) * q.A + a0 + al, tile[al * AP_TS + bl * 17 + kh * 4 + kw]);
        }
    } else if (q.type == AP_FLAT) {
        const long long e = (long long)u * 1024 + tid * 4;
        if (e + 4 <= q.n && ((q.off + e) & 3) == 0) ap_upd4(a, k, (size_t)(q.off + e));
        else for (long long i = e; i < e + 4 && i < q.n; ++i) ap_upd1(a, k, (size_t)(q.off + i));
    } else if (q.type == AP_T16) {                // [A][16] -> dst[16][A] (the classifier weight in the last block's NHWC order)
        const int r0 = u * 256, nr = min(256, q.A - r0);
        for (int f = tid; f < nr * 4; f += 256) {
            const float4 w = ap_upd4(a, k, (size_t)q.off + (size_t)r0 * 16 + f * 4);
            float* d = tile + (f >> 2) * 17 + (f & 3) * 4;
            d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
        }
        __syncthreads();
        for (int e = tid; e < nr * 16; e += 256) { const int tap = e / nr, rl = e - tap * nr; q.dst[(size_t)tap * q.A + r0 + rl] = tile[rl * 17 + tap]; }
    } else if (q.type == AP_TAPS) {               // one-channel convs: w[c][tap] -> dst[tap][c]; <= 1024 weights (+ <= 256 bias values)
        float pp[4], gg[4], mm[4], vv[4], pb = 0.f, gb = 0.f, mb = 0.f, vb = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = tid + 256 * r;
            if (e < q.n) { pp[r] = a.p[q.off + e]; gg[r] = a.g[q.off + e]; mm[r] = a.m[q.off + e]; vv[r] = a.v[q.off + e]; adam_upd(k, pp[r], gg[r], mm[r], vv[r]); }
        }
        if (tid < q.n2) { pb = a.p[q.off2 + tid]; gb = a.g[q.off2 + tid]; mb = a.m[q.off2 + tid]; vb = a.v[q.off2 + tid]; adam_upd(k, pb, gb, mb, vb); }
        if (q.wait > 0) {
            // the riders form these values from the ranges as they are NOW: nothing is written until all of them have read
            // (bounded: a rider never waits, so the count is reached as soon as the last one has been dispatched)
            if (tid == 0) {
                unsigned it = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)q.wait && ++it < (1u << 20))
                    __builtin_amdgcn_s_sleep(16);
                __hip_atomic_store(counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __syncthreads();
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int e = tid + 256 * r;
            if (e < q.n) {
                a.p[q.off + e] = pp[r]; a.m[q.off + e] = mm[r]; a.v[q.off + e] = vv[r];
                if (a.wb) a.g[q.off + e] = gg[r];
                q.dst[(size_t)(e % q.Bc) * q.A + e / q.Bc] = pp[r];
            }
        }
        if (tid < q.n2) {
            a.p[q.off2 + tid] = pb; a.m[q.off2 + tid] = mb; a.v[q.off2 + tid] = vb;
            if (a.wb) a.g[q.off2 + tid] = gb;
        }
    } else {                                      // BN: gamma / beta of 256 channels, then [scale | shift | mean | rstd] (prepare_unit's expressions)
        const int C = q.A, c = u * 256 + tid;
        if (c < C) {
            const int tix = perm16(c, q.Bc);
            const float gam = ap_upd1(a, k, (size_t)q.off + tix), bet = ap_upd1(a, k, (size_t)q.off2 + tix);
            const float rstd = 1.0f / sqrtf(q.rvar[tix] + bn_eps);
            const float sc = gam * rstd;
            q.dst[c] = sc; q.dst[C + c] = bet - q.rmean[tix] * sc; q.dst[2 * C + c] = q.rmean[tix]; q.dst[3 * C + c] = rstd;
        }
    }
}
struct ApRideDev { const float* x; void* out; int B, S; float slope; long long w_off, b_off; };
template <class T, bool RIDE>
__global__ __launch_bounds__(256) void k_adam_pack(const ApTable t, const ApArena a, DevState* __restrict__ st, float w1, float beta2,
                                                   float w2, float eps, const AdamHost h, float bn_eps, unsigned nprep,
                                                   const ApRideDev r, unsigned* counter) {
    AdamK k; k.w1 = w1; k.beta2 = beta2; k.w2 = w2; k.eps = eps;
    adam_fused_scalars(h, st, k);
    if (blockIdx.x < nprep) { ap_unit(t, a, k, bn_eps, blockIdx.x, counter); return; }
    if (RIDE) {
        Conv1Ride rd;
        rd.pw = a.p + r.w_off; rd.gw = a.g + r.w_off; rd.mw = a.m + r.w_off; rd.vw = a.v + r.w_off;
        rd.pb = a.p + r.b_off; rd.gb = a.g + r.b_off; rd.mb = a.m + r.b_off; rd.vb = a.v + r.b_off;
        rd.counter = counter; rd.k = k;
        conv1_fwd_block<T, true>(r.x, r.B, r.x, nullptr, nullptr, nullptr, r.slope, (T*)r.out, r.S, blockIdx.x - nprep, &rd);
    }
}
bool launch_adam_pack(const ApTable& t, float* p, float* g, float* m, float* v, DevState* st, float* steps, int ntensors,
                      double tstep, double lr, double beta1, double beta2, double eps, float grad_scale, float clip_max_norm,
                      float* metric_norm, const float* sumsq_partial, float* metric_skipped, float bn_eps,
                      const ApRide* ride, hipStream_t s) {
    if (t.overflow || t.njobs == 0) return false;
    const AdamHost h = adam_host(tstep, lr, beta1, beta2, grad_scale, clip_max_norm, steps, ntensors, metric_norm, metric_skipped, sumsq_partial);
    ApArena a; a.p = p; a.g = g; a.m = m; a.v = v; a.wb = (clip_max_norm > 0.f || grad_scale != 1.0f) ? 1 : 0;
    const unsigned nprep = (unsigned)t.prefix[t.njobs];
    const size_t lds = (size_t)(16 * AP_TS) * sizeof(float);        // CONV tile 16 x 273 (T16: 256 x 17 fits)
    const float fw1 = (float)(1.0 - beta1), fb2 = (float)beta2, fw2 = (float)(1.0 - beta2), fe = (float)eps;
    if (!ride) {
        hipLaunchKernelGGL((k_adam_pack<float, false>), dim3(nprep), dim3(256), lds, s, t, a, st, fw1, fb2, fw2, fe, h, bn_eps, nprep,
                           ApRideDev{}, (unsigned*)nullptr);
        return true;
    }
    ApRideDev r; r.x = ride->x; r.out = ride->out; r.B = ride->B; r.S = ride->S; r.slope = ride->slope; r.w_off = ride->w_off; r.b_off = ride->b_off;
    const unsigned nride = (unsigned)(ride->B * (ride->S / 4));
    SIGGAN_DT_SWITCH(ride->dt, T, hipLaunchKernelGGL((k_adam_pack<T, true>), dim3(nprep + nride), dim3(256), lds, s, t, a, st, fw1, fb2, fw2, fe,
                                                      h, bn_eps, nprep, r, ride->counter));
    return true;
}

}  // namespace siggan
