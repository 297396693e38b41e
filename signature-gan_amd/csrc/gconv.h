// gconv.h -- argument blocks of the MFMA implicit-GEMM kernels (4x4 stride-2 pad-1 family).
// All activations are NHWC inside the library, element type `dt` (fp32 by default; bf16 / f16 for the narrow variants,
// see act.h); see DESIGN.md "data layout in HBM".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

#include "act.h"

namespace siggan {

enum Epilogue : int {
    EPI_RAW = 0,          // store the accumulator
    EPI_BIAS_LRELU_DROP,  // leaky(acc + bias[c]) * noise[n,c]         (Discriminator block forward)
    EPI_AFFINE_RELU,      // relu(acc * scale[c] + shift[c])           (Generator block, BN eval folded)
    EPI_LRELU_BWD,        // acc * leaky'(aref) * noise[n,c]           (Discriminator input-gradient)
    EPI_BN_BWD_STATS,     // store the accumulator AND the BatchNorm-backward sums of the tensor it is the gradient of (Generator
                          // input-gradient): per workgroup one partial row of sum(dr) and sum(dr * xhat), dr = relu'(.) * acc with
                          // the mask re-derived from aref = the pre-BatchNorm tensor y (fma(y, scale, shift) > 0, k_bn_relu's own
                          // expression) and xhat = (y - mean) * rstd -- what k_colreduce<FBnBwd> computes in a pass of its own
};

// out[n, opix, co] = sum_{tap, ci} in[n, pix(tap), ci] * wp[cls][co][tap*Ci + ci]
struct GConvArgs {
    int dt;               // element type of in / wp / out / aref (DT_F32, DT_BF16, DT_F16)
    const void* in;       // [B][Hi][Wi][Ci]
    const void* wp;       // packed weights [ncls][Co][ntaps*Ci]
    void* out;            // [B][Ho][Wo][Co]
    int B, Hi, Wi, Ci, Co;
    int lgHr, lgWr;       // log2 of the per-image row grid (down: Ho,Wo; up: Hi,Wi)
    int Ho, Wo;
    int form;             // 0 = down (16 taps), 1 = up (4 parity classes x 4 taps)
    int M;                // B * Hr * Wr rows (per class)
    int epi;
    const float* bias;    // [Co]
    const float* noise;   // [B][Co] dropout multipliers (0 or 1/(1-p)); nullptr = none
    const float* scale;   // [Co]
    const float* shift;   // [Co]
    const void* aref;     // [B][Ho][Wo][Co] stored activation (EPI_LRELU_BWD) / pre-BatchNorm tensor (EPI_BN_BWD_STATS)
    const float* bnp;     // EPI_BN_BWD_STATS: [scale | shift | mean | rstd] of the output's BatchNorm, Co floats each
    float* stat0;         // EPI_BN_BWD_STATS: partial rows [nrows][Co] of sum(dr); launch_gconv returns nrows and puts the
    float* stat1;         //   rows of sum(dr * xhat) right behind them (stat1 = stat0 + nrows * Co, filled by launch_gconv)
    int64_t stat_cap;     //   floats available at stat0: a launch whose 2 * nrows * Co rows would not fit stores the raw accumulator
                          //   instead (EPI_RAW) and returns 0 -- the caller then runs the generic reduction pass (k_colreduce)
    const float* cls_w;   // optional (last Discriminator block): the classifier's weight in the output's own (NHWC) order, Ho*Wo*Co
    float* cls_part;      //   floats.  When the launch ends in k_splitk_epilogue, each of its workgroups also writes the partial
                          //   dot product of its 1024 stored output values with cls_w to cls_part[workgroup] (P = Ho*Wo*Co / 1024
                          //   consecutive partials per image); launch_gconv returns P, or 0 when the epilogue ran elsewhere (the
                          //   caller then runs k_cls_fwd)
    float slope;
    // split-K scratch (optional): nsplit fp32 slabs of the whole output, summed by k_splitk_epilogue
    float* slab;
    int64_t slab_floats;
    size_t slab_stride;   // filled by launch_gconv
    const float* zeros;   // >= 16 bytes of zeros (source of out-of-image taps)
};

// EPI_BN_BWD_STATS, one float4 of the output: v = the accumulator as it is stored (rounded to T), y = the pre-BatchNorm values
// at the same place; the arithmetic is k_colreduce<FBnBwd>'s, element for element.
struct BnBwdParams { f32x4 sc, sf, mu, rs; };
__device__ __forceinline__ BnBwdParams bn_bwd_params(const float* __restrict__ bnp, int Co, int co) {
    BnBwdParams q;
    q.sc = *reinterpret_cast<const f32x4*>(bnp + co); q.sf = *reinterpret_cast<const f32x4*>(bnp + Co + co);
    q.mu = *reinterpret_cast<const f32x4*>(bnp + 2 * Co + co); q.rs = *reinterpret_cast<const f32x4*>(bnp + 3 * Co + co);
    return q;
}
template <class T>
__device__ __forceinline__ void bn_bwd_stat_terms(const f32x4 v, const f32x4 y, const BnBwdParams& q, f32x4& s0, f32x4& s1) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float vq = (float)(T)v[e];
        const float d = fmaf(y[e], q.sc[e], q.sf[e]) > 0.f ? vq : 0.f;
        s0[e] += d;
        s1[e] = fmaf(d, (y[e] - q.mu[e]) * q.rs[e], s1[e]);
    }
}

// slab[z][i][tap*Cl + l] = sum_{pix in split z} S[pix][i] * L[n, 2p-1+kh, 2q-1+kw][l]
struct WgradArgs {
    int dt;               // element type of S and L; slab / dw / db are always fp32
    const void* S;        // [B][Hs][Ws][Cs]   (small spatial)
    const void* L;        // [B][2Hs][2Ws][Cl] (large spatial)
    float* slab;          // [nsplit][Cs][16*Cl]
    float* dw;            // result, torch layout: dw[(i*Cl + l)*16 + tap]
    float* db;            // optional: db[i] = sum_pix S[pix][i] (the bias gradient when S is d(pre-activation)); partials
                          // live behind the weight slabs, slab[nsplit*Cs*16*Cl + z*Cs + i]
    int B, Cs, Cl;
    int lgHs, lgWs, lgCl;
    int K;                // B*Hs*Ws pixels
    int kchunk;           // pixels per split (multiple of 32)
    const float* zeros;   // >= 16 bytes of zeros
};

// Optional per-kernel timing (bench.py's roofline leg): when a profiler is installed every MFMA
// launch is bracketed by HIP events on its own stream and tagged with its algorithmic FLOPs.
struct Prof {
    struct Rec { int id; double flops, bytes; hipEvent_t e0, e1; };   // algorithmic FLOPs and HBM bytes (operands read once + result written once)
    static constexpr int NID = 8;
    std::vector<Rec> recs;
    static const char* name(int id);
    void begin(int id, double flops, double bytes, hipStream_t st);
    void clear();
};
extern Prof* g_prof;

// returns the number of partial rows written to stat0 / stat1 (EPI_BN_BWD_STATS), 0 otherwise
int launch_gconv(const GConvArgs& a, hipStream_t st);
// 16-bit operand kernels (gconv16.hip); cfg: 0 = 128x128, 2 = 64x64, 3 = 128x32 tiles; e0 / e1: optional timing events
void launch_gconv16(int cfg, const GConvArgs& a, dim3 grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, int kq = 1);
void launch_wgrad16(bool small, const WgradArgs& a, dim3 grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1);
// fills a.dw (through the slabs + k_wgrad_reduce when K is split); returns the number of K splits
// (slab must hold max_splits*Cs*(16*Cl + 1) floats)
int launch_wgrad(WgradArgs a, int max_splits, hipStream_t st);

}  // namespace siggan
