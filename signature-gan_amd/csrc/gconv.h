// gconv.h -- argument blocks of the MFMA implicit-GEMM kernels (4x4 stride-2 pad-1 family).
// All activations are NHWC fp32 inside the library; see DESIGN.md "data layout in HBM".
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace siggan {

enum Epilogue : int {
    EPI_RAW = 0,          // store the accumulator
    EPI_BIAS_LRELU_DROP,  // leaky(acc + bias[c]) * noise[n,c]         (Discriminator block forward)
    EPI_AFFINE_RELU,      // relu(acc * scale[c] + shift[c])           (Generator block, BN eval folded)
    EPI_LRELU_BWD,        // acc * leaky'(aref) * noise[n,c]           (Discriminator input-gradient)
};

// out[n, opix, co] = sum_{tap, ci} in[n, pix(tap), ci] * wp[cls][co][tap*Ci + ci]
struct GConvArgs {
    const float* in;      // [B][Hi][Wi][Ci]
    const float* wp;      // packed weights [ncls][Co][ntaps*Ci]
    float* out;           // [B][Ho][Wo][Co]
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
    const float* aref;    // [B][Ho][Wo][Co] stored activation (EPI_LRELU_BWD)
    float slope;
    // split-K scratch (optional): nsplit slabs of the whole output, summed by k_splitk_epilogue
    float* slab;
    int64_t slab_floats;
    size_t slab_stride;   // filled by launch_gconv
    const float* zeros;   // >= 16 bytes of zeros (source of out-of-image taps)
};

// slab[z][i][tap*Cl + l] = sum_{pix in split z} S[pix][i] * L[n, 2p-1+kh, 2q-1+kw][l]
struct WgradArgs {
    const float* S;       // [B][Hs][Ws][Cs]   (small spatial)
    const float* L;       // [B][2Hs][2Ws][Cl] (large spatial)
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
    struct Rec { int id; double flops; hipEvent_t e0, e1; };
    static constexpr int NID = 8;
    std::vector<Rec> recs;
    static const char* name(int id);
    void begin(int id, double flops, hipStream_t st);
    void end(hipStream_t st);
    void clear();
};
extern Prof* g_prof;

void launch_gconv(const GConvArgs& a, hipStream_t st);
// fills a.dw (through the slabs + k_wgrad_reduce when K is split); returns the number of K splits
// (slab must hold max_splits*Cs*(16*Cl + 1) floats)
int launch_wgrad(WgradArgs a, int max_splits, hipStream_t st);
// torch layout (O,I,4,4) -> down pack [O][tap*I + i];  torch (I,O,4,4) -> up pack [4][O][t*I + i]
void launch_pack_down(const float* w, float* wp, int O, int I, hipStream_t st);
void launch_pack_up(const float* w, float* wp, int I, int O, hipStream_t st);

}  // namespace siggan
