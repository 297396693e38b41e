// fc.hip -- Generator.fc = Linear(latent, F) + BatchNorm1d(F) + ReLU (generator_vanilla_gan.py:124-128) on the fp32
// matrix cores, forward and backward each as ONE launch (gfx950).
//
// The GEMM is tiny (B x F x latent = 64 x 4096 x 100: 52 MFLOP, 1.6 MB of weights) and sits at the head of every Generator
// pass and at the tail of its backward pass, so what matters is the number of dependent launches, not the FLOP rate:
//   k_fc_fwd_mfma : y = z W^T + b on v_mfma_f32_32x32x2_f32, and -- because a workgroup owns ALL batch rows of its 32
//                   features -- BatchNorm1d's batch statistics, the running-statistics update, the scale/shift table the
//                   backward pass reuses, and ReLU in the same kernel (training: 4 launches -> 1; eval: the folded table).
//   k_fc_bwd_mfma : ReLU mask + BatchNorm1d backward (two batch sums per feature) + dW = dy^T z on the matrix cores + db,
//                   again per 32-feature workgroup (4 launches -> 1).
// Operands go straight from global memory / L2 to registers (no LDS staging: every operand byte is used by exactly one
// workgroup): a lane's float4 covers four consecutive MFMA k-steps in the permuted pairing k = 8c + 4*(lane>>5) + t that
// gconv.hip uses, identical for both operands.  The four waves of a workgroup split K (forward) or the weight columns
// (backward); forward partial sums meet in LDS.  Features are produced in the NHWC order f' = hw*C0 + c the next layer
// reads (weight row f = c*16 + hw), as before.
#include "ops.h"
#include "rng.h"

namespace siggan {

// the four standard normals of RNG group g of stream sid: exactly what k_randn writes to elements [4g, 4g + 4)
__device__ __forceinline__ f32x4 randn4(const DevState* st, uint64_t g, uint32_t sid) {
    return normal4(draw_raw(st->seed, st->rng_ctr, g, sid));
}

struct FcFwdArgs {
    const float* z;        // [B][K] or nullptr: drawn here (K % 4 == 0), also written to z_out
    const float* W;        // [F][K] torch layout, f = c*16 + hw
    const float* bias;     // [F] torch order
    void* y;               // training: pre-BN output [B][F] (NHWC feature order), element type T
    void* a;               // relu(BN(y)) [B][F], element type T
    const float* gamma; const float* beta;       // training: BN affine (torch order)
    float* rmean; float* rvar; int64_t* batches; // training: running statistics (torch order), num_batches_tracked
    float* bn;             // training: out [6*F]: scale | shift | mean | rstd | (2 slots the backward fills)
    const float* bne;      // eval: folded [scale | shift] table (feature order f')
    float* z_out;
    const DevState* st; uint32_t sid;
    int B, K, C0;
    float momentum, eps;
};

// one workgroup = 32 features x all B rows (B <= 32*MT); wave w takes the k-chunks c = w, w+4, ... of 8
template <class T, int MT>
__global__ __launch_bounds__(256) void k_fc_fwd_mfma(const FcFwdArgs p) {
    __shared__ __attribute__((aligned(16))) float red[3][MT][16][64];      // partial accumulators of waves 1..3
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int F = p.C0 * 16, K = p.K, B = p.B;
    const int fp0 = blockIdx.x * 32;                            // features f' = fp0 .. fp0+31
    const int fpl = fp0 + li, frow = (fpl % p.C0) * 16 + fpl / p.C0;       // this lane's B-operand row of W
    const float* const wrow = p.W + (size_t)frow * K;
    const int nch = (K + 7) >> 3;
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    // a wave's chunks are c = wave + 4g: the operands of up to four of them are loaded before their MFMAs start, so the
    // (uncoalesced, L2-latency-bound) loads of a group are in flight together
    constexpr int G = MT <= 2 ? 4 : (MT == 4 ? 2 : 1);           // chunks in flight per wave (register budget)
    for (int c0 = wave; c0 < nch; c0 += 4 * G) {
        f32x4 fb[G], fa[G][MT];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int k = 8 * (c0 + 4 * g) + 4 * lh;
            fb[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (k + 3 < K) fb[g] = *reinterpret_cast<const f32x4*>(wrow + k);
            else { for (int t = 0; t < 4; ++t) if (k + t < K) fb[g][t] = wrow[k + t]; }
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const int n = 32 * m + li;
                fa[g][m] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (n < B && k < K) {
                    if (p.z) {
                        if (k + 3 < K) fa[g][m] = *reinterpret_cast<const f32x4*>(p.z + (size_t)n * K + k);
                        else { for (int t = 0; t < 4; ++t) if (k + t < K) fa[g][m][t] = p.z[(size_t)n * K + k + t]; }
                    } else {
                        fa[g][m] = randn4(p.st, ((uint64_t)n * K + k) >> 2, p.sid);      // K % 4 == 0: one RNG group
                        if (blockIdx.x == 0 && p.z_out) *reinterpret_cast<f32x4*>(p.z_out + (size_t)n * K + k) = fa[g][m];
                    }
                }
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m)
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[g][m][t], fb[g][t], acc[m], 0, 0, 0);
    }
    if (wave > 0) {
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave - 1][m][r][lane] = acc[m][r];
    }
    __syncthreads();
    if (wave != 0) return;
    // wave 0 owns the tile from here: element (row 32m + (r&3) + 8(r>>2) + 4lh, feature fp0 + li)
    const float bias = p.bias[frow];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            acc[m][r] = ((acc[m][r] + red[0][m][r][lane]) + red[1][m][r][lane]) + red[2][m][r][lane] + bias;
    T* const yo = reinterpret_cast<T*>(p.y);
    T* const ao = reinterpret_cast<T*>(p.a);
    float sc, sf;
    if (p.bne) {                                   // eval: BatchNorm folded into one scale / shift per feature
        sc = p.bne[fpl]; sf = p.bne[F + fpl];
    } else {
        // BatchNorm1d over the batch: mean, then the centred sum of squares (biased variance for the normalisation,
        // unbiased into running_var -- nn.BatchNorm1d), each lane half over its rows, the halves joined by a lane swap.
        // What the statistics see is the value the backward pass will read, i.e. y as stored (rounded for a narrow T).
        float s = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n < B) {
                    float v = acc[m][r];
                    if (sizeof(T) != 4) { v = (float)(T)v; acc[m][r] = v; }
                    s += v;
                }
            }
        s += __shfl_xor(s, 32);
        const float mean = s / (float)B;
        float q = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (n < B) { const float d = acc[m][r] - mean; q = fmaf(d, d, q); }
            }
        q += __shfl_xor(q, 32);
        const float var = q / (float)B;
        const float rstd = 1.0f / sqrtf(var + p.eps);
        sc = p.gamma[frow] * rstd; sf = p.beta[frow] - mean * sc;
        if (lh == 0) {
            p.bn[fpl] = sc; p.bn[F + fpl] = sf; p.bn[2 * F + fpl] = mean; p.bn[3 * F + fpl] = rstd;
            const float unb = B > 1 ? var * ((float)B / (float)(B - 1)) : var;
            p.rmean[frow] = p.momentum * mean + (1.0f - p.momentum) * p.rmean[frow];
            p.rvar[frow] = p.momentum * unb + (1.0f - p.momentum) * p.rvar[frow];
        }
        if (blockIdx.x == 0 && lane == 0 && p.batches) p.batches[0] += 1;
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = 32 * m + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (n >= B) continue;
            const float v = acc[m][r];
            if (!p.bne) st1<T>(yo + (size_t)n * F + fpl, v);
            st1<T>(ao + (size_t)n * F + fpl, fmaxf(fmaf(v, sc, sf), 0.f));
        }
}

bool launch_fc_fwd_fused(int dt, const float* z, const float* W, const float* bias, void* y, void* a, const float* gamma,
                         const float* beta, float* rmean, float* rvar, int64_t* batches, float* bn, const float* bne,
                         float* z_out, const DevState* st, uint32_t sid, int B, int K, int C0, float momentum, float eps,
                         hipStream_t s) {
    if (B > 256 || (C0 * 16) % 32 != 0 || (!z && (K & 3) != 0)) return false;       // caller falls back to the generic kernels
    FcFwdArgs p{z, W, bias, y, a, gamma, beta, rmean, rvar, batches, bn, bne, z_out, st, sid, B, K, C0, momentum, eps};
    const dim3 grid(C0 * 16 / 32), blk(256);
    const int mt = (B + 31) / 32;
#define FCF(T, MT) hipLaunchKernelGGL((k_fc_fwd_mfma<T, MT>), grid, blk, 0, s, p)
    SIGGAN_DT_SWITCH(dt, T, {
        if (mt <= 1) FCF(T, 1); else if (mt <= 2) FCF(T, 2); else if (mt <= 4) FCF(T, 4); else FCF(T, 8);
    });
#undef FCF
    return true;
}

// ------------------------------------------------------------------------------------------
// backward: relu mask + BatchNorm1d backward + dW = dy^T z + db, one workgroup per 32 features
// ------------------------------------------------------------------------------------------
struct FcBwdArgs {
    const void* da;        // d(relu output) [B][F], element type T
    const void* y;         // pre-BN [B][F], T
    const float* z;        // [B][K]
    float* bn;             // [6F]: scale | shift | mean | rstd (from the forward)
    float* dW; float* db;  // [F][K], [F] torch order
    float* dgamma; float* dbeta;
    int B, K, C0;
};

template <class T>
__global__ __launch_bounds__(256) void k_fc_bwd_mfma(const FcBwdArgs p) {
    extern __shared__ float sdy[];                 // [Bpad][33] dy of the tile (rows >= B zero), then [Bpad][33] xhat
    __shared__ float s0[8][32], s1[8][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int F = p.C0 * 16, K = p.K, B = p.B;
    const int Bp = (B + 1) & ~1;
    const int fp0 = blockIdx.x * 32;
    const int fl = tid & 31, nl = tid >> 5;                      // feature lane, row lane (8 row lanes)
    const int fp = fp0 + fl;
    const T* const da = reinterpret_cast<const T*>(p.da);
    const T* const y = reinterpret_cast<const T*>(p.y);
    const float sc = p.bn[fp], sf = p.bn[F + fp], mu = p.bn[2 * F + fp], rs = p.bn[3 * F + fp];
    // pass 1: masked gradient and its two batch sums (four rows' loads in flight at a time; xhat parked in LDS for pass 2)
    float* const sxh = sdy + Bp * 33;
    float a0 = 0.f, a1 = 0.f;
    for (int nb = nl; nb < Bp; nb += 32) {
        float yy[4], gg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = nb + 8 * u;
            yy[u] = n < B ? ld1<T>(y + (size_t)n * F + fp) : 0.f;
            gg[u] = n < B ? ld1<T>(da + (size_t)n * F + fp) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int n = nb + 8 * u;
            if (n >= Bp) continue;
            const float g = (n < B && fmaf(yy[u], sc, sf) > 0.f) ? gg[u] : 0.f;
            const float xh = n < B ? (yy[u] - mu) * rs : 0.f;
            sdy[n * 33 + fl] = g; sxh[n * 33 + fl] = xh;
            a0 += g; a1 = fmaf(g, xh, a1);
        }
    }
    s0[nl][fl] = a0; s1[nl][fl] = a1;
    __syncthreads();
    float dbeta = 0.f, dgamma = 0.f;
#pragma unroll
    for (int q = 0; q < 8; ++q) { dbeta += s0[q][fl]; dgamma += s1[q][fl]; }
    const float invB = 1.0f / (float)B;
    const float c1 = dbeta * invB, c2 = dgamma * invB;
    // pass 2: dy = scale * (g - mean(g) - xhat * mean(g * xhat)); its batch sum is the bias gradient
    float sdb = 0.f;
    for (int n = nl; n < B; n += 8) {
        const float d = sc * (sdy[n * 33 + fl] - c1 - sxh[n * 33 + fl] * c2);
        sdy[n * 33 + fl] = d;
        sdb += d;
    }
    __syncthreads();                                  // every read of s0 / s1 above is done: reuse s0 for the bias sums
    s0[nl][fl] = sdb;
    const int frow_t = (fp % p.C0) * 16 + fp / p.C0;
    if (nl == 0) { p.dbeta[frow_t] = dbeta; p.dgamma[frow_t] = dgamma; }
    __syncthreads();
    if (nl == 0) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 8; ++q) t += s0[q][fl];
        p.db[frow_t] = t;
    }
    // dW[f][k] = sum_n dy[n][f'] z[n][k]: A = dy^T (from LDS), B = z (global, coalesced over k); wave w takes the
    // 32-column groups w, w+4, ...
    const int fpi = fp0;                              // accumulator row i <-> feature fp0 + i
    for (int kt = wave; kt * 32 < K; kt += 4) {
        const int k = kt * 32 + li;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        for (int n0 = 0; n0 < Bp; n0 += 32) {         // 16 k-steps per group: their z loads are in flight together
            float fb[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int n = n0 + 2 * u + lh;
                fb[u] = (n < B && k < K) ? p.z[(size_t)n * K + k] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int n = n0 + 2 * u + lh;
                const float fa = n < Bp ? sdy[n * 33 + li] : 0.f;
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb[u], acc, 0, 0, 0);
            }
        }
        if (k < K) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f2 = fpi + (r & 3) + 8 * (r >> 2) + 4 * lh;
                p.dW[(size_t)((f2 % p.C0) * 16 + f2 / p.C0) * K + k] = acc[r];
            }
        }
    }
}

bool launch_fc_bwd_fused(int dt, const void* da, const void* y, const float* z, float* bn, float* dW, float* db, float* dgamma,
                         float* dbeta, int B, int K, int C0, hipStream_t s) {
    const int Bp = (B + 1) & ~1;
    const size_t lds = (size_t)2 * Bp * 33 * sizeof(float);      // dy and xhat of the tile
    if (lds > 96 * 1024 || (C0 * 16) % 32 != 0) return false;
    FcBwdArgs p{da, y, z, bn, dW, db, dgamma, dbeta, B, K, C0};
    SIGGAN_DT_SWITCH(dt, T, hipLaunchKernelGGL(k_fc_bwd_mfma<T>, dim3(C0 * 16 / 32), dim3(256), lds, s, p));
    return true;
}

}  // namespace siggan
