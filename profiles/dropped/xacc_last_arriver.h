// xacc.h -- BatchNorm statistics without a finalize launch: exact accumulators + a last-arriver finalize.
//
// A per-channel statistic (sum of y, sum of y^2, sum of dy, sum of dy * xhat) is the sum of one fp32 partial per workgroup
// of the kernel that produces the tensor.  Round 1-3 wrote the partials as rows and ran a finalize kernel behind the
// producer (fixed order: bitwise reproducible); folding that kernel into the producer failed twice because the last
// workgroup had to fetch every row through an agent-scope fence.  Here the partials never exist as rows:
//
//   * every workgroup ADDS its partial into a 160-bit fixed-point accumulator with 64-bit integer atomics (XAcc: five
//     32-bit digits held in 64-bit words, least significant bit 2^-104, so up to 2^31 addends cannot carry out of a
//     word).  Integer addition commutes: the sum does not depend on the order the workgroups arrive in -- bitwise
//     reproducible like the fixed-order rows, and exact (the only rounding is the partial's own and the final conversion);
//   * the data flow between workgroups is agent-scope atomics only (performed at the device's coherence point), so the
//     workgroup whose ticket add returns last may read the totals with agent-scope atomic loads and NO fence
//     (MI355X_MICROARCH.md, "Valid forms": all producers add to ONE counter, the last adder loads after its add returned,
//     its other waves after a workgroup barrier);
//   * that workgroup finalizes (mean / rstd / scale / shift + running statistics, or the two BatchNorm-backward means +
//     dgamma / dbeta), writes the per-channel table the NEXT kernel reads as ordinary kernel output, and leaves accumulators
//     and ticket zero for the next use (atomic stores: no stale copy stays in an XCD's L2).
//
// A partial that is not finite, or too large for the window (|p| >= 2^55), poisons the accumulator: the total reads NaN (the
// fp16 overflow guard relies on non-finite gradients staying non-finite).  |p| < 2^-104 is flushed to zero; bits of a partial
// below 2^-104 are truncated toward zero.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "act.h"

namespace siggan {

struct XAcc { unsigned long long d[6]; };      // d[0..4]: digits (signed accumulation, two's complement); d[5]: poison count

__device__ __forceinline__ void xacc_add(XAcc* a, float p) {
    const unsigned u = __float_as_uint(p);
    const int e = (int)((u >> 23) & 0xffu);
    unsigned m = u & 0x7fffffu;
    if (e == 0xff) { atomicAdd(&a->d[5], 1ull); return; }            // inf / NaN
    if (e != 0) m |= 0x800000u;
    if (m == 0) return;
    int pos = (e == 0 ? 1 : e) - 150 + 104;                          // bit position of m's least significant bit
    if (pos + 24 > 159) { atomicAdd(&a->d[5], 1ull); return; }       // |p| >= 2^55
    unsigned long long v = m;
    if (pos < 0) { v = pos <= -24 ? 0ull : (v >> (-pos)); pos = 0; }
    if (v == 0) return;
    const int i = pos >> 5;
    v <<= (pos & 31);                                                // < 2^55; digit i + 1 <= 4 whenever its part is non-zero
    unsigned long long lo = v & 0xffffffffull, hi = v >> 32;
    if (u >> 31) { lo = 0ull - lo; hi = 0ull - hi; }
    if (lo) atomicAdd(&a->d[i], lo);
    if (hi) atomicAdd(&a->d[i + 1], hi);
}

// the total (every adder's atomics are complete: the caller holds the last ticket); leaves the accumulator zero
__device__ __forceinline__ double xacc_take(XAcc* a) {
    unsigned long long w[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) w[i] = __hip_atomic_load(&a->d[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (w[i]) __hip_atomic_store(&a->d[i], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (w[5]) return __builtin_nan("");
    unsigned dg[5];
    long long c = 0;
#pragma unroll
    for (int i = 0; i < 5; ++i) { const long long t = (long long)w[i] + c; dg[i] = (unsigned)(t & 0xffffffffll); c = t >> 32; }
    const bool neg = c < 0;                                          // (c is the sign extension: 0 or -1)
    if (neg) {
        unsigned long long carry = 1;
#pragma unroll
        for (int i = 0; i < 5; ++i) { const unsigned long long t = (unsigned long long)(~dg[i]) + carry; dg[i] = (unsigned)t; carry = t >> 32; }
    }
    double v = 0.0;
#pragma unroll
    for (int i = 4; i >= 0; --i) v = v * 4294967296.0 + (double)dg[i];
    v = ldexp(v, -104);
    return neg ? -v : v;
}

// What the last workgroup of a statistics epilogue does with the totals.  kind 1: forward statistics of the pre-BatchNorm
// tensor y, accumulated as sum(y - shift), sum((y - shift)^2) with shift = the channel's running mean BEFORE this update (any
// value every workgroup agrees on keeps E[d^2] - E[d]^2 away from cancellation; the running mean tracks the batch mean);
// kind 2: BatchNorm-backward sums sum(dr), sum(dr * xhat).
struct BnFin {
    int kind;             // 0 none, 1 forward, 2 backward
    XAcc* acc;            // [2][C]
    unsigned* ticket;     // one counter per site, zero between uses
    float* bn;            // the layer's table [scale | shift | mean | rstd | c1 | c2], C floats each
    const float* gamma;   // forward
    const float* beta;
    float* rmean;         // running statistics, updated in place as nn.BatchNorm does (momentum, unbiased variance)
    float* rvar;
    int64_t* batches;     // num_batches_tracked
    float* dgamma;        // backward
    float* dbeta;
    int64_t R;            // rows the statistic runs over
    float momentum, eps;
};

// Called by EVERY thread of EVERY workgroup of the producing kernel after its xacc_add calls (uniform control flow).
// nwg: workgroups taking a ticket; sflag: one LDS word.  Returns after the table is written when this workgroup was last.
__device__ __forceinline__ void bn_fin_last_arriver(const BnFin& f, int C, unsigned nwg, unsigned* sflag) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // this wave's accumulator atomics are performed
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(f.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *sflag = t;
    }
    __syncthreads();
    if (*sflag != nwg - 1) return;
    if (threadIdx.x == 0) __hip_atomic_store(f.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float invR = 1.0f / (float)f.R;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const double s = xacc_take(f.acc + c), q = xacc_take(f.acc + C + c);
        if (f.kind == 1) {
            const double dm = s / (double)f.R;
            const float mean = (float)((double)f.rmean[c] + dm);
            float var = (float)(q / (double)f.R - dm * dm);
            var = var > 0.f ? var : 0.f;                             // (a NaN total gives var = 0 and a NaN mean, as k_bn_train_fin does)
            const float rstd = 1.0f / sqrtf(var + f.eps);
            const float sc = f.gamma[c] * rstd;
            f.bn[c] = sc; f.bn[C + c] = f.beta[c] - mean * sc; f.bn[2 * C + c] = mean; f.bn[3 * C + c] = rstd;
            const float unb = f.R > 1 ? var * ((float)f.R / (float)(f.R - 1)) : var;
            f.rmean[c] = f.momentum * mean + (1.0f - f.momentum) * f.rmean[c];
            f.rvar[c] = f.momentum * unb + (1.0f - f.momentum) * f.rvar[c];
        } else {
            const float sf = (float)s, qf = (float)q;
            f.dbeta[c] = sf; f.dgamma[c] = qf;
            f.bn[4 * C + c] = sf * invR; f.bn[5 * C + c] = qf * invR;
        }
    }
    if (f.kind == 1 && threadIdx.x == 0 && f.batches) f.batches[0] += 1;
}

// Column sums of a statistics epilogue -> accumulators -> last-arriver finalize.  Called by every thread of the workgroup
// (uniform); `owner`: this thread holds the workgroup's sums s0 / s1 of the four channels from co on.  sflag: one LDS word.
__device__ __forceinline__ void bn_stats_commit(const BnFin& f, int C, int co, bool owner, const f32x4 s0, const f32x4 s1,
                                                unsigned nwg, unsigned* sflag) {
    if (owner) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { xacc_add(f.acc + co + e, s0[e]); xacc_add(f.acc + C + co + e, s1[e]); }
    }
    bn_fin_last_arriver(f, C, nwg, sflag);
}

}  // namespace siggan
