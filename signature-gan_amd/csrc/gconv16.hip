// gconv16.hip -- the implicit-GEMM kernels of gconv.hip for 16-bit operands (bf16 / f16 activations and packed
// weights, fp32 accumulate): v_mfma_f32_32x32x16_{bf16,f16}, gfx950 only.  BASELINE.json configs[2] / configs[4].
//
// Same GEMM views, tile shapes, split-K / slab protocol and fused epilogues as the fp32 kernels; what differs is the
// operand path.  A K-tile is 32 elements (64 bytes per row):
//   k_gconv16 : both operands are K-contiguous in memory (NHWC im2col rows; [Co][tap*Ci + ci] weight packs), so the LDS
//               tiles are [row][32 + 8 pad] (80-byte row stride = 5 sixteen-byte slots, odd) and a lane's MFMA
//               fragment -- A[row l&31][k = 8(l>>5) .. +7] -- is ONE conflict-free ds_read_b128;
//   k_wgrad16 : the contraction axis is the pixel, both operands are pixel-major in memory ([pix][channel]), so the
//               tiles are staged as they lie ([k][channels + 32 pad]: coalesced 16-byte chunks) and the fragments are
//               gathered with ds_read_b64_tr_b16, the hardware transposing LDS read (two per fragment); the 64-byte
//               row pad puts the four k-rows of a read on disjoint bank quarters.
// 16-bit storage makes the step launch- and HBM-bound rather than MFMA-bound, so these kernels keep the
// fp32 kernels' simple one-tile-ahead register staging; the accumulators, split-K slabs, weight-gradient outputs and
// every epilogue computation are fp32.
#include "gconv.h"
#include <hip/hip_ext.h>

namespace siggan {

static constexpr int BK = 32;     // K-tile, elements

__device__ __forceinline__ int xcd_remap16(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

template <class T> struct Mma;
template <> struct Mma<bf16_t> {
    typedef bf16x8 V;
    static __device__ __forceinline__ f32x16 run(V a, V b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct Mma<f16_t> {
    typedef f16x8 V;
    static __device__ __forceinline__ f32x16 run(V a, V b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

// KQ = 2: two wave quads per workgroup, each on its own half of the K range with its own LDS buffers, accumulators added
// through LDS before the epilogue -- a two-way K split without slabs or a second kernel (see k_gconv in gconv.hip)
template <class T, int BM, int BN, int WM, int WN, int NSET = 2, int KQ = 1>
__global__ __launch_bounds__(256 * KQ) void k_gconv16(const GConvArgs a) {
    typedef typename Mma<T>::V Frag;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int PA = BM / 64 > 0 ? BM / 64 : 1, PB = BN / 64 > 0 ? BN / 64 : 1;   // staging passes (64 rows x 4 chunks per pass)
    constexpr int LD = BK + 8;                       // elements; 80-byte rows
    constexpr int QUAD_SHORTS = 2 * LD * (BM + BN);
    // (KQ = 2: the fp32 hand-over of quad 1's accumulators needs 64 lanes x 16 registers x 4 waves x 4 bytes per 32x32 block)
    constexpr int RED_SHORTS = KQ > 1 ? TM * TN * 4 * 16 * 64 * 2 : 0;
    __shared__ __attribute__((aligned(16))) unsigned short smem_raw[KQ * QUAD_SHORTS > QUAD_SHORTS + RED_SHORTS ? KQ * QUAD_SHORTS : QUAD_SHORTS + RED_SHORTS];
    const int quad = KQ > 1 ? (int)(threadIdx.x >> 8) : 0;
    T* const sA = reinterpret_cast<T*>(smem_raw) + quad * QUAD_SHORTS;
    T* const sB = sA + 2 * LD * BM;

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;      // thread / wave index inside the quad
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const T* const in = reinterpret_cast<const T*>(a.in);
    const T* const wp = reinterpret_cast<const T*>(a.wp);

    const int tiles_n = a.Co / BN;
    const int bid = xcd_remap16(blockIdx.x, gridDim.x);
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const int cls = blockIdx.z, ph = cls >> 1, pw = cls & 1;
    const int Hr = 1 << a.lgHr, Wr = 1 << a.lgWr;
    const int ntaps = a.form == 0 ? 16 : 4;
    const int Ktot = ntaps * a.Ci;
    const int lgcpt = 31 - __builtin_clz(a.Ci / BK);
    const int nk_all = ntaps << lgcpt;
    const int kper = (nk_all + gridDim.y * KQ - 1) / (gridDim.y * KQ);
    const int k_lo = (blockIdx.y * KQ + quad) * kper;
    const int k_hi = min(nk_all, k_lo + kper);
    const int nk = k_hi - k_lo;

    // staging coordinates: row rloc + 64p, 16-byte chunk kc (8 elements) of the 32-element k-row
    const int kc = tid & 3, rloc = tid >> 2;
    const T* a_base[PA];
    int a_ih0[PA], a_iw0[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int m = m0 + ((rloc + 64 * p) & (BM - 1));
        if (m < a.M) {
            const int n = m >> (a.lgHr + a.lgWr);
            const int rh = (m >> a.lgWr) & (Hr - 1), rw = m & (Wr - 1);
            a_base[p] = in + (size_t)n * a.Hi * a.Wi * a.Ci + kc * 8;
            if (a.form == 0) { a_ih0[p] = 2 * rh - 1; a_iw0[p] = 2 * rw - 1; }
            else             { a_ih0[p] = rh + ph;    a_iw0[p] = rw + pw; }
        } else {
            a_base[p] = in; a_ih0[p] = -(1 << 20); a_iw0[p] = -(1 << 20);
        }
    }
    const T* wcur = wp + ((size_t)cls * a.Co + n0 + (rloc & (BN - 1))) * Ktot + kc * 8 + (size_t)k_lo * BK;
    const T* const zeros = reinterpret_cast<const T*>(a.zeros);

    const T* a_cur[PA];
    int a_step[PA];
    int l_tap = k_lo >> lgcpt, l_cc = k_lo & ((1 << lgcpt) - 1);
    auto set_tap = [&]() __attribute__((always_inline)) {
        int dh, dw;
        if (a.form == 0) { dh = l_tap >> 2; dw = l_tap & 3; } else { dh = -(l_tap >> 1); dw = -(l_tap & 1); }
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int ih = a_ih0[p] + dh, iw = a_iw0[p] + dw;
            const bool ok = (unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi;
            a_cur[p] = ok ? a_base[p] + ((size_t)(ih * a.Wi + iw) * a.Ci + l_cc * BK) : zeros;
            a_step[p] = ok ? BK : 0;
        }
    };
    s16x8 ra[NSET][PA], rb[NSET][PB];    // register sets: tile t waits in set t % NSET
    auto load_tile = [&](int set = 0) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            ra[set][p] = *reinterpret_cast<const s16x8*>(a_cur[p]);
            a_cur[p] += a_step[p];
        }
#pragma unroll
        for (int p = 0; p < PB; ++p) rb[set][p] = *reinterpret_cast<const s16x8*>(wcur + (size_t)(64 * p) * Ktot);
        wcur += BK;
        if (++l_cc == (1 << lgcpt)) { l_cc = 0; ++l_tap; set_tap(); }
    };
    auto store_tile = [&](int buf, int set = 0) __attribute__((always_inline)) {
        T* dA = sA + buf * LD * BM + (rloc & (BM - 1)) * LD + kc * 8;
        T* dB = sB + buf * LD * BN + (rloc & (BN - 1)) * LD + kc * 8;
#pragma unroll
        for (int p = 0; p < PA; ++p) *reinterpret_cast<s16x8*>(dA + 64 * p * LD) = ra[set][p];
#pragma unroll
        for (int p = 0; p < PB; ++p) *reinterpret_cast<s16x8*>(dB + 64 * p * LD) = rb[set][p];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // one K-tile out of LDS buffer `buf`; st: the LDS write of tile kt+1 out of register set `sset` rides along behind the first
    // MFMAs, ld: the loads of a later tile into that set behind the second
    auto tile = [&](const int buf, const int sset, const bool st, const bool ld) __attribute__((always_inline)) {
        const T* pA = sA + buf * LD * BM + (wm * (32 * TM) + li) * LD + 8 * lh;
        const T* pB = sB + buf * LD * BN + (wn * (32 * TN) + li) * LD + 8 * lh;
        Frag fa[2][TM], fb[2][TN];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[s][i] = *reinterpret_cast<const Frag*>(pA + 32 * i * LD + 16 * s);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[s][j] = *reinterpret_cast<const Frag*>(pB + 32 * j * LD + 16 * s);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(fa[s][i], fb[s][j], acc[i][j]);
            if (s == 0 && st) store_tile(buf ^ 1, sset);
            if (s == 1 && ld) load_tile(sset);
        }
        __syncthreads();
    };
    if (nk >= 2 * NSET + 1) {
        // loads NSET tiles ahead, as in the fp32 kernel (gconv.hip): register set t % NSET, steady loop unrolled by NSET with
        // unconditional loads so that the wait before the LDS write covers the oldest set only
        set_tap();
#pragma unroll
        for (int t = 0; t < NSET; ++t) load_tile(t);
        store_tile(0, 0);
        load_tile(0);
        __syncthreads();
        int kt = 0;
        for (; kt + NSET + NSET < nk; kt += NSET) {
#pragma unroll
            for (int h = 0; h < NSET; ++h) tile(h & 1, (h + 1) % NSET, true, true);
        }
        for (; kt < nk; kt += NSET) {
#pragma unroll
            for (int h = 0; h < NSET; ++h)
                if (kt + h < nk) tile(h & 1, (h + 1) % NSET, kt + h + 1 < nk, kt + h + 1 + NSET < nk);
        }
    } else {
        if (nk > 0) {
            set_tap();
            load_tile();
            store_tile(0);
            if (nk > 1) load_tile();
        }
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) tile(kt & 1, 0, kt + 1 < nk, kt + 2 < nk);
    }

    // ---- epilogue (fp32 arithmetic; the split-K slabs are fp32, the tensor itself is T) ------------------------
    // The accumulators go through LDS (free after the loop's last barrier), 64 rows at a time, so that the fused arithmetic
    // and the stores work on four consecutive channels per lane: 8-byte stores of T (16-byte ones into a slab) covering a
    // row's contiguous bytes, instead of one 2-byte store per element.  Same values, same operations per element.
    const bool to_slab = gridDim.y > 1;
    float* const slab = a.slab + (size_t)blockIdx.y * a.slab_stride;
    T* const outp = reinterpret_cast<T*>(a.out);
    const T* const aref = reinterpret_cast<const T*>(a.aref);
    const int epi = to_slab ? (int)EPI_RAW : a.epi;
    constexpr int LDT = BN + 4, SR = BM < 64 ? BM : 64, NPASS = BM / SR;
    static_assert(SR * LDT * 4 <= 2 * LD * (BM + BN) * 2, "a 64-row slice of the output tile must fit the staging buffers");
    float* const sT = reinterpret_cast<float*>(smem_raw);
    if (KQ > 1) {                                    // quad 1's accumulators -> LDS -> added to quad 0's (lane-for-lane)
        float* const red = reinterpret_cast<float*>(smem_raw + QUAD_SHORTS);
        if (quad == 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[(((wave * TM + i) * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
        }
        __syncthreads();
        if (quad == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += red[(((wave * TM + i) * TN + j) * 16 + r) * 64 + lane];
        }
    }
    constexpr int C4 = BN / 4, RPP = 256 / C4;
    const int c4 = tid % C4, r0 = tid / C4;
    const int co = n0 + c4 * 4;
    const f32x4 bias4 = epi == EPI_BIAS_LRELU_DROP ? *reinterpret_cast<const f32x4*>(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
    if (epi == EPI_AFFINE_RELU) { sc4 = *reinterpret_cast<const f32x4*>(a.scale + co); sh4 = *reinterpret_cast<const f32x4*>(a.shift + co); }
    const bool use_noise = (epi == EPI_BIAS_LRELU_DROP || epi == EPI_LRELU_BWD) && a.noise != nullptr;
    BnBwdParams bq;
    f32x4 st0 = {0.f, 0.f, 0.f, 0.f}, st1 = st0;
    if (epi == EPI_BN_BWD_STATS) bq = bn_bwd_params(a.bnp, a.Co, co);
#pragma unroll
    for (int q = 0; q < NPASS; ++q) {
        if (q) __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int rb = wm * (32 * TM) + 32 * i - q * SR;              // this wave's 32-row block inside the slice?
            if (rb < 0 || rb >= SR || quad != 0) continue;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sT[(rb + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDT + wn * (32 * TN) + 32 * j + li] = acc[i][j][r];
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < SR / RPP; ++p) {
            const int row = r0 + RPP * p, m = m0 + q * SR + row;
            if (m >= a.M || quad != 0) continue;
            const int n = m >> (a.lgHr + a.lgWr);
            size_t opix;
            if (a.form == 0) {
                opix = (size_t)m;
            } else {
                const int rh = (m >> a.lgWr) & (Hr - 1), rw = m & (Wr - 1);
                opix = ((size_t)n * a.Ho + 2 * rh + ph) * a.Wo + 2 * rw + pw;
            }
            const size_t o = opix * a.Co + co;
            f32x4 v = *reinterpret_cast<const f32x4*>(sT + row * LDT + c4 * 4);
            if (epi == EPI_BN_BWD_STATS) {
                bn_bwd_stat_terms<T>(v, ld4<T>(aref + o), bq, st0, st1);      // (v as it is stored: rounded to T)
            } else if (epi == EPI_BIAS_LRELU_DROP) {
#pragma unroll
                for (int e = 0; e < 4; ++e) { float t = v[e] + bias4[e]; v[e] = t > 0.f ? t : t * a.slope; }
                if (use_noise) {
                    const f32x4 nz = *reinterpret_cast<const f32x4*>(a.noise + (size_t)n * a.Co + co);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= nz[e];
                }
            } else if (epi == EPI_AFFINE_RELU) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = fmaxf(fmaf(v[e], sc4[e], sh4[e]), 0.f);
            } else if (epi == EPI_LRELU_BWD) {
                const f32x4 ar = ld4<T>(aref + o);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= ar[e] > 0.f ? 1.f : a.slope;
                if (use_noise) {
                    const f32x4 nz = *reinterpret_cast<const f32x4*>(a.noise + (size_t)n * a.Co + co);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] *= nz[e];
                }
            }
            if (to_slab) *reinterpret_cast<f32x4*>(slab + o) = v; else st4<T>(outp + o, v);
        }
    }
    if (epi == EPI_BN_BWD_STATS) {                   // the tile's column sums: see k_gconv
        __syncthreads();
        if (quad == 0) {
            *reinterpret_cast<f32x4*>(sT + (size_t)tid * 8) = st0;
            *reinterpret_cast<f32x4*>(sT + (size_t)tid * 8 + 4) = st1;
        }
        __syncthreads();
        if (quad == 0 && r0 == 0) {
#pragma unroll 4
            for (int k = 1; k < RPP; ++k) {
                st0 += *reinterpret_cast<const f32x4*>(sT + (size_t)(k * C4 + c4) * 8);
                st1 += *reinterpret_cast<const f32x4*>(sT + (size_t)(k * C4 + c4) * 8 + 4);
            }
            const size_t prow = (size_t)cls * (gridDim.x / tiles_n) + bid / tiles_n;
            *reinterpret_cast<f32x4*>(a.stat0 + prow * a.Co + co) = st0;
            *reinterpret_cast<f32x4*>(a.stat1 + prow * a.Co + co) = st1;
        }
    }
}

void launch_gconv16(int cfg, const GConvArgs& a, dim3 grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1, int kq) {
    if (kq == 2) {          // 64x64 tiles, two-way K split inside the workgroup
        if (a.dt == DT_BF16) hipExtLaunchKernelGGL((k_gconv16<bf16_t, 64, 64, 2, 2, 2, 2>), grid, dim3(512), 0, st, e0, e1, 0, a);
        else hipExtLaunchKernelGGL((k_gconv16<f16_t, 64, 64, 2, 2, 2, 2>), grid, dim3(512), 0, st, e0, e1, 0, a);
        return;
    }
#define GC16(T, BM, BN, WM, WN) hipExtLaunchKernelGGL((k_gconv16<T, BM, BN, WM, WN>), grid, dim3(256), 0, st, e0, e1, 0, a)
    if (a.dt == DT_BF16) {
        if (cfg == 0) GC16(bf16_t, 128, 128, 2, 2); else if (cfg == 2) GC16(bf16_t, 64, 64, 2, 2); else GC16(bf16_t, 128, 32, 4, 1);
    } else {
        if (cfg == 0) GC16(f16_t, 128, 128, 2, 2); else if (cfg == 2) GC16(f16_t, 64, 64, 2, 2); else GC16(f16_t, 128, 32, 4, 1);
    }
#undef GC16
}

// ------------------------------------------------------------------------------------------
// weight gradient: C[i][j] = sum_pix S[pix][i] * Lg[pix][j],  j = tap*Cl + l   (fp32 result)
// ------------------------------------------------------------------------------------------
// transposing fragment read: for the [k][col] LDS image M (row stride ld elements) returns, for this lane,
// M[k0 + 0..3][col] with col = c0 + 16*((lane>>4)&1) + (lane&15) and k0 = kb + 8*(lane>>5) (+4 for the second call):
// the 16 lanes of a group supply the addresses of a 4-row x 16-column block (lane 4q+p: row q, columns 4p..4p+3) and
// receive its columns.
__device__ __forceinline__ s16x4 tr_read(const unsigned short* M, int ld, int kb, int c0, int lane) {
    const int g = lane >> 4, idx = lane & 15;
    const unsigned short* p = M + (kb + 8 * (g >> 1) + (idx >> 2)) * ld + c0 + 16 * (g & 1) + 4 * (idx & 3);
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((s16x4 __attribute__((address_space(3)))*)(p));
}

// BKP = pixels per K-tile (64: four MFMA k-steps per barrier)
template <class T, int BM, int BN, int WM, int WN, int BKP>
__global__ __launch_bounds__(256) void k_wgrad16(const WgradArgs a) {
    typedef typename Mma<T>::V Frag;
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    // row stride = 64 bytes mod 256: the four k-rows of a transposing read then sit on disjoint bank quarters
    constexpr int LDA = BM == 32 ? 32 : BM + 32, LDB = BN == 32 ? 32 : BN + 32;
    constexpr int CA = BM / 8, RA = 256 / CA, PA = BKP / RA > 0 ? BKP / RA : 1;     // 16-byte chunks per k-row, k-rows per pass
    constexpr int CB = BN / 8, RB = 256 / CB, PB = BKP / RB > 0 ? BKP / RB : 1;
    __shared__ __attribute__((aligned(16))) unsigned short smem[2 * BKP * (LDA + LDB)];
    unsigned short* const sA = smem;
    unsigned short* const sB = smem + 2 * BKP * LDA;
    const T* const S = reinterpret_cast<const T*>(a.S);
    const T* const L = reinterpret_cast<const T*>(a.L);
    const T* const zeros = reinterpret_cast<const T*>(a.zeros);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int N = 16 << a.lgCl;
    const int tiles_n = N / BN;
    // (XCD-aware order, as in k_wgrad: the tiles of one K split share an XCD's L2)
    const int lid = xcd_remap16(blockIdx.z * gridDim.x + blockIdx.x, gridDim.x * gridDim.z);
    const int bx = lid % gridDim.x, bz = lid / gridDim.x;
    const int i0 = (bx / tiles_n) * BM, j0 = (bx % tiles_n) * BN;
    const int kbeg = bz * a.kchunk;
    const int kend = min(a.K, kbeg + a.kchunk);
    const int nk = (kend - kbeg + BKP - 1) / BKP;
    const int Hs = 1 << a.lgHs, Ws = 1 << a.lgWs, Hl = 2 * Hs, Wl = 2 * Ws, Cl = 1 << a.lgCl;

    const int ca = tid % CA, ka = (tid / CA) % BKP;
    const int cb = tid % CB, kb = (tid / CB) % BKP;
    const int jj = j0 + cb * 8, tap = jj >> a.lgCl, lch = jj & (Cl - 1);
    const int kh = tap >> 2, kw = tap & 3;

    s16x8 ra[PA], rb[PB];
#define WG_LOAD_TILE(KT)                                                                              \
    {                                                                                                 \
        const int kbase = kbeg + (KT) * BKP;                                                           \
        _Pragma("unroll") for (int p = 0; p < PA; ++p) {                                              \
            const int pix = kbase + ka + RA * p;                                                      \
            const T* src = pix < kend ? S + ((size_t)pix * a.Cs + i0 + ca * 8) : zeros;               \
            ra[p] = *reinterpret_cast<const s16x8*>(src);                                            \
        }                                                                                             \
        _Pragma("unroll") for (int p = 0; p < PB; ++p) {                                              \
            const int pix = kbase + kb + RB * p;                                                      \
            const int n = pix >> (a.lgHs + a.lgWs);                                                   \
            const int ih = 2 * ((pix >> a.lgWs) & (Hs - 1)) - 1 + kh, iw = 2 * (pix & (Ws - 1)) - 1 + kw; \
            const bool ok = pix < kend && (unsigned)ih < (unsigned)Hl && (unsigned)iw < (unsigned)Wl; \
            const T* src = ok ? L + ((((size_t)n * Hl + ih) * Wl + iw) * Cl + lch) : zeros;           \
            rb[p] = *reinterpret_cast<const s16x8*>(src);                                            \
        }                                                                                             \
    }
#define WG_STORE_TILE(BUF)                                                                            \
    {                                                                                                 \
        _Pragma("unroll") for (int p = 0; p < PA; ++p)                                                \
            *reinterpret_cast<s16x8*>(sA + (BUF) * BKP * LDA + (ka + RA * p) * LDA + ca * 8) = ra[p]; \
        _Pragma("unroll") for (int p = 0; p < PB; ++p)                                                \
            *reinterpret_cast<s16x8*>(sB + (BUF) * BKP * LDB + (kb + RB * p) * LDB + cb * 8) = rb[p]; \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nk > 0) {
        WG_LOAD_TILE(0)
        WG_STORE_TILE(0)
        if (nk > 1) WG_LOAD_TILE(1)
    }
    __syncthreads();
    // column sums of S (bias gradient) in the first column tile, as in the fp32 kernel
    constexpr int NQ = 256 / BM, KQ = BKP / NQ;
    const bool bias_blk = a.db != nullptr && j0 == 0;
    const int bcol = tid % BM, kq = tid / BM;
    float bsum = 0.f;
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const unsigned short* tA = sA + buf * BKP * LDA;
        const unsigned short* tB = sB + buf * BKP * LDB;
        if (bias_blk) {
            const T* col = reinterpret_cast<const T*>(tA) + kq * KQ * LDA + bcol;
#pragma unroll
            for (int k = 0; k < KQ; ++k) bsum += (float)col[k * LDA];
        }
        constexpr int NS = BKP / 16;
        Frag fa[NS][TM], fb[NS][TN];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const s16x4 lo = tr_read(tA, LDA, 16 * s, wm * (32 * TM) + 32 * i, lane);
                const s16x4 hi = tr_read(tA, LDA, 16 * s + 4, wm * (32 * TM) + 32 * i, lane);
                fa[s][i] = __builtin_bit_cast(Frag, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const s16x4 lo = tr_read(tB, LDB, 16 * s, wn * (32 * TN) + 32 * j, lane);
                const s16x4 hi = tr_read(tB, LDB, 16 * s + 4, wn * (32 * TN) + 32 * j, lane);
                fb[s][j] = __builtin_bit_cast(Frag, s16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
            }
        }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(fa[s][i], fb[s][j], acc[i][j]);
            if (s == 0 && kt + 1 < nk) WG_STORE_TILE(buf ^ 1)
            if (s == 1 && kt + 2 < nk) WG_LOAD_TILE(kt + 2)
        }
        __syncthreads();
    }
#undef WG_LOAD_TILE
#undef WG_STORE_TILE
    (void)li; (void)lh;
    float* const fsm = reinterpret_cast<float*>(smem);
    if (bias_blk) {                                   // the main loop's last barrier has passed: LDS is free
        fsm[kq * BM + bcol] = bsum;
        __syncthreads();
        if (tid < BM) {
            float t = fsm[tid];
#pragma unroll
            for (int q = 1; q < NQ; ++q) t += fsm[q * BM + tid];
            if (gridDim.z == 1) a.db[i0 + tid] = t;
            else a.slab[(size_t)gridDim.z * a.Cs * N + (size_t)bz * a.Cs + i0 + tid] = t;
        }
    }
    const int li2 = lane & 31, lh2 = lane >> 5;
    if (gridDim.z == 1 && a.dw) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * (32 * TM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh2;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = j0 + wn * (32 * TN) + 32 * j + li2;
                    a.dw[((size_t)row * Cl + (col & (Cl - 1))) * 16 + (col >> a.lgCl)] = acc[i][j][r];
                }
            }
        return;
    }
    float* const out = a.slab + (size_t)bz * a.Cs * N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = i0 + wm * (32 * TM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh2;
#pragma unroll
            for (int j = 0; j < TN; ++j)
                out[(size_t)row * N + j0 + wn * (32 * TN) + 32 * j + li2] = acc[i][j][r];
        }
}

void launch_wgrad16(bool small, const WgradArgs& a, dim3 grid, hipStream_t st, hipEvent_t e0, hipEvent_t e1) {
#define WG16(T, BM, BN, WM, WN) hipExtLaunchKernelGGL((k_wgrad16<T, BM, BN, WM, WN, 64>), grid, dim3(256), 0, st, e0, e1, 0, a)
    if (a.dt == DT_BF16) { if (small) WG16(bf16_t, 32, 128, 1, 4); else WG16(bf16_t, 64, 64, 2, 2); }
    else                 { if (small) WG16(f16_t, 32, 128, 1, 4); else WG16(f16_t, 64, 64, 2, 2); }
#undef WG16
}

}  // namespace siggan
