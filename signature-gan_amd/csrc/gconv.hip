// gconv.hip -- fp32 MFMA implicit-GEMM kernels for the 4x4 / stride-2 / pad-1 convolution family
// of the signature GAN (gfx950 only).
//
//   "down" form : Conv2d forward (discriminator_vanilla_gan.py:51-58) and the input-gradient of
//                 ConvTranspose2d (generator_vanilla_gan.py:46-54)
//   "up"   form : ConvTranspose2d forward and the input-gradient of Conv2d, as four sub-pixel
//                 (output-parity) classes of 2x2 taps each -- no col2im scatter
//   wgrad       : weight gradient of both, split over the pixel (K) axis into partial slabs
//
// GEMM view: C[m][n] = sum_k A[m][k] * Bw[n][k];  m = output pixel (down) / input pixel (up),
// n = output channel, k = (tap, input channel).  One workgroup = 4 waves (256 threads); each wave
// owns TM x TN tiles of 32x32 computed with v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD).
// LDS tiles are stored k-major ([k][row]) so a fragment read is one conflict-free ds_read_b32 per
// lane (lane l reads row l&31 of k-row 2s + (l>>5)); the fp32 MFMA takes 64 cycles, so LDS and
// VALU have a large budget and the kernels are MFMA-bound once the grid fills the chip.
// Global->LDS staging is register-staged and software-pipelined one K-tile ahead (loads for tile
// t+1 are issued before the MFMAs of tile t, written to the other LDS buffer after them).
#include "gconv.h"
#include <hip/hip_ext.h>

namespace siggan {


static constexpr int BK = 32;    // K-tile (floats)
static constexpr int PAD = 4;    // LDS row padding (floats)

// XCD-aware bijective remap (8 XCDs, blocks b and b+8 share an L2): consecutive logical tiles,
// which share an A row-panel, land on one XCD.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// KQ = 2: two wave quads per workgroup, each running the whole pipeline on its own half of the workgroup's K range and its
// own LDS buffers; the second quad's accumulators are added to the first's through LDS before the epilogue.  This is a
// two-way K split WITHOUT slabs or a second kernel (sum order quad 0 + quad 1 = slab 0 + slab 1 of the split-K protocol:
// bit-identical); used for the launches that would otherwise be split in two.
template <int BM, int BN, int WM, int WN, int BK = 32, int DEEP = 0, int KQ = 1>
__global__ __launch_bounds__(256 * KQ) void k_gconv(const GConvArgs a) {
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int PA = BM / 32, PB = BN / 32;
    constexpr int KC = BK / 32;        // 16-byte chunks a thread stages per row and K-tile
    // LDS tiles are [row][k] with a row stride of 36 floats (= 4 * odd): a staged float4 is ONE
    // ds_write_b128 (8 lanes cover a row's 128 B), and a lane's MFMA operands for 4 consecutive
    // sub-steps are ONE conflict-free ds_read_b128 (the 16 lanes of a b128 group hit 16 distinct
    // 16-byte slots because rows differ by 9 slots).  K is consumed in the permuted pairing
    // k = 8c + 4*(lane>>5) + t, identical for A and B, so only the fp32 summation order changes.
    constexpr int LD = BK + 4;
    __shared__ __attribute__((aligned(16))) float smem[KQ * 2 * LD * (BM + BN)];
    const int quad = KQ > 1 ? (int)(threadIdx.x >> 8) : 0;
    float* const sA = smem + quad * (2 * LD * (BM + BN));
    float* const sB = sA + 2 * LD * BM;

    const int tid = threadIdx.x & 255, lane = tid & 63, wave = tid >> 6;      // thread / wave index inside the quad
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const float* const a_in = static_cast<const float*>(a.in);
    const float* const a_wp = static_cast<const float*>(a.wp);

    const int tiles_n = a.Co / BN;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
    const int cls = blockIdx.z, ph = cls >> 1, pw = cls & 1;
    const int Hr = 1 << a.lgHr, Wr = 1 << a.lgWr;
    const int ntaps = a.form == 0 ? 16 : 4;
    const int Ktot = ntaps * a.Ci;
    const int lgcpt = 31 - __builtin_clz(a.Ci / BK);      // K-tiles per tap = Ci / 32, a power of two
    const int nk_all = ntaps << lgcpt;
    // split-K: this block (this quad of it) owns K-tiles [k_lo, k_hi); with KQ = 2 the launcher guarantees equal halves,
    // so both quads pass the same number of barriers
    const int kper = (nk_all + gridDim.y * KQ - 1) / (gridDim.y * KQ);
    const int k_lo = (blockIdx.y * KQ + quad) * kper;
    const int k_hi = min(nk_all, k_lo + kper);
    const int nk = k_hi - k_lo;

    // ---- per-thread staging coordinates: row rloc + 32p, 16-byte chunk kc of the 32-float k-row ------
    const int kc = tid & 7, rloc = tid >> 3;
    const float* a_base[PA];     // &in[n, 0, 0, kc*4] of the row's image
    int a_ih0[PA], a_iw0[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int m = m0 + rloc + 32 * p;
        if (m < a.M) {
            const int n = m >> (a.lgHr + a.lgWr);
            const int rh = (m >> a.lgWr) & (Hr - 1), rw = m & (Wr - 1);
            a_base[p] = a_in + (size_t)n * a.Hi * a.Wi * a.Ci + kc * 4;
            if (a.form == 0) { a_ih0[p] = 2 * rh - 1; a_iw0[p] = 2 * rw - 1; }
            else             { a_ih0[p] = rh + ph;    a_iw0[p] = rw + pw; }
        } else {
            a_base[p] = a_in; a_ih0[p] = -(1 << 20); a_iw0[p] = -(1 << 20);
        }
    }
    const float* wcur = a_wp + ((size_t)cls * a.Co + n0 + rloc) * Ktot + kc * 4 + (size_t)k_lo * BK;

    // load cursor: walks (tap, channel chunk) in K order; per-row source pointers are rebuilt only
    // when the tap changes, otherwise they advance by 32 floats (0 for out-of-image taps, which
    // read a 16-byte page of zeros so that the loads stay unconditional)
    const float* a_cur[PA];
    int a_step[PA];
    int l_tap = k_lo >> lgcpt, l_cc = k_lo & ((1 << lgcpt) - 1);
    auto set_tap = [&]() __attribute__((always_inline)) {
        int dh, dw;
        if (a.form == 0) { dh = l_tap >> 2; dw = l_tap & 3; } else { dh = -(l_tap >> 1); dw = -(l_tap & 1); }
#pragma unroll
        for (int p = 0; p < PA; ++p) {
            const int ih = a_ih0[p] + dh, iw = a_iw0[p] + dw;
            const bool ok = (unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi;
            a_cur[p] = ok ? a_base[p] + ((size_t)(ih * a.Wi + iw) * a.Ci + l_cc * BK) : a.zeros;
            a_step[p] = ok ? BK : 0;
        }
    };
    constexpr int NSET = DEEP ? 2 : 1;      // register sets: tile t waits in set t % NSET
    f32x4 ra[NSET][PA][KC], rb[NSET][PB][KC];
    auto load_tile = [&](int set = 0) __attribute__((always_inline)) {
#pragma unroll
        for (int p = 0; p < PA; ++p) {
#pragma unroll
            for (int q = 0; q < KC; ++q) ra[set][p][q] = *reinterpret_cast<const f32x4*>(a_cur[p] + 32 * q);   // (zero page: 256 B)
            a_cur[p] += a_step[p];
        }
#pragma unroll
        for (int p = 0; p < PB; ++p)
#pragma unroll
            for (int q = 0; q < KC; ++q) rb[set][p][q] = *reinterpret_cast<const f32x4*>(wcur + (size_t)(32 * p) * Ktot + 32 * q);
        wcur += BK;
        if (++l_cc == (1 << lgcpt)) { l_cc = 0; ++l_tap; set_tap(); }
    };
    auto store_tile = [&](int buf, int set = 0) __attribute__((always_inline)) {
        float* dA = sA + buf * LD * BM + rloc * LD + kc * 4;
        float* dB = sB + buf * LD * BN + rloc * LD + kc * 4;
#pragma unroll
        for (int p = 0; p < PA; ++p)
#pragma unroll
            for (int q = 0; q < KC; ++q) *reinterpret_cast<f32x4*>(dA + 32 * p * LD + 32 * q) = ra[set][p][q];
#pragma unroll
        for (int p = 0; p < PB; ++p)
#pragma unroll
            for (int q = 0; q < KC; ++q) *reinterpret_cast<f32x4*>(dB + 32 * p * LD + 32 * q) = rb[set][p][q];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // The register set always holds the NEXT tile: it is written to the other LDS buffer right
    // behind the first MFMAs of a tile and immediately re-loaded with the tile after that, so every
    // global load has a full K-tile of MFMA time to land (hipcc drains vmcnt to 0 at the LDS write
    // whatever is newer in flight, so the write sits where nothing newer is).
    if (DEEP && nk >= 5) {
        // Two tiles ahead: the loads of tile t are issued while tile t-3 is multiplied and are needed (LDS write) while tile
        // t-1 is -- two K-tiles of time to land instead of one.  A K-tile of this kernel lasts about as long as a load
        // that misses L2 takes, so with one tile of slack every tile waited for its slowest load.  Two register sets; the
        // loop is unrolled by two so that set and buffer indices are compile-time, and the steady part issues its loads
        // unconditionally: only then can hipcc count the loads in flight and wait for the OLDER set alone (vmcnt(4));
        // one conditional load anywhere on the path and it drains the queue (vmcnt(0)) at every LDS write.
        auto tile = [&](const int h, const int kt, const bool st, const bool ld) __attribute__((always_inline)) {
            const float* pA = sA + h * LD * BM + (wm * (32 * TM) + li) * LD + 4 * lh;
            const float* pB = sB + h * LD * BN + (wn * (32 * TN) + li) * LD + 4 * lh;
            f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(pA + 32 * i * LD);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(pB + 32 * j * LD);
#pragma unroll
            for (int c = 0; c < BK / 8; ++c) {
                const int cur = c & 1, nxt = cur ^ 1;
                if (c + 1 < BK / 8) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) fa[nxt][i] = *reinterpret_cast<const f32x4*>(pA + 32 * i * LD + 8 * (c + 1));
#pragma unroll
                    for (int j = 0; j < TN; ++j) fb[nxt][j] = *reinterpret_cast<const f32x4*>(pB + 32 * j * LD + 8 * (c + 1));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][t], fb[cur][j][t], acc[i][j], 0, 0, 0);
                if (c == 0 && st) store_tile(h ^ 1, h ^ 1);     // tile kt+1 (set (kt+1)&1): loaded two tiles ago
                if (c == 1 && ld) load_tile(h ^ 1);             // tile kt+3 into the set just written out
            }
            __syncthreads();
        };
        set_tap();
        load_tile(0);
        load_tile(1);
        store_tile(0, 0);
        load_tile(0);
        __syncthreads();
        int kt = 0;
        for (; kt + 4 < nk; kt += 2) {           // both halves have a tile kt+3 to load
            tile(0, kt, true, true);
            tile(1, kt + 1, true, true);
        }
        for (; kt < nk; kt += 2) {               // the last three or four tiles
            tile(0, kt, kt + 1 < nk, kt + 3 < nk);
            if (kt + 1 < nk) tile(1, kt + 1, kt + 2 < nk, kt + 4 < nk);
        }
    } else {
    if (nk > 0) {
        set_tap();
        load_tile();
        store_tile(0);
        if (nk > 1) load_tile();
    }
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const float* pA = sA + buf * LD * BM + (wm * (32 * TM) + li) * LD + 4 * lh;
        const float* pB = sB + buf * LD * BN + (wn * (32 * TN) + li) * LD + 4 * lh;
        f32x4 fa[2][TM], fb[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[0][i] = *reinterpret_cast<const f32x4*>(pA + 32 * i * LD);
#pragma unroll
        for (int j = 0; j < TN; ++j) fb[0][j] = *reinterpret_cast<const f32x4*>(pB + 32 * j * LD);
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            const int cur = c & 1, nxt = cur ^ 1;
            if (c + 1 < BK / 8) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[nxt][i] = *reinterpret_cast<const f32x4*>(pA + 32 * i * LD + 8 * (c + 1));
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[nxt][j] = *reinterpret_cast<const f32x4*>(pB + 32 * j * LD + 8 * (c + 1));
            }
            __builtin_amdgcn_sched_barrier(0);   // keep the next chunk's reads ahead of these MFMAs
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][t], fb[cur][j][t], acc[i][j], 0, 0, 0);
            if (c == 0 && kt + 1 < nk) store_tile(buf ^ 1);     // tile kt+1: loaded a tile ago
            if (c == 1 && kt + 2 < nk) load_tile();             // tile kt+2: lands during this tile
        }
        __syncthreads();
    }
    }

    // ---- epilogue ----------------------------------------------------------------------
    // The accumulators go through LDS once (the main loop's last barrier has passed: the staging buffers are free) so that the
    // fused arithmetic and the stores work on four consecutive channels per lane: a quarter of the store / aref-load
    // instructions, each covering a row's BN * 4 contiguous bytes.  Same values, same operations per element.
    float* const outp = gridDim.y > 1 ? a.slab + (size_t)blockIdx.y * a.slab_stride : static_cast<float*>(a.out);
    const int epi = gridDim.y > 1 ? (int)EPI_RAW : a.epi;
    constexpr int LDT = BN + 4;                      // 16-byte slots per row: odd multiple for BN = 32, 64, 128
    static_assert(BM * LDT <= 2 * LD * (BM + BN), "output tile must fit the staging buffers");
    float* const sT = smem;
    if (KQ > 1) {                                    // quad 1's accumulators -> LDS -> added to quad 0's (lane-for-lane)
        float* const red = smem + 2 * LD * (BM + BN);        // quad 1's own staging area
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
    if (quad == 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    sT[(wm * (32 * TM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh) * LDT + wn * (32 * TN) + 32 * j + li] = acc[i][j][r];
    }
    __syncthreads();
    if (quad != 0 && epi != EPI_BN_BWD_STATS) return;       // (with the statistics every wave stays for the barriers below)
    constexpr int C4 = BN / 4, RPP = 256 / C4;       // float4 columns per row, rows per pass
    const int c4 = tid % C4, r0 = tid / C4;
    const int co = n0 + c4 * 4;
    BnBwdParams bq;
    f32x4 st0 = {0.f, 0.f, 0.f, 0.f}, st1 = st0;
    if (epi == EPI_BN_BWD_STATS) bq = bn_bwd_params(a.bnp, a.Co, co);
    const f32x4 bias4 = epi == EPI_BIAS_LRELU_DROP ? *reinterpret_cast<const f32x4*>(a.bias + co) : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 sc4 = {1.f, 1.f, 1.f, 1.f}, sh4 = {0.f, 0.f, 0.f, 0.f};
    if (epi == EPI_AFFINE_RELU) { sc4 = *reinterpret_cast<const f32x4*>(a.scale + co); sh4 = *reinterpret_cast<const f32x4*>(a.shift + co); }
    const bool use_noise = (epi == EPI_BIAS_LRELU_DROP || epi == EPI_LRELU_BWD) && a.noise != nullptr;
#pragma unroll
    for (int p = 0; p < BM / RPP; ++p) {
        const int row = r0 + RPP * p, m = m0 + row;
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
            bn_bwd_stat_terms<float>(v, *reinterpret_cast<const f32x4*>(static_cast<const float*>(a.aref) + o), bq, st0, st1);
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
            const f32x4 ar = *reinterpret_cast<const f32x4*>(static_cast<const float*>(a.aref) + o);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= ar[e] > 0.f ? 1.f : a.slope;
            if (use_noise) {
                const f32x4 nz = *reinterpret_cast<const f32x4*>(a.noise + (size_t)n * a.Co + co);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] *= nz[e];
            }
        }
        *reinterpret_cast<f32x4*>(outp + o) = v;
    }
    if (epi == EPI_BN_BWD_STATS) {
        // the tile's column sums: the RPP row lanes of a channel group meet in LDS (the tile is stored: sT is free after the
        // barrier) and are added in lane order -- one partial row per workgroup, a fixed sum order
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

// ------------------------------------------------------------------------------------------
// "up" form with all four output-parity classes in one workgroup, for the short-K layers with 32 output channels
// (Generator blocks 3-4: K = 4 taps x 32-64 channels per class, i.e. 4-8 K-tiles -- the per-class kernel above spends
// most of its time in prologue and epilogue there).  The four classes of an input pixel read a 3x3 neighbourhood
// between them, each a 2x2 corner of it: input offset (oh, ow) in {-1,0,1}^2 feeds the classes with
// ph in P(oh), pw in P(ow), P(-1) = {0}, P(0) = {0,1}, P(1) = {1}, through tap th = ph - oh, tw = pw - ow.
// One workgroup = 128 consecutive input pixels of one image x 32 channels x 4 classes.  Its whole input -- the pixel
// rows it covers plus a one-pixel halo, all CI channels -- is brought into LDS ONCE (one burst of independent loads,
// zero outside the image); the nine taps are then shifted views of that patch, so the K loop stages only the small
// weight tiles.  Four accumulators per wave; the 2x2 output pixels of the 128 input pixels form one contiguous 64 KB
// range of the NHWC output, assembled in LDS and stored with 16-byte lanes.
// ------------------------------------------------------------------------------------------
template <int CI>
__global__ __launch_bounds__(256) void k_gconv_up4(const GConvArgs a) {
    constexpr int BM = 128, BN = 32, LD = BK + 4, LP = CI + 4, NCC = CI / BK;
    constexpr int PATCH_PIX = 264;                   // (128 / Wr + 2) * (Wr + 2) for Wr = 16, 32, 64: 180, 204, 264
    constexpr int SMEM = (PATCH_PIX * LP + 2 * 4 * BN * LD) > (BM * 4 * BN) ? (PATCH_PIX * LP + 2 * 4 * BN * LD) : (BM * 4 * BN);
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float* const sP = smem;                          // [patch pixel][LP]
    float* const sB = smem + PATCH_PIX * LP;         // [2][4 classes][BN][LD]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = xcd_remap(blockIdx.x, gridDim.x) * BM;
    const int Hr = 1 << a.lgHr, Wr = 1 << a.lgWr, Wp = Wr + 2;
    const int Ktot = 4 * CI;
    const int n_img = m0 >> (a.lgHr + a.lgWr), rh0 = (m0 >> a.lgWr) & (Hr - 1), rows = BM >> a.lgWr;

    // ---- the input patch: rows rh0-1 .. rh0+rows, columns -1 .. Wr, every channel ----------------------------
    {
        const float* const img = static_cast<const float*>(a.in) + (size_t)n_img * a.Hi * a.Wi * CI;
        const int nchunk = (rows + 2) * Wp * (CI / 4);
        constexpr int NI = (PATCH_PIX * (CI / 4) + 255) / 256;      // loads per thread, all issued before the first LDS write
        f32x4 v[NI];
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int i = tid + 256 * q;
            const int pix = i / (CI / 4), ch = i - pix * (CI / 4);
            const int pr = pix / Wp, pc = pix - pr * Wp;
            const int ih = rh0 - 1 + pr, iw = pc - 1;
            const bool ok = i < nchunk && (unsigned)ih < (unsigned)a.Hi && (unsigned)iw < (unsigned)a.Wi;
            const float* src = ok ? img + ((size_t)ih * a.Wi + iw) * CI + ch * 4 : a.zeros;
            v[q] = *reinterpret_cast<const f32x4*>(src);
        }
#pragma unroll
        for (int q = 0; q < NI; ++q) {
            const int i = tid + 256 * q;
            if (i < nchunk) {
                const int pix = i / (CI / 4), ch = i - pix * (CI / 4);
                *reinterpret_cast<f32x4*>(sP + pix * LP + ch * 4) = v[q];
            }
        }
    }
    // ---- weight tiles: K-tile kt = (class, channel chunk); its four taps' [32 co][32 ci] tiles are staged together by
    // thread (rloc = output channel, kc = 16-byte chunk), one tile ahead of the MFMAs ----------------------------------
    const int kc = tid & 7, rloc = tid >> 3;
    const float* const w_row = static_cast<const float*>(a.wp) + (size_t)rloc * Ktot + kc * 4;
    f32x4 rb[4];
    int l_k = 0;
    auto load_tile = [&]() __attribute__((always_inline)) {
        const int cls = l_k / NCC, cc = l_k - cls * NCC;
#pragma unroll
        for (int t = 0; t < 4; ++t)
            rb[t] = *reinterpret_cast<const f32x4*>(w_row + (size_t)cls * a.Co * Ktot + t * CI + cc * BK);
        ++l_k;
    };
    auto store_tile = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
            *reinterpret_cast<f32x4*>(sB + ((buf * 4 + t) * BN + rloc) * LD + kc * 4) = rb[t];
    };

    f32x16 acc[4];
#pragma unroll
    for (int cls = 0; cls < 4; ++cls)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cls][r] = 0.f;

    load_tile();
    store_tile(0);
    load_tile();
    __syncthreads();

    // this lane's pixel (A-operand row) inside the patch
    const int ml = wave * 32 + li;
    const float* const pP = sP + (((ml >> a.lgWr) + 1) * Wp + (ml & (Wr - 1)) + 1) * LP + 4 * lh;

#pragma unroll
    for (int cls = 0; cls < 4; ++cls) {
        const int ph = cls >> 1, pw = cls & 1;
        for (int cc = 0; cc < NCC; ++cc) {
            const int kt = cls * NCC + cc, buf = kt & 1;
            // tap t = (th, tw) reads the patch at input offset (ph - th, pw - tw)
            const float* pA[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) pA[t] = pP + ((ph - (t >> 1)) * Wp + (pw - (t & 1))) * LP + cc * BK;
            const float* pB = sB + (buf * 4 * BN + li) * LD + 4 * lh;
            f32x4 fa[2][4], fb[2][4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[0][t] = *reinterpret_cast<const f32x4*>(pA[t]);
                fb[0][t] = *reinterpret_cast<const f32x4*>(pB + t * BN * LD);
            }
#pragma unroll
            for (int c = 0; c < BK / 8; ++c) {
                const int cur = c & 1, nxt = cur ^ 1;
                if (c + 1 < BK / 8) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        fa[nxt][t] = *reinterpret_cast<const f32x4*>(pA[t] + 8 * (c + 1));
                        fb[nxt][t] = *reinterpret_cast<const f32x4*>(pB + t * BN * LD + 8 * (c + 1));
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        acc[cls] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][t][u], fb[cur][t][u], acc[cls], 0, 0, 0);
                if (c == 0 && kt + 1 < 4 * NCC) store_tile(buf ^ 1);
                if (c == 1 && kt + 2 < 4 * NCC) load_tile();
            }
            __syncthreads();
        }
    }

    // ---- epilogue: assemble the contiguous output range in LDS in memory order, store with 16-byte lanes ---------
    const float sc = a.epi == EPI_AFFINE_RELU ? a.scale[li] : 1.f, sf = a.epi == EPI_AFFINE_RELU ? a.shift[li] : 0.f;
    const int Wo = 2 * Wr;
    float* const so = smem;                              // 128 * 4 * 32 floats; the main loop's last barrier has passed
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mr = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int rl = mr >> a.lgWr, rw = mr & (Wr - 1);
#pragma unroll
        for (int cls = 0; cls < 4; ++cls) {
            float v = acc[cls][r];
            if (a.epi == EPI_AFFINE_RELU) v = fmaxf(fmaf(v, sc, sf), 0.f);
            so[(((2 * rl + (cls >> 1)) * Wo) + 2 * rw + (cls & 1)) * BN + li] = v;
        }
    }
    __syncthreads();
    f32x4* const dst = reinterpret_cast<f32x4*>(static_cast<float*>(a.out) + ((size_t)n_img * a.Ho + 2 * rh0) * a.Wo * BN);
    const f32x4* const src = reinterpret_cast<const f32x4*>(so);
#pragma unroll
    for (int q = 0; q < (BM * 4 * BN / 4) / 256; ++q) dst[q * 256 + tid] = src[q * 256 + tid];
}

// split-K tail: out = epilogue(sum_z slab[z]) over the NHWC output (4 channels per thread)
template <class T>
__global__ __launch_bounds__(256) void k_splitk_epilogue(const GConvArgs a, int nsplit, int64_t total4) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (a.epi == EPI_BN_BWD_STATS) {
        // 256 consecutive float4 = 256 / C4 rows of all C4 channel groups (launch_gconv checks C4 | 256): the row lanes of a
        // group meet in LDS, one partial row per workgroup, a fixed sum order
        __shared__ __attribute__((aligned(16))) float sh[256 * 8];
        const int C4 = a.Co / 4, c4 = threadIdx.x & (C4 - 1);
        f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
        if (i < total4) {
            const f32x4* sl = reinterpret_cast<const f32x4*>(a.slab);
            const size_t stride4 = a.slab_stride / 4;
            f32x4 v = sl[i];
            for (int z = 1; z < nsplit; ++z) v += sl[(size_t)z * stride4 + i];
            bn_bwd_stat_terms<T>(v, ld4<T>(static_cast<const T*>(a.aref) + i * 4), bn_bwd_params(a.bnp, a.Co, c4 * 4), s0, s1);
            st4<T>(static_cast<T*>(a.out) + i * 4, v);
        }
        *reinterpret_cast<f32x4*>(sh + threadIdx.x * 8) = s0;
        *reinterpret_cast<f32x4*>(sh + threadIdx.x * 8 + 4) = s1;
        __syncthreads();
        if ((int)threadIdx.x < C4) {
            for (int k = 1; k < 256 / C4; ++k) {
                s0 += *reinterpret_cast<const f32x4*>(sh + (k * C4 + c4) * 8);
                s1 += *reinterpret_cast<const f32x4*>(sh + (k * C4 + c4) * 8 + 4);
            }
            *reinterpret_cast<f32x4*>(a.stat0 + (size_t)blockIdx.x * a.Co + c4 * 4) = s0;
            *reinterpret_cast<f32x4*>(a.stat1 + (size_t)blockIdx.x * a.Co + c4 * 4) = s1;
        }
        return;
    }
    if (i >= total4) return;
    const float4* sl = reinterpret_cast<const float4*>(a.slab);
    const size_t stride4 = a.slab_stride / 4;
    float4 v = sl[i];
    for (int z = 1; z < nsplit; ++z) {
        const float4 w = sl[(size_t)z * stride4 + i];
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
    }
    const int C4 = a.Co / 4;
    const int64_t per_img = (int64_t)C4 * a.Ho * a.Wo;
    const bool p2 = (C4 & (C4 - 1)) == 0 && (per_img & (per_img - 1)) == 0;       // every layer: shifts instead of 64-bit divisions
    const int c = (p2 ? (int)(i & (C4 - 1)) : (int)(i % C4)) * 4;
    const int64_t n = p2 ? i >> (63 - __builtin_clzll(per_img)) : i / per_img;
    float* e = reinterpret_cast<float*>(&v);
    if (a.epi == EPI_BIAS_LRELU_DROP) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float x = e[k] + a.bias[c + k];
            x = x > 0.f ? x : x * a.slope;
            if (a.noise) x *= a.noise[n * a.Co + c + k];
            e[k] = x;
        }
    } else if (a.epi == EPI_AFFINE_RELU) {
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = fmaxf(fmaf(e[k], a.scale[c + k], a.shift[c + k]), 0.f);
    } else if (a.epi == EPI_LRELU_BWD) {
        const f32x4 ar = ld4<T>(static_cast<const T*>(a.aref) + i * 4);
        const float* r = reinterpret_cast<const float*>(&ar);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float x = e[k] * (r[k] > 0.f ? 1.f : a.slope);
            if (a.noise) x *= a.noise[n * a.Co + c + k];
            e[k] = x;
        }
    }
    st4<T>(static_cast<T*>(a.out) + i * 4, f32x4{v.x, v.y, v.z, v.w});
    if (a.cls_w) {
        // the classifier's dot product over the values just stored (as stored: rounded to T), 1024 of them per workgroup; every
        // thread is here (launch_gconv: total4 % 256 == 0, per_img % 256 == 0)
        __shared__ float shc[4];
        const f32x4 w4 = *reinterpret_cast<const f32x4*>(a.cls_w + (size_t)(p2 ? (i & (per_img - 1)) : (i % per_img)) * 4);
        float d = (float)(T)v.x * w4[0];
        d = fmaf((float)(T)v.y, w4[1], d); d = fmaf((float)(T)v.z, w4[2], d); d = fmaf((float)(T)v.w, w4[3], d);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) d += __shfl_down(d, o, 64);
        if ((threadIdx.x & 63) == 0) shc[threadIdx.x >> 6] = d;
        __syncthreads();
        if (threadIdx.x == 0) a.cls_part[blockIdx.x] = ((shc[0] + shc[1]) + shc[2]) + shc[3];
    }
}

Prof* g_prof = nullptr;

const char* Prof::name(int id) {
    static const char* n[NID] = {"k_gconv<128,128>", "k_gconv<128,64>", "k_gconv<64,64>", "k_gconv<128,32>",
                                 "k_wgrad<64,64>", "k_gconv_up4", "k_wgrad<32,128>", "?"};
    return n[id < 0 || id >= NID ? NID - 1 : id];
}
// The two events of a record are handed to hipExtLaunchKernelGGL, which stamps them with the
// kernel's own begin / end (the dispatch's completion-signal timestamps), so a record's elapsed time
// is the kernel duration rocprofv3 reports -- not the launch-to-launch gap a plain
// hipEventRecord pair would include.
void Prof::begin(int id, double flops, double bytes, hipStream_t) {
    Rec r; r.id = id; r.flops = flops; r.bytes = bytes;
    (void)hipEventCreate(&r.e0); (void)hipEventCreate(&r.e1);
    recs.push_back(r);
}
void Prof::clear() {
    for (auto& r : recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    recs.clear();
}

// algorithmic HBM bytes of one implicit-GEMM launch: input tensor and packed weights read once, output written once
// (+ the stored activation the fused LeakyReLU-backward epilogue reads); split-K slabs and tile re-reads are not counted
static double gconv_bytes(const GConvArgs& a) {
    const double es = (double)dt_size(a.dt);
    const double in = (double)a.B * a.Hi * a.Wi * a.Ci, out = (double)a.B * a.Ho * a.Wo * a.Co, w = 16.0 * a.Ci * a.Co;
    return es * (in + w + out * (a.epi == EPI_LRELU_BWD ? 2.0 : 1.0));
}
static double wgrad_bytes(const WgradArgs& a) {
    const double es = (double)dt_size(a.dt);
    return es * ((double)a.K * a.Cs + 4.0 * a.K * a.Cl) + 4.0 * 16.0 * a.Cs * a.Cl;
}

template <int BM, int BN, int WM, int WN, int BKT = 32, int DEEP = 0, int KQ = 1>
static int launch_cfg(const GConvArgs& a_in, hipStream_t st, int id, int nsplit) {
    GConvArgs a = a_in;
    const int tiles = ((a.M + BM - 1) / BM) * (a.Co / BN);
    const int ncls = a.form == 0 ? 1 : 4;
    const int64_t total4 = (int64_t)a.B * a.Ho * a.Wo * a.Co / 4;
    int rows = 0;
    if (a.epi == EPI_BN_BWD_STATS) {
        // one partial row per workgroup of whichever kernel runs the epilogue; k_splitk_epilogue needs whole rows per workgroup
        if (nsplit > 1 && (256 % (a.Co / 4)) != 0) nsplit = 1;
        rows = nsplit > 1 ? (int)((total4 + 255) / 256) : ((a.M + BM - 1) / BM) * ncls;
        if ((int64_t)2 * rows * a.Co > a.stat_cap) { a.epi = EPI_RAW; rows = 0; }     // (a batch whose partial rows outgrow the carve)
        else a.stat1 = a.stat0 + (size_t)rows * a.Co;
    }
    if (a.cls_w) {
        const int64_t per_img4 = (int64_t)a.Ho * a.Wo * a.Co / 4;
        if (nsplit > 1 && a.epi != EPI_BN_BWD_STATS && total4 % 256 == 0 && per_img4 % 256 == 0) rows = (int)(per_img4 / 256);
        else a.cls_w = nullptr;
    }
    dim3 grid(tiles, nsplit, ncls);
    // algorithmic FLOPs = 2 * M * Co * (taps * Ci) per class (== 2 * conv MACs, padding taps included)
    if (g_prof) g_prof->begin(id, 2.0 * a.M * a.Co * (double)((a.form == 0 ? 16 : 4) * a.Ci) * ncls, gconv_bytes(a), st);
    hipEvent_t e0 = g_prof ? g_prof->recs.back().e0 : nullptr, e1 = g_prof ? g_prof->recs.back().e1 : nullptr;
    if (a.dt != DT_F32) {
        launch_gconv16(id, a, grid, st, e0, e1, KQ);
    } else if (g_prof) {
        hipExtLaunchKernelGGL((k_gconv<BM, BN, WM, WN, BKT, DEEP, KQ>), grid, dim3(256 * KQ), 0, st, e0, e1, 0, a);
    } else {
        hipLaunchKernelGGL((k_gconv<BM, BN, WM, WN, BKT, DEEP, KQ>), grid, dim3(256 * KQ), 0, st, a);
    }
    if (nsplit > 1)
        SIGGAN_DT_SWITCH(a.dt, T, hipLaunchKernelGGL(k_splitk_epilogue<T>, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, a, nsplit, total4));
    return rows;
}

int launch_gconv(const GConvArgs& a_in, hipStream_t st) {
    // Tile choice: the fp32 MFMA is slow enough that a 32x32 accumulator per wave already runs the
    // matrix pipe at full rate, so what matters is having >= 2 workgroups per CU (512 on the chip)
    // to cover each other's barrier / first-fragment bubbles.  Take the largest tile that still
    // gives that; when even 64x64 tiles cannot, split K into slabs (summed by k_splitk_epilogue).
    GConvArgs a = a_in;
    const int ncls = a.form == 0 ? 1 : 4;
    auto blocks = [&](int bm, int bn) { return ((a.M + bm - 1) / bm) * (a.Co / bn) * ncls; };
    const int nk = (a.form == 0 ? 16 : 4) * (a.Ci / BK);
    auto splits = [&](int nblk) {
        int ns = 1;
        if (a.slab && nblk < 384) {
            ns = (512 + nblk - 1) / nblk;
            if (ns > nk / 4) ns = nk / 4;
            if (ns > 8) ns = 8;
            const int64_t out_floats = (int64_t)a.B * a.Ho * a.Wo * a.Co;
            if (ns > 1 && (int64_t)ns * out_floats > a.slab_floats) ns = (int)(a.slab_floats / out_floats);
            if (ns < 1) ns = 1;
        }
        a.slab_stride = (size_t)a.B * a.Ho * a.Wo * a.Co;
        return ns;
    };
    if (a.Co >= 64) {
        // 128x128 (four accumulators per wave, 82 % MFMA-busy) only when it still yields two workgroups per CU; the
        // 128x64 middle size measured below 64x64 on every shape of the step (86 vs 93 TFLOP/s) and is not built
        if (a.Co >= 128 && blocks(128, 128) >= 512) return launch_cfg<128, 128, 2, 2, 32, 1>(a, st, 0, 1);
        const int ns = splits(blocks(64, 64));
        // a two-way split runs as ONE launch of 8-wave workgroups (no slabs, no k_splitk_epilogue); nk is a power of two,
        // so the halves are equal
        if (ns == 2 && (nk & 1) == 0) return launch_cfg<64, 64, 2, 2, 32, 1, 2>(a, st, 2, 1);
        // 16-bit: a four-way split too (unless the classifier rides in the slab epilogue) -- the K loop is twice as long but
        // short either way, the epilogue launch is what counts: bf16 step 0.6172 -> 0.6077 ms (fp32: step unchanged, the
        // Generator's eval forward 87.7 -> 90.4 us: stays split)
        if (ns == 4 && !a.cls_w && a.dt != DT_F32 && (nk & 1) == 0) return launch_cfg<64, 64, 2, 2, 32, 1, 2>(a, st, 2, 1);
        return launch_cfg<64, 64, 2, 2, 32, 1>(a, st, 2, ns);
    }
    if (a.dt == DT_F32 && a.form == 1 && a.Co == 32 && (a.epi == EPI_RAW || a.epi == EPI_AFFINE_RELU) && (a.Ci == 32 || a.Ci == 64) &&
        a.M / 128 >= (a.Ci == 32 ? 384 : 768) && a.M % 128 == 0 &&
        ((1 << (a.lgHr + a.lgWr)) % 128) == 0 && a.lgWr >= 4 && a.lgWr <= 6) {
        // all four parity classes per workgroup, input patch resident in LDS (k_gconv_up4): the short-K Generator blocks
        dim3 grid(a.M / 128);
        if (g_prof) g_prof->begin(5, 2.0 * a.M * a.Co * (double)(4 * a.Ci) * 4, gconv_bytes(a), st);
        hipEvent_t e0 = g_prof ? g_prof->recs.back().e0 : nullptr, e1 = g_prof ? g_prof->recs.back().e1 : nullptr;
        if (a.Ci == 32) hipExtLaunchKernelGGL(k_gconv_up4<32>, grid, dim3(256), 0, st, e0, e1, 0, a);
        else hipExtLaunchKernelGGL(k_gconv_up4<64>, grid, dim3(256), 0, st, e0, e1, 0, a);
        return 0;
    }
    const int ns = splits(blocks(128, 32));
    return launch_cfg<128, 32, 4, 1, 32, 1>(a, st, 3, ns);   // Co == 32
}

// ------------------------------------------------------------------------------------------
// weight gradient: C[i][j] = sum_pix S[pix][i] * Lg[pix][j],  j = tap*Cl + l
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void k_wgrad(const WgradArgs a) {
    constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);
    constexpr int LDA = BM + PAD, LDB = BN + PAD;
    constexpr int CA = BM / 4, RA = 256 / CA, PA = BK / RA;     // float4 chunks per k-row, rows per pass
    constexpr int CB = BN / 4, RB = 256 / CB, PB = BK / RB;
    __shared__ __attribute__((aligned(16))) float smem[2 * BK * (LDA + LDB)];
    float* const sA = smem;
    float* const sB = smem + 2 * BK * LDA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 31, lh = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int N = 16 << a.lgCl;
    const int tiles_n = N / BN;
    // Workgroups are handed out round-robin over the 8 XCDs; remapped, a run of consecutive logical ids -- the tiles of ONE K
    // split, which share its S rows and its gathered L pixels -- lands on one XCD and meets in that L2: 84.4 -> 40.1 MB of HBM
    // traffic per launch (32x128: 108 -> 51.5; the step 2.60 -> 2.28 GB).  The kernel's own time does not move (50.6 us: its
    // loads are not what it waits for, round 2 found the same), the pipelined step gains 0.2-1 %: the bytes it no longer
    // moves are there for the kernels beside it.
    const int lin = blockIdx.z * gridDim.x + blockIdx.x;
    const int lid = xcd_remap(lin, gridDim.x * gridDim.z);
    const int bx = lid % gridDim.x, bz = lid / gridDim.x;
    const int i0 = (bx / tiles_n) * BM, j0 = (bx % tiles_n) * BN;
    const int kbeg = bz * a.kchunk;
    const int kend = min(a.K, kbeg + a.kchunk);
    const int nk = (kend - kbeg + BK - 1) / BK;
    const int Hs = 1 << a.lgHs, Ws = 1 << a.lgWs, Hl = 2 * Hs, Wl = 2 * Ws, Cl = 1 << a.lgCl;

    const int ca = tid % CA, ka = tid / CA;
    const int cb = tid % CB, kb = tid / CB;
    const float* const a_S = static_cast<const float*>(a.S);
    const float* const a_L = static_cast<const float*>(a.L);
    const int jj = j0 + cb * 4, tap = jj >> a.lgCl, lch = jj & (Cl - 1);
    const int kh = tap >> 2, kw = tap & 3;

    f32x4 ra[2][PA], rb[2][PB];          // two register sets: tile t waits in set t & 1 (second set: the deep path only)
    // Tiles are loaded strictly in K order, so the loader keeps running state instead of deriving addresses from the tile
    // index: the S pointer advances by a constant, and the gathered L address is a handful of shifts (every extent is a power
    // of two) off the running pixel index.  The first version recomputed both from scratch with 64-bit multiplies: 1100 cycles
    // of VALU work per K-tile, which the matrix pipe does not hide (a wave's VALU instructions queue behind its own MFMAs).
    // (offsets from the kernel-argument pointers, not running pointers: a loop-carried pointer loses its address space and
    // hipcc falls back to flat loads, whose counter semantics also cost the exact vmcnt waits of the deep path)
    size_t a_off[PA];
    int a_pix[PA], b_pix[PB];
#pragma unroll
    for (int p = 0; p < PA; ++p) { a_pix[p] = kbeg + ka + RA * p; a_off[p] = (size_t)a_pix[p] * a.Cs + i0 + ca * 4; }
#pragma unroll
    for (int p = 0; p < PB; ++p) b_pix[p] = kbeg + kb + RB * p;
    const int a_adv = BK * a.Cs;
    const int lgWl = a.lgWs + 1;
    // (macros, not lambdas: hipcc left the staged float4 arrays in scratch when these were lambdas)
#define WG_LOAD_TILE(KT, SET)                                                                         \
    {                                                                                                 \
        _Pragma("unroll") for (int p = 0; p < PA; ++p) {                                              \
            const float* src = a_pix[p] < kend ? a_S + a_off[p] : a.zeros;                            \
            ra[SET][p] = *reinterpret_cast<const f32x4*>(src);                                       \
            a_off[p] += a_adv; a_pix[p] += BK;                                                        \
        }                                                                                             \
        _Pragma("unroll") for (int p = 0; p < PB; ++p) {                                              \
            const int pix = b_pix[p];                                                                 \
            const int y = pix >> a.lgWs;                       /* n * Hs + py */                       \
            const int ih = 2 * (y & (Hs - 1)) - 1 + kh, iw = 2 * (pix & (Ws - 1)) - 1 + kw;           \
            const bool ok = pix < kend && (unsigned)ih < (unsigned)Hl && (unsigned)iw < (unsigned)Wl; \
            /* n * Hl + ih = 2 * y - 1 + kh: the large-image row index is linear in y */              \
            const unsigned off = (((unsigned)(2 * y - 1 + kh) << lgWl) + (unsigned)iw) << a.lgCl;     \
            const float* src = ok ? a_L + ((size_t)off + lch) : a.zeros;                              \
            rb[SET][p] = *reinterpret_cast<const f32x4*>(src);                                       \
            b_pix[p] += BK;                                                                           \
        }                                                                                             \
    }
#define WG_STORE_TILE(BUF, SET)                                                                       \
    {                                                                                                 \
        _Pragma("unroll") for (int p = 0; p < PA; ++p)                                                \
            *reinterpret_cast<f32x4*>(sA + (BUF) * BK * LDA + (ka + RA * p) * LDA + ca * 4) = ra[SET][p]; \
        _Pragma("unroll") for (int p = 0; p < PB; ++p)                                                \
            *reinterpret_cast<f32x4*>(sB + (BUF) * BK * LDB + (kb + RB * p) * LDB + cb * 4) = rb[SET][p]; \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // column sums of S (the bias gradient when S is d(pre-activation)) ride along in the first column tile: thread t
    // owns column t % BM and the k rows [kq * KQ, kq * KQ + KQ) of every K-tile, the slices are added at the end
    constexpr int NQ = 256 / BM, KQ = BK / NQ;
    const bool bias_blk = a.db != nullptr && j0 == 0;
    const int bcol = tid % BM, kq = tid / BM;
    float bsum = 0.f;
    // one K-tile out of LDS buffer BUF; ST / LD: an LDS write / a global load of a later tile rides along (see below)
#define WG_TILE(BUF, ST, LD)                                                                          \
    {                                                                                                 \
        if (bias_blk) {                                                                               \
            const float* col = sA + (BUF) * BK * LDA + kq * KQ * LDA + bcol;                          \
            _Pragma("unroll") for (int k = 0; k < KQ; ++k) bsum += col[k * LDA];                      \
        }                                                                                             \
        const float* pA = sA + (BUF) * BK * LDA + lh * LDA + wm * (32 * TM) + li;                     \
        const float* pB = sB + (BUF) * BK * LDB + lh * LDB + wn * (32 * TN) + li;                     \
        /* fragments are fetched a chunk of four k-steps ahead of the MFMAs that use them (256 cycles of matrix-pipe time to   \
           cover the LDS latency; one step ahead -- 64 cycles -- left every MFMA waiting for its operands) */                  \
        float fa[2][4][TM], fb[2][4][TN];                                                             \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                               \
            _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[0][q][i] = pA[(2 * q) * LDA + 32 * i];  \
            _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[0][q][j] = pB[(2 * q) * LDB + 32 * j];  \
        }                                                                                             \
        _Pragma("unroll") for (int c = 0; c < BK / 8; ++c) {                                          \
            const int cur = c & 1, nxt = cur ^ 1;                                                     \
            if (c + 1 < BK / 8) {                                                                     \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                       \
                    _Pragma("unroll") for (int i = 0; i < TM; ++i) fa[nxt][q][i] = pA[(8 * c + 8 + 2 * q) * LDA + 32 * i]; \
                    _Pragma("unroll") for (int j = 0; j < TN; ++j) fb[nxt][q][j] = pB[(8 * c + 8 + 2 * q) * LDB + 32 * j]; \
                }                                                                                     \
            }                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                        \
            _Pragma("unroll") for (int q = 0; q < 4; ++q)                                             \
                _Pragma("unroll") for (int i = 0; i < TM; ++i)                                        \
                    _Pragma("unroll") for (int j = 0; j < TN; ++j)                                    \
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][q][i], fb[cur][q][j], acc[i][j], 0, 0, 0); \
            if (c == 0) { ST }                                                                        \
            if (c == 1) { LD }                                                                        \
        }                                                                                             \
        __syncthreads();                                                                              \
    }
    if (nk >= 5) {
        // two tiles ahead (see k_gconv): tile t is loaded while tile t-3 is multiplied and written to LDS while tile t-1 is;
        // the steady loop issues its loads unconditionally so that the LDS write waits for the older register set only
        WG_LOAD_TILE(0, 0)
        WG_LOAD_TILE(1, 1)
        WG_STORE_TILE(0, 0)
        WG_LOAD_TILE(2, 0)
        __syncthreads();
        int kt = 0;
        for (; kt + 4 < nk; kt += 2) {
            WG_TILE(0, WG_STORE_TILE(1, 1), WG_LOAD_TILE(kt + 3, 1))
            WG_TILE(1, WG_STORE_TILE(0, 0), WG_LOAD_TILE(kt + 4, 0))
        }
        for (; kt < nk; kt += 2) {
            WG_TILE(0, if (kt + 1 < nk) WG_STORE_TILE(1, 1), if (kt + 3 < nk) WG_LOAD_TILE(kt + 3, 1))
            if (kt + 1 < nk) WG_TILE(1, if (kt + 2 < nk) WG_STORE_TILE(0, 0), if (kt + 4 < nk) WG_LOAD_TILE(kt + 4, 0))
        }
    } else {
        if (nk > 0) {
            WG_LOAD_TILE(0, 0)
            WG_STORE_TILE(0, 0)
            if (nk > 1) WG_LOAD_TILE(1, 0)
        }
        __syncthreads();
        for (int kt = 0; kt < nk; ++kt) {
            if (kt & 1) WG_TILE(1, if (kt + 1 < nk) WG_STORE_TILE(0, 0), if (kt + 2 < nk) WG_LOAD_TILE(kt + 2, 0))
            else WG_TILE(0, if (kt + 1 < nk) WG_STORE_TILE(1, 0), if (kt + 2 < nk) WG_LOAD_TILE(kt + 2, 0))
        }
    }
#undef WG_TILE

#undef WG_LOAD_TILE
#undef WG_STORE_TILE
    if (bias_blk) {                                   // the main loop's last barrier has passed: LDS is free
        smem[kq * BM + bcol] = bsum;
        __syncthreads();
        if (tid < BM) {
            float t = smem[tid];
#pragma unroll
            for (int q = 1; q < NQ; ++q) t += smem[q * BM + tid];
            if (gridDim.z == 1) a.db[i0 + tid] = t;
            else a.slab[(size_t)gridDim.z * a.Cs * N + (size_t)bz * a.Cs + i0 + tid] = t;
        }
    }
    if (gridDim.z == 1 && a.dw) {
        // a single split: un-permute straight into the torch layout, column j = tap*Cl + l -> dw[(row*Cl + l)*16 + tap]
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = i0 + wm * (32 * TM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = j0 + wn * (32 * TN) + 32 * j + li;
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
            const int row = i0 + wm * (32 * TM) + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lh;
#pragma unroll
            for (int j = 0; j < TN; ++j)
                out[(size_t)row * N + j0 + wn * (32 * TN) + 32 * j + li] = acc[i][j][r];
        }
}

static void launch_wgrad_reduce(const float* slab, float* dw, float* db, int nsplit, int Cs, int Cl, hipStream_t st);

int launch_wgrad(WgradArgs a, int max_splits, hipStream_t st) {
    // 64x64 tiles (one 32x32 accumulator per wave keeps the fp32 MFMA pipe full): many tiles,
    // hence few K splits and little slab traffic; Cs == 32 uses a 32x128 tile.
    // (measured and dropped in round 3, DESIGN 4: 64x128 tiles; three / four workgroups per CU through more K splits)
    const int N = 16 << a.lgCl;
    const bool small = a.Cs < 64;
    const int tiles = small ? (N / 128) : (a.Cs / 64) * (N / 64);
    int ktiles = (a.K + BK - 1) / BK;
    int nsplit = (512 + tiles - 1) / tiles;                  // ~2 workgroups per CU
    if (nsplit > ktiles / 4) nsplit = ktiles / 4;            // at least 4 K-tiles per split
    if (nsplit > max_splits) nsplit = max_splits;
    if (nsplit < 1) nsplit = 1;
    int per = (ktiles + nsplit - 1) / nsplit;
    a.kchunk = per * BK;
    nsplit = (ktiles + per - 1) / per;
    dim3 grid(tiles, 1, nsplit);
    if (g_prof) g_prof->begin(small ? 6 : 4, 2.0 * a.Cs * (double)N * a.K, wgrad_bytes(a), st);
    hipEvent_t e0 = g_prof ? g_prof->recs.back().e0 : nullptr, e1 = g_prof ? g_prof->recs.back().e1 : nullptr;
    if (a.dt != DT_F32) {
        launch_wgrad16(small, a, grid, st, e0, e1);
    } else if (g_prof) {
        if (small) hipExtLaunchKernelGGL((k_wgrad<32, 128, 1, 4>), grid, dim3(256), 0, st, e0, e1, 0, a);
        else hipExtLaunchKernelGGL((k_wgrad<64, 64, 2, 2>), grid, dim3(256), 0, st, e0, e1, 0, a);
    } else if (small) {
        hipLaunchKernelGGL((k_wgrad<32, 128, 1, 4>), grid, dim3(256), 0, st, a);
    } else {
        hipLaunchKernelGGL((k_wgrad<64, 64, 2, 2>), grid, dim3(256), 0, st, a);
    }
    if (nsplit > 1) launch_wgrad_reduce(a.slab, a.dw, a.db, nsplit, a.Cs, 1 << a.lgCl, st);
    return nsplit;
}

// dw[(s*Cl + l)*16 + tap] = sum_z slab[z][s][tap*Cl + l].  A block owns 64 consecutive slab columns;
// SL lanes per column each add every SL-th slab, then LDS combines the lanes in lane order (the sum
// order is fixed by (nsplit, SL) alone: bitwise reproducible).
__global__ __launch_bounds__(1024) void k_wgrad_reduce(const float* __restrict__ slab, float* __restrict__ dw, float* __restrict__ db,
                                                       int nsplit, int Cs, int lgCl) {
    __shared__ float sh[16][64];
    const int N = 16 << lgCl, Cl = 1 << lgCl;
    const size_t total = (size_t)Cs * N;
    const int cl = threadIdx.x & 63, zl = threadIdx.x >> 6, SL = blockDim.x >> 6;
    const size_t wblocks = total / 64;
    const bool bias = blockIdx.x >= wblocks;                       // the blocks behind the weight columns: db
    const size_t idx = bias ? (size_t)(blockIdx.x - wblocks) * 64 + cl : (size_t)blockIdx.x * 64 + cl;
    const float* src = bias ? slab + (size_t)nsplit * total : slab;
    const size_t zstride = bias ? (size_t)Cs : total;
    const bool live = !bias || idx < (size_t)Cs;
    float acc = 0.f;
    if (live) {
#pragma unroll 8
        for (int z = zl; z < nsplit; z += SL) acc += src[(size_t)z * zstride + idx];
    }
    if (SL > 1) {
        sh[zl][cl] = acc;
        __syncthreads();
        if (zl != 0) return;
        for (int k = 1; k < SL; ++k) acc += sh[k][cl];
    }
    if (!live) return;
    if (bias) { db[idx] = acc; return; }
    const int j = (int)(idx & (size_t)(N - 1)), s_ = (int)(idx >> (4 + lgCl));      // N = 16 << lgCl
    const int tap = j >> lgCl, l = j & (Cl - 1);
    dw[((size_t)s_ * Cl + l) * 16 + tap] = acc;
}

static int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

static void launch_wgrad_reduce(const float* slab, float* dw, float* db, int nsplit, int Cs, int Cl, hipStream_t st) {
    const size_t total = (size_t)Cs * 16 * Cl;                 // a multiple of 64 (Cs, Cl >= 32)
    int SL = 1;
    while (SL < 16 && SL < nsplit) SL *= 2;
    const unsigned blocks = (unsigned)(total / 64) + (db ? (unsigned)((Cs + 63) / 64) : 0u);
    hipLaunchKernelGGL(k_wgrad_reduce, dim3(blocks), dim3(64 * SL), 0, st, slab, dw, db, nsplit, Cs, ilog2(Cl));
}

}  // namespace siggan
