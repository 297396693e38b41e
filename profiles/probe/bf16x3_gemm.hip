// go / no-go microbenchmark: fp32 GEMM C[M][N] = A[M][K] . B[N][K]^T with the products formed on the bf16 matrix cores from an
// error-free 3-way split of every fp32 operand (x = hi + mid + lo, RNE at each level), six partial products per k-step:
// hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid  (dropped: mid.lo, lo.mid, lo.lo <= 2^-26 relative).
// B is pre-split (weights: once per optimiser step).  A: PRE = 0 split on the fly in the loader (round 2's experiment: the
// kernel became bound by loads + split, 26.9 us with ONE product); PRE = 1 pre-split too -- three bf16 planes written where the
// operand is PRODUCED (an activation epilogue would store hi / mid / lo), so the GEMM streams bf16 like k_gconv16 does
// (round 4).  Operand bytes per element: 4 (fp32 MFMA) -> 6 (three bf16 planes).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BK = 32, LD = BK + 8;

struct Tri { bf16x8 h, m, l; };
__device__ __forceinline__ Tri split8(const f32x4 a, const f32x4 b) {
    Tri t;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float x = i < 4 ? a[i] : b[i - 4];
        const bf16_t h = (bf16_t)x;
        const float r1 = x - (float)h;
        const bf16_t m = (bf16_t)r1;
        const float r2 = r1 - (float)m;
        t.h[i] = h; t.m[i] = m; t.l[i] = (bf16_t)r2;
    }
    return t;
}
__global__ void k_split_b(const float* __restrict__ B, bf16_t* __restrict__ Bp, size_t n) {   // planes [3][n]
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= n) return;
    const Tri t = split8(*reinterpret_cast<const f32x4*>(B + i), *reinterpret_cast<const f32x4*>(B + i + 4));
    *reinterpret_cast<bf16x8*>(Bp + i) = t.h; *reinterpret_cast<bf16x8*>(Bp + n + i) = t.m; *reinterpret_cast<bf16x8*>(Bp + 2 * n + i) = t.l;
}

template <int NT, int PRE>   // NT = number of partial products: 6 (default), 3 (hi.hi + hi.mid + mid.hi), 1 (plain bf16)
__global__ __launch_bounds__(256) void k_gemm3(const float* __restrict__ A, const bf16_t* __restrict__ Ap, const bf16_t* __restrict__ Bp,
                                               float* __restrict__ C, int M, int N, int K) {
    constexpr int BM = 64, BN = 64;
    __shared__ __attribute__((aligned(16))) bf16_t sA[2][3][BM * LD];
    __shared__ __attribute__((aligned(16))) bf16_t sB[2][3][BN * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = N / BN;
    const int m0 = (blockIdx.x / tiles_n) * BM, n0 = (blockIdx.x % tiles_n) * BN;
    const int kc = tid & 3, rloc = tid >> 2;
    const float* ap = A + (size_t)(m0 + rloc) * K + kc * 8;
    const bf16_t* app = Ap + (size_t)(m0 + rloc) * K + kc * 8;
    const size_t plane = (size_t)N * K, plane_a = (size_t)M * K;
    const bf16_t* bp = Bp + (size_t)(n0 + rloc) * K + kc * 8;
    const int nk = K / BK;
    f32x4 ra0, ra1; bf16x8 rb[3], rap[3];
    auto load_tile = [&]() __attribute__((always_inline)) {
        if (PRE) {
#pragma unroll
            for (int p = 0; p < 3; ++p) if (p == 0 || NT > 1) rap[p] = *reinterpret_cast<const bf16x8*>(app + p * plane_a);
            app += BK;
        } else { ra0 = *reinterpret_cast<const f32x4*>(ap); ra1 = *reinterpret_cast<const f32x4*>(ap + 4); ap += BK; }
#pragma unroll
        for (int p = 0; p < 3; ++p) rb[p] = *reinterpret_cast<const bf16x8*>(bp + p * plane);
        bp += BK;
    };
    auto store_tile = [&](int buf) __attribute__((always_inline)) {
        const int o = rloc * LD + kc * 8;
        if (PRE) {
#pragma unroll
            for (int p = 0; p < 3; ++p) if (p == 0 || NT > 1) *reinterpret_cast<bf16x8*>(&sA[buf][p][o]) = rap[p];
        } else {
            const Tri t = split8(ra0, ra1);
            *reinterpret_cast<bf16x8*>(&sA[buf][0][o]) = t.h; *reinterpret_cast<bf16x8*>(&sA[buf][1][o]) = t.m; *reinterpret_cast<bf16x8*>(&sA[buf][2][o]) = t.l;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(&sB[buf][p][o]) = rb[p];
    };
    f32x16 acc0, acc1, acc2;      // three independent chains (hi.hi | hi.mid + mid.hi | the three small products): no MFMA waits for the one before it
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; }
    load_tile(); store_tile(0);
    if (nk > 1) load_tile();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        const int oa = (wm * 32 + li) * LD + 8 * lh, ob = (wn * 32 + li) * LD + 8 * lh;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 fa[3], fb[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                fa[p] = *reinterpret_cast<const bf16x8*>(&sA[buf][p][oa + 16 * s]);
                fb[p] = *reinterpret_cast<const bf16x8*>(&sB[buf][p][ob + 16 * s]);
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc0, 0, 0, 0);
            if (NT >= 3) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], acc1, 0, 0, 0);
            if (NT >= 6) acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], acc2, 0, 0, 0);
            if (NT >= 3) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], acc1, 0, 0, 0);
            if (NT >= 6) {
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[2], acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], acc2, 0, 0, 0);
            }
            if (s == 0 && kt + 1 < nk) store_tile(buf ^ 1);
            if (s == 1 && kt + 2 < nk) load_tile();
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh, n = n0 + wn * 32 + li;
        C[(size_t)m * N + n] = acc0[r] + (acc1[r] + acc2[r]);
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 16384, N = argc > 2 ? atoi(argv[2]) : 128, K = argc > 3 ? atoi(argv[3]) : 1024;
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    srand(1);
    for (auto& v : hA) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : hB) v = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f;
    float *A, *B, *C; bf16_t *Bp, *Ap;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&B, hB.size() * 4); hipMalloc(&C, (size_t)M * N * 4); hipMalloc(&Bp, hB.size() * 6); hipMalloc(&Ap, hA.size() * 6);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_split_b, dim3((unsigned)((hB.size() / 8 + 255) / 256)), dim3(256), 0, 0, B, Bp, hB.size());
    hipLaunchKernelGGL(k_split_b, dim3((unsigned)((hA.size() / 8 + 255) / 256)), dim3(256), 0, 0, A, Ap, hA.size());
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<float> hC((size_t)M * N);
    auto run = [&](int nt, int pre) {
        const dim3 g((M / 64) * (N / 64)), b(256);
        auto go = [&]() {
            if (pre) {
                if (nt == 6) hipLaunchKernelGGL((k_gemm3<6, 1>), g, b, 0, 0, A, Ap, Bp, C, M, N, K);
                else if (nt == 3) hipLaunchKernelGGL((k_gemm3<3, 1>), g, b, 0, 0, A, Ap, Bp, C, M, N, K);
                else hipLaunchKernelGGL((k_gemm3<1, 1>), g, b, 0, 0, A, Ap, Bp, C, M, N, K);
            } else {
                if (nt == 6) hipLaunchKernelGGL((k_gemm3<6, 0>), g, b, 0, 0, A, Ap, Bp, C, M, N, K);
                else if (nt == 3) hipLaunchKernelGGL((k_gemm3<3, 0>), g, b, 0, 0, A, Ap, Bp, C, M, N, K);
                else hipLaunchKernelGGL((k_gemm3<1, 0>), g, b, 0, 0, A, Ap, Bp, C, M, N, K);
            }
        };
        for (int i = 0; i < 5; ++i) go();
        hipEventRecord(e0, 0);
        for (int i = 0; i < 50; ++i) go();
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost);
        // accuracy on sampled entries vs double, and the error an fp32 FMA chain makes on the same entries
        double worst = 0, worst32 = 0, scale = 0;
        for (int t = 0; t < 4000; ++t) {
            const int m = rand() % M, n = rand() % N;
            double ref = 0; float f = 0.f;
            for (int k = 0; k < K; ++k) { ref += (double)hA[(size_t)m * K + k] * hB[(size_t)n * K + k]; f = fmaf(hA[(size_t)m * K + k], hB[(size_t)n * K + k], f); }
            worst = fmax(worst, fabs(hC[(size_t)m * N + n] - ref)); worst32 = fmax(worst32, fabs((double)f - ref)); scale = fmax(scale, fabs(ref));
        }
        const double us = ms * 1e3 / 50, fl = 2.0 * M * N * K;
        printf("M=%d N=%d K=%d  A %s  products=%d : %7.1f us  %6.1f TFLOP/s (algorithmic)   max err / scale %.2e   (fp32 FMA chain: %.2e)\n",
               M, N, K, pre ? "pre-split planes" : "split in the loader", nt, us, fl / us / 1e6, worst / scale, worst32 / scale);
    };
    for (int pre = 1; pre >= 0; --pre) { run(6, pre); run(3, pre); run(1, pre); }
    return 0;
}
