// What a grid-wide hand-off costs inside ONE launch, against a kernel boundary (round 4; VERDICT r3 #1 asked for chain kernels
// budgeted against MI355X_MICROARCH.md's barrier-xcd row -- this measures the row on the box):
//   hipcc --offload-arch=gfx950 -O2 -o grid_barrier grid_barrier.hip && ./grid_barrier
// A persistent kernel of G workgroups x 256 threads runs N phases; in every phase a workgroup writes `bytes` of a buffer, all
// workgroups meet at a barrier (one agent-scope counter; release before, acquire after: the data crosses XCDs), and the next
// phase reads what ANOTHER workgroup (on another XCD: index + 1) wrote.  Compared with N launches of the same phase body on
// one stream (kernel boundary = the hand-off).  G = 256 and 512 (<= resident capacity: every workgroup is on the chip).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ void phase_body(float* buf, int floats_per_wg, int phase, int wg, int nwg, float* sink) {
    const int src = (wg + 1) % nwg;
    float acc = 0.f;
    for (int i = threadIdx.x; i < floats_per_wg; i += 256) acc += buf[(size_t)src * floats_per_wg + i];      // the neighbour's last phase
    __syncthreads();
    for (int i = threadIdx.x; i < floats_per_wg; i += 256) buf[(size_t)wg * floats_per_wg + i] = acc * 1e-9f + (float)phase;
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_persistent(float* buf, int floats_per_wg, int nphase, unsigned* counter, float* sink) {
    const int wg = blockIdx.x, nwg = gridDim.x;
    for (int p = 0; p < nphase; ++p) {
        phase_body(buf, floats_per_wg, p, wg, nwg, sink);
        // grid barrier: release this workgroup's writes, arrive, wait for everyone, acquire
        __syncthreads();
        if (threadIdx.x == 0) {
            __atomic_thread_fence(__ATOMIC_RELEASE);       // (hipcc: agent scope -- write back this XCD's L2)
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(p + 1) * (unsigned)nwg;
            unsigned spins = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < (1u << 24)) __builtin_amdgcn_s_sleep(2);
            __atomic_thread_fence(__ATOMIC_ACQUIRE);       // (invalidate what other XCDs may have changed)
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void k_phase(float* buf, int floats_per_wg, int phase, float* sink) {
    phase_body(buf, floats_per_wg, phase, blockIdx.x, gridDim.x, sink);
}
int main() {
    const int N = 200;
    float *buf, *sink; unsigned* ctr;
    CK(hipMalloc(&buf, 512 * 16384 * sizeof(float))); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&ctr, 64));
    CK(hipMemset(buf, 0, 512 * 16384 * sizeof(float)));
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int grids[] = {256, 512};
    const int floats[] = {64, 1024, 4096, 16384};          // 256 B, 4 KB, 16 KB, 64 KB per workgroup and phase
    for (int g : grids) for (int f : floats) {
        double best_p = 1e9, best_l = 1e9;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemsetAsync(ctr, 0, 64, s)); CK(hipStreamSynchronize(s));
            double t0 = now();
            hipLaunchKernelGGL(k_persistent, dim3(g), dim3(256), 0, s, buf, f, N, ctr, sink);
            CK(hipStreamSynchronize(s));
            double dt = (now() - t0) / N * 1e6; if (dt < best_p) best_p = dt;
            t0 = now();
            for (int p = 0; p < N; ++p) hipLaunchKernelGGL(k_phase, dim3(g), dim3(256), 0, s, buf, f, p, sink);
            CK(hipStreamSynchronize(s));
            dt = (now() - t0) / N * 1e6; if (dt < best_l) best_l = dt;
        }
        printf("G = %3d workgroups, %6.1f KB per workgroup and phase (%5.1f MB per phase): one launch with grid barriers %6.2f us per phase, one launch per phase %6.2f us\n",
               g, f * 4 / 1024.0, (double)g * f * 4 / 1e6, best_p, best_l);
    }
    return 0;
}
