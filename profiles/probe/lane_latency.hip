// What a lane operation costs on this runtime (round 4, DESIGN 4 "what a lane operation costs"):
//   hipcc --offload-arch=gfx950 -O2 -o lane_latency lane_latency.hip && ./lane_latency
// Chains of N dependent tiny kernels, timed on the host around a synchronize:
//   same     : all on one stream (the kernel-to-kernel boundary)
//   record   : one stream, a hipEventRecord between the kernels (nobody waits for it)
//   pingpong : kernels alternate between two streams, each waits for the other's event (a cross-queue dependency per kernel)
//   stopev   : as pingpong, but the event is the kernel's own completion (hipExtLaunchKernel's stop event), no marker packet
// each with fence-free events (hipEventDisableSystemFence) and with hipEventDisableTiming alone.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void k_tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.0f; }
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const int N = 2000;
    float* d; CK(hipMalloc(&d, 256)); CK(hipMemset(d, 0, 256));
    hipStream_t s[2]; CK(hipStreamCreateWithFlags(&s[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s[1], hipStreamNonBlocking));
    for (int flags_i = 0; flags_i < 2; ++flags_i) {
        const unsigned fl = flags_i == 0 ? (hipEventDisableTiming | hipEventDisableSystemFence) : hipEventDisableTiming;
        std::vector<hipEvent_t> ev(64);
        for (auto& e : ev) CK(hipEventCreateWithFlags(&e, fl));
        for (int mode = 0; mode < 4; ++mode) {
            double best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipDeviceSynchronize());
                const double t0 = now();
                for (int i = 0; i < N; ++i) {
                    hipEvent_t e = ev[i % 64];
                    if (mode == 0) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s[0], d); }
                    else if (mode == 1) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s[0], d); CK(hipEventRecord(e, s[0])); }
                    else if (mode == 2) {
                        hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s[i & 1], d);
                        CK(hipEventRecord(e, s[i & 1])); CK(hipStreamWaitEvent(s[(i + 1) & 1], e, 0));
                    } else {
                        hipExtLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s[i & 1], nullptr, e, 0, d);
                        CK(hipStreamWaitEvent(s[(i + 1) & 1], e, 0));
                    }
                }
                CK(hipDeviceSynchronize());
                const double dt = (now() - t0) / N * 1e6;
                if (dt < best) best = dt;
            }
            const char* names[] = {"same", "record", "pingpong", "stopev"};
            printf("%-24s %-9s %7.2f us per kernel\n", flags_i == 0 ? "fence-free events" : "default (fenced) events", names[mode], best);
        }
        for (auto& e : ev) (void)hipEventDestroy(e);
    }
    float h = 0; CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost)); printf("kernels run: %.0f\n", h);
    return 0;
}
