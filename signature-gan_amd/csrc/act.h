// act.h -- element type of the library-owned activation / gradient tensors and of the packed weight
// copies the MFMA kernels read (gfx950).
//
// The reference computes in fp32 only (vanilla_gan_model.py:107-120: no autocast / half anywhere); fp32 is
// the default and the parity path.  DT_BF16 / DT_F16 are the build-defined narrow variants BASELINE.json's
// configs[2] / configs[4] name: activations and activation gradients are STORED in HBM as 16-bit values,
// every kernel still computes in fp32 (MFMA accumulators, BatchNorm statistics, losses, weight gradients,
// Adam and the master weights stay fp32), and only the implicit-GEMM kernels read 16-bit operands
// (v_mfma_f32_32x32x16_{bf16,f16}).  All kernels that touch such a tensor are templates over its element
// type T and go through ld4 / st4 below, so for T = float they compile to exactly the fp32 code.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

// `done` (the launch_* functions that take one, GConvArgs::done): the event is bound to the launch's LAST kernel as hipExtLaunchKernel's stop event -- it
// rides on that dispatch packet's own completion signal.  A side lane forked this way (hipStreamWaitEvent on `done`) costs the
// producing lane nothing; hipEventRecord behind the kernel is a marker packet of its own, ~5 us before the next kernel starts.
#define SIGGAN_LAUNCH_EV(ev, kernel, grid, block, shmem, stream, ...)                                            \
    do {                                                                                                         \
        if (ev) hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, nullptr, ev, 0, __VA_ARGS__);          \
        else hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                \
    } while (0)


namespace siggan {

typedef __bf16 bf16_t;
typedef _Float16 f16_t;
enum DType : int { DT_F32 = 0, DT_BF16 = 1, DT_F16 = 2 };
static inline size_t dt_size(int dt) { return dt == DT_F32 ? 4 : 2; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef f16_t f16x4 __attribute__((ext_vector_type(4)));
typedef f16_t f16x8 __attribute__((ext_vector_type(8)));

// four consecutive elements (16 bytes of fp32, 8 bytes of a 16-bit type) <-> four floats; the pointer must be
// aligned to the access (every tensor row is a multiple of 4 elements).  Conversions round to nearest even.
template <class T> __device__ __forceinline__ f32x4 ld4(const T* p);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
template <> __device__ __forceinline__ f32x4 ld4<bf16_t>(const bf16_t* p) {
    const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <> __device__ __forceinline__ f32x4 ld4<f16_t>(const f16_t* p) {
    const f16x4 v = *reinterpret_cast<const f16x4*>(p);
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
template <class T> __device__ __forceinline__ void st4(T* p, f32x4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, f32x4 v) { *reinterpret_cast<f32x4*>(p) = v; }
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, f32x4 v) {
    *reinterpret_cast<bf16x4*>(p) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
}
template <> __device__ __forceinline__ void st4<f16_t>(f16_t* p, f32x4 v) {
    *reinterpret_cast<f16x4*>(p) = f16x4{(f16_t)v[0], (f16_t)v[1], (f16_t)v[2], (f16_t)v[3]};
}
template <class T> __device__ __forceinline__ float ld1(const T* p) { return (float)*p; }
template <class T> __device__ __forceinline__ void st1(T* p, float v) { *p = (T)v; }

// run BODY with `T` bound to the element type of dtype code DT
#define SIGGAN_DT_SWITCH(DT, T, ...)                                     \
    do {                                                                 \
        if ((DT) == DT_F32) { using T = float; __VA_ARGS__; }            \
        else if ((DT) == DT_BF16) { using T = bf16_t; __VA_ARGS__; }     \
        else { using T = f16_t; __VA_ARGS__; }                           \
    } while (0)

}  // namespace siggan
