#ifdef FUSED_TRAIN_FIN
// EXPERIMENT (scratch/build_variant.sh -DFUSED_TRAIN_FIN): k_bn_train_fin folded into k_bn_relu the same way -- the prologue adds
// k_colreduce<FStats>'s nch partial rows (shifted sums around row 0 of y) for the workgroup's 32 channels, forms scale / shift,
// and the (slice, chunk 0) workgroup writes the layer's table and the running statistics.
template <class T>
__global__ __launch_bounds__(256) void k_bn_relu_fin(const T* __restrict__ y, T* __restrict__ a, int64_t R, int C, float* __restrict__ bn,
                                                     const float* __restrict__ p0, const float* __restrict__ p1, int nch,
                                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                                     float* __restrict__ rmean, float* __restrict__ rvar, int64_t* __restrict__ batches,
                                                     float momentum, float eps, int rows_per_chunk) {
    __shared__ f32x4 sh[2][32][8];
    const int c4 = threadIdx.x & 7, rl = threadIdx.x >> 3;
    const int ch = blockIdx.x * 32 + c4 * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, q = s;
#pragma unroll 4
    for (int k = rl; k < nch; k += 32) {
        s += *reinterpret_cast<const f32x4*>(p0 + (size_t)k * C + ch);
        q += *reinterpret_cast<const f32x4*>(p1 + (size_t)k * C + ch);
    }
    sh[0][rl][c4] = s; sh[1][rl][c4] = q;
    __syncthreads();
    s = sh[0][0][c4]; q = sh[1][0][c4];
#pragma unroll
    for (int k = 1; k < 32; ++k) { s += sh[0][k][c4]; q += sh[1][k][c4]; }
    const float invR = 1.0f / (float)R;
    const f32x4 y0 = ld4<T>(y + ch), g4 = *reinterpret_cast<const f32x4*>(gamma + ch), b4 = *reinterpret_cast<const f32x4*>(beta + ch);
    f32x4 sc, sf, mean, var, rstd;
#pragma unroll
    for (int e = 0; e < 4; ++e) {                  // k_bn_train_fin's expressions
        const float d = s[e] * invR;
        mean[e] = y0[e] + d;
        float v = q[e] * invR - d * d;
        var[e] = v > 0.f ? v : 0.f;
        rstd[e] = 1.0f / sqrtf(var[e] + eps);
        sc[e] = g4[e] * rstd[e]; sf[e] = b4[e] - mean[e] * sc[e];
    }
    if (blockIdx.y == 0 && rl == 0) {
        *reinterpret_cast<f32x4*>(bn + ch) = sc; *reinterpret_cast<f32x4*>(bn + C + ch) = sf;
        *reinterpret_cast<f32x4*>(bn + 2 * C + ch) = mean; *reinterpret_cast<f32x4*>(bn + 3 * C + ch) = rstd;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float unb = R > 1 ? var[e] * ((float)R / (float)(R - 1)) : var[e];
            rmean[ch + e] = momentum * mean[e] + (1.0f - momentum) * rmean[ch + e];
            rvar[ch + e] = momentum * unb + (1.0f - momentum) * rvar[ch + e];
        }
        if (blockIdx.x == 0 && c4 == 0 && batches) batches[0] += 1;
    }
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_chunk, r1 = r0 + rows_per_chunk < R ? r0 + rows_per_chunk : R;
#pragma unroll 4
    for (int64_t r = r0 + rl; r < r1; r += 32) {
        const size_t i = (size_t)r * C + ch;
        const f32x4 v = ld4<T>(y + i);
        st4<T>(a + i, f32x4{fmaxf(fmaf(v[0], sc[0], sf[0]), 0.f), fmaxf(fmaf(v[1], sc[1], sf[1]), 0.f),
                            fmaxf(fmaf(v[2], sc[2], sf[2]), 0.f), fmaxf(fmaf(v[3], sc[3], sf[3]), 0.f)});
    }
}
// statistics (k_colreduce<FStats>) + [finalize + scale / shift / ReLU] in two launches instead of three
void launch_bn_train_relu(int dt, const void* yv, void* av, int64_t R, int C, const float* gamma, const float* beta, float* rmean,
                          float* rvar, int64_t* batches, float* bn, float* partial, float momentum, float eps, hipStream_t s) {
    const ColPlan pl = col_plan(R, C);
    float* p0 = partial; float* p1 = partial + (size_t)pl.nch * C;
    const int slices = C / 32;
    int64_t chunks = 256 / slices; if (chunks < 1) chunks = 1;
    if (chunks > (R + 63) / 64) chunks = (R + 63) / 64;
    int rpc = (int)((R + chunks - 1) / chunks); rpc = ((rpc + 31) / 32) * 32;
    chunks = (R + rpc - 1) / rpc;
    SIGGAN_DT_SWITCH(dt, T, {
        const T* y = (const T*)yv;
        hipLaunchKernelGGL((k_colreduce<FStats<T>>), dim3(pl.cbx, pl.nch), dim3(256), 0, s, FStats<T>{y}, R, C, pl.cg, pl.rows, p0, p1);
        hipLaunchKernelGGL(k_bn_relu_fin<T>, dim3(slices, (unsigned)chunks), dim3(256), 0, s, y, (T*)av, R, C, bn, p0, p1, pl.nch, gamma, beta,
                           rmean, rvar, batches, momentum, eps, rpc);
    });
}
#endif
