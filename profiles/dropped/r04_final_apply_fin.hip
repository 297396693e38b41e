#ifdef FUSED_FINAL_FIN
// EXPERIMENT (-DFUSED_FINAL_FIN): the last block's BatchNorm-backward finalize folded into its apply kernel the same way
// (k_bn_bwd_fin_apply): 512 workgroups walk the strips (as k_final_bwd_reduce does), each adds the 512 partial rows of the 32
// channels in its prologue; the final conv's weight-gradient rows are then summed off the critical lane (launch_final_wsum).
template <class T>
__global__ __launch_bounds__(256) void k_final_bnbwd_apply_fin(const float* __restrict__ dpre, const float* __restrict__ Wt,
                                                               const T* __restrict__ y, float* __restrict__ bn, T* __restrict__ dy, int S,
                                                               int nstrips, const float* __restrict__ p0, const float* __restrict__ p1,
                                                               int nch, int64_t R, float* __restrict__ dgamma, float* __restrict__ dbeta) {
    constexpr int RY = 4, C = 32;
    __shared__ f4v shr[2][32][8];
    __shared__ float sp[(RY + 2) * 34];
    const int c4 = threadIdx.x & 7, xi = threadIdx.x >> 3;
    f4v s = {0.f, 0.f, 0.f, 0.f}, q = s;
#pragma unroll 4
    for (int k = xi; k < nch; k += 32) { s += ldg4(p0 + (size_t)k * C + c4 * 4); q += ldg4(p1 + (size_t)k * C + c4 * 4); }
    shr[0][xi][c4] = s; shr[1][xi][c4] = q;
    f4v w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = ldg4(Wt + k * 32 + c4 * 4);
    const f4v sc = ldg4(bn + c4 * 4), sf = ldg4(bn + C + c4 * 4), mu = ldg4(bn + 2 * C + c4 * 4), rs = ldg4(bn + 3 * C + c4 * 4);
    __syncthreads();
    s = shr[0][0][c4]; q = shr[1][0][c4];
#pragma unroll
    for (int k = 1; k < 32; ++k) { s += shr[0][k][c4]; q += shr[1][k][c4]; }
    const float invR = 1.0f / (float)R;
    const f4v c1 = s * invR, c2 = q * invR;
    if (blockIdx.x == 0 && xi == 0) {
        *reinterpret_cast<f4v*>(dbeta + c4 * 4) = s; *reinterpret_cast<f4v*>(dgamma + c4 * 4) = q;
        *reinterpret_cast<f4v*>(bn + 4 * C + c4 * 4) = c1; *reinterpret_cast<f4v*>(bn + 5 * C + c4 * 4) = c2;
    }
    for (int sid = blockIdx.x; sid < nstrips; sid += gridDim.x) {
        const StripId t = strip_of<RY>(sid, S, xi);
        const float pe = dpre_patch_elem<RY>(dpre, t.n, t.y0, t.x - xi, S, threadIdx.x);
        const size_t o0 = (((size_t)t.n * S + t.y0) * S + t.x) * C + c4 * 4;
        f4v yv[RY];
#pragma unroll
        for (int r = 0; r < RY; ++r) yv[r] = ld4<T>(y + o0 + (size_t)r * S * C);
        __syncthreads();                       // (the previous strip's patch is read out)
        if (threadIdx.x < (RY + 2) * 34) sp[threadIdx.x] = pe;
        __syncthreads();
        float d[RY + 2][3];
        read_dpre_patch<RY>(sp, xi, d);
#pragma unroll
        for (int r = 0; r < RY; ++r) {
            const f4v g = final_dact<RY>(d, w, r);
            const f4v v = yv[r];
            f4v o;
            o.x = sc.x * ((fmaf(v.x, sc.x, sf.x) > 0.f ? g.x : 0.f) - c1.x - (v.x - mu.x) * rs.x * c2.x);
            o.y = sc.y * ((fmaf(v.y, sc.y, sf.y) > 0.f ? g.y : 0.f) - c1.y - (v.y - mu.y) * rs.y * c2.y);
            o.z = sc.z * ((fmaf(v.z, sc.z, sf.z) > 0.f ? g.z : 0.f) - c1.z - (v.z - mu.z) * rs.z * c2.z);
            o.w = sc.w * ((fmaf(v.w, sc.w, sf.w) > 0.f ? g.w : 0.f) - c1.w - (v.w - mu.w) * rs.w * c2.w);
            st4<T>(dy + o0 + (size_t)r * S * C, o);
        }
    }
}
#endif
