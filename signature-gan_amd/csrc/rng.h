// rng.h -- the library's counter-based RNG (Philox4x32-10): z ~ N(0,1) and the Dropout2d keep tables.
// counter = (element group index, stream id, call counter), key = seed ^ (call counter's high half); one draw yields four
// values.  The normal transform is Box-Muller on the hardware transcendental units (v_log / v_sin / v_cos / v_sqrt): the
// latent batch is drawn inside the fc kernel by every workgroup that needs it, so a draw has to cost tens of cycles, not the
// hundreds of the range-reduced libm routines.  Every kernel that draws goes through these functions, so the value of
// element e of stream s at call counter c is the same whichever kernel produces it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "act.h"

namespace siggan {

struct DevState;

__device__ __forceinline__ uint4 philox(uint4 c, uint2 k) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(M0, c.x), lo0 = M0 * c.x;
        const uint32_t hi1 = __umulhi(M1, c.z), lo1 = M1 * c.z;
        c = make_uint4(hi1 ^ c.y ^ k.x, lo1, hi0 ^ c.w ^ k.y, lo0);
        k.x += W0; k.y += W1;
    }
    return c;
}
__device__ __forceinline__ float u01(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }
__device__ __forceinline__ uint4 draw_raw(unsigned long long seed, unsigned long long ctr, uint64_t idx, uint32_t stream_id) {
    return philox(make_uint4((uint32_t)idx, (uint32_t)(idx >> 32), stream_id, (uint32_t)ctr),
                  make_uint2((uint32_t)seed, (uint32_t)(seed >> 32) ^ (uint32_t)(ctr >> 32)));
}
// four standard normals from one draw (elements [4*idx, 4*idx + 4) of the stream)
__device__ __forceinline__ f32x4 normal4(const uint4 r) {
    const float r0 = __fsqrt_rn(-2.0f * __logf(u01(r.x))), r1 = __fsqrt_rn(-2.0f * __logf(u01(r.z)));
    const float a0 = 6.283185307179586f * u01(r.y), a1 = 6.283185307179586f * u01(r.w);
    return f32x4{r0 * __cosf(a0), r0 * __sinf(a0), r1 * __cosf(a1), r1 * __sinf(a1)};
}

}  // namespace siggan
