// ops.h -- the bandwidth-bound kernels around the MFMA GEMMs (gfx950).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "act.h"

namespace siggan {

// device-resident per-call state: nothing a captured launch needs lives in kernel arguments
struct DevState {
    unsigned long long seed;
    unsigned long long rng_ctr;      // bumped once per step phase
    // scalars of the pending Adam apply (written by k_adam_prepare)
    float step_size;                 // lr / (1 - beta1^t)
    float bc2_sqrt;                  // sqrt(1 - beta2^t)
    float grad_mul;                  // grad_scale * clip coefficient
    float grad_norm;                 // pre-clip global L2 norm (after grad_scale)
    float sumsq;                     // scratch of the norm reduction
    int skip;                        // the pending update is skipped (fp16: the gradient arena holds an inf / NaN)
    float pad[2];
};

// ---- RNG (Philox4x32-10, counter = (index, stream, call counter)) --------------------------
void launch_randn(float* out, int64_t n, const DevState* st, uint32_t stream_id, hipStream_t s);
// dropout multiplier tables (0 or 1/keep) of several blocks in one launch: table t = elements [elem0[t], elem0[t] + n[t])
// (elem0 % 4 == 0) of stream sid[t], drawn at counter rng_ctr + ctr_add
void launch_dropnoise_multi(int nt, float* const* out, const int64_t* n, const int64_t* elem0, const uint32_t* sid, float keep,
                            const DevState* st, hipStream_t s, uint32_t ctr_add = 0);
// out = mask * (1/keep)
void launch_mask_to_noise(const float* mask, float* out, int64_t n, float keep, hipStream_t s);
void launch_tick(DevState* st, hipStream_t s);

// ---- fused "prepare" pass: every weight re-pack / BN-eval fold of one network in ONE launch -----
enum PrepType : int { PREP_PACK_DOWN = 0, PREP_PACK_UP, PREP_FC_T, PREP_CLS, PREP_BN_EVAL, PREP_SCALE, PREP_TAPS };
struct PrepJob {
    int type, O, I, perm;        // PACK_*: (O,I) channel counts; FC_T: O=K, I=C0; CLS: O=C; BN_EVAL: O=C, perm=perm_c0
    int dt;                      // PACK_DOWN / PACK_UP: element type of dst (DT_F32 / DT_BF16 / DT_F16); fp32 for the rest
    const float* mul;            // PACK_* / CLS / SCALE: optional device scalar every weight is multiplied by (1 / sigma of a
                                 // spectrally normalised layer); SCALE: dst[i] = src[i] * mul over O elements (fp32 copy)
    const float* src;            // weights (torch layout) / gamma
    const float* src2;           // beta
    const float* src3;           // running_mean
    const float* src4;           // running_var
    float* dst;
};
struct PrepTable {
    static constexpr int MAXJ = 32;
    int njobs;
    int overflow;                // set by prep_add when a job did not fit: launch_prepare then fails instead of dropping it
    long long prefix[MAXJ + 1];  // prefix sums of the jobs' element counts
    PrepJob job[MAXJ];
};
void prep_add(PrepTable& t, const PrepJob& j, long long count);
bool launch_prepare(const PrepTable& t, float bn_eps, hipStream_t s);   // false: the table overflowed (nothing launched)
// the same plus a first-Discriminator-block forward (no dropout table) of B images, as further blocks of the one launch
bool launch_prepare_conv1(const PrepTable& t, float bn_eps, int dt, const float* x, const float* W, const float* b, float slope,
                          void* out, int B, int S, hipStream_t s);

// ---- Generator pieces ---------------------------------------------------------------------
// y[n][f'] = z[n,:] . W[f,:] + b[f],  f' = hw*C0 + c  <->  f = c*16 + hw   (NHWC feature order)
// bn_affine_relu != nullptr (eval): y = relu((z.Wt + b) * scale + shift) with [scale | shift] of the folded BatchNorm1d.
// z == nullptr: z ~ N(0,1) is drawn inside the kernel (the values launch_randn(z_out, B*K, st, stream_id) would write) and
// also stored to z_out
// (dt: element type of the activation / gradient tensors passed as void*, see act.h)
void launch_fc_fwd(int dt, const float* z, const float* Wt, const float* b, void* y, int B, int K, int C0, hipStream_t s,
                   const float* bn_affine_relu = nullptr, const DevState* st = nullptr, uint32_t stream_id = 0,
                   float* z_out = nullptr);
// fc.hip: Linear + BatchNorm1d + ReLU in ONE launch on the fp32 matrix cores.  bne != nullptr: eval mode (folded [scale | shift]
// table, only `a` is written); else training: y (pre-BN), a, the bn table [scale | shift | mean | rstd | . | .] the backward
// reuses, and the running statistics / batch counter are written.  W is the torch-layout weight (no transposed copy).
// Returns false (nothing launched) when the shape is outside what the kernel covers (B > 256, or RNG draw with K % 4 != 0).
bool launch_fc_fwd_fused(int dt, const float* z, const float* W, const float* bias, void* y, void* a, const float* gamma,
                         const float* beta, float* rmean, float* rvar, int64_t* batches, float* bn, const float* bne,
                         float* z_out, const DevState* st, uint32_t sid, int B, int K, int C0, float momentum, float eps,
                         hipStream_t s);
// ReLU mask + BatchNorm1d backward + dW / db of the Linear in ONE launch: da, y element type dt; dW, db, dgamma, dbeta fp32 (torch order)
bool launch_fc_bwd_fused(int dt, const void* da, const void* y, const float* z, float* bn, float* dW, float* db, float* dgamma,
                         float* dbeta, int B, int K, int C0, hipStream_t s);
// dW[f][k] = sum_n dy[n][f'] z[n][k];  db[f] = sum_n dy[n][f']
void launch_fc_wgrad(int dt, const void* dy, const float* z, float* dW, float* db, int B, int K, int C0, hipStream_t s);

// training statistics over R rows + running-stat update (momentum, unbiased var) + batches++
void launch_bn_train_stats(int dt, const void* y, int64_t R, int C, const float* gamma, const float* beta,
                           float* rmean, float* rvar, int64_t* batches, float* bn, float* partial,
                           int perm_c0, float momentum, float eps, hipStream_t s);
// a = relu(y*scale + shift)
void launch_bn_relu(int dt, const void* y, void* a, int64_t R, int C, const float* bn, hipStream_t s);
// backward through relu(BN(y)): da (in) -> dy (in place); dgamma/dbeta (torch order) written.  The relu mask is
// re-derived from y and the layer's scale/shift (the forward's own expression), so the activation is not read.
void launch_bn_bwd(int dt, void* da, const void* y, int64_t R, int C, float* bn, float* partial,
                   float* dgamma, float* dbeta, int perm_c0, hipStream_t s, int pre_rows = 0, hipEvent_t done = nullptr);

// final 3x3 conv (C->1) + tanh, and its backward pieces.  act: [B][S][S][C] NHWC, img [B][S][S].
// bn != nullptr (training): `act` is the last block's PRE-BatchNorm tensor y and bn its [scale | shift] table -- the
// activation relu(fma(y, scale, shift)) is formed on load and never stored
void launch_final_fwd(int dt, const void* act, const float* Wt, const float* b, float* img, int B, int S, int C, hipStream_t s,
                      const float* bn = nullptr, hipEvent_t done = nullptr);
// backward of the last Generator block from dpre.  (1) launch_final_bwd_reduce: ONE read of y gives the BatchNorm-backward sums
// (d(act) of the final conv recomputed from dpre, never stored; relu mask re-derived from y) into `partial` AND the partial
// rows of the final conv's weight / bias gradient (activation re-derived from y) into `partial_w`;  (2) launch_final_bn_bwd_apply:
// ONE finalizer launch (dW / db from partial_w, dgamma / dbeta and the two means from `partial`), then dy
void launch_final_bwd_reduce(int dt, const float* dpre, const float* Wt, const void* y, int B, int S, int C, const float* bn,
                             float* partial, float* partial_w, hipStream_t s);
void launch_final_bn_bwd_apply(int dt, const float* dpre, const float* Wt, const void* y, void* dy, int B, int S, int C, float* bn,
                               const float* partial, const float* partial_w, float* dW, float* db, float* dgamma, float* dbeta,
                               hipStream_t s, hipEvent_t done = nullptr);

// ---- Discriminator pieces -----------------------------------------------------------------
// first block (Cin = 1): x = two segments (x0: n < n0, x1: the rest), out [B][S/2][S/2][C]
void launch_conv1_fwd(int dt, const float* x0, int n0, const float* x1, const float* W, const float* b,
                      const float* noise, float slope, void* out, int B, int S, int C, hipStream_t s);
void launch_conv1_wgrad(int dt, const void* dv, const float* x0, int n0, const float* x1, float* dW, float* db,
                        float* partial, int B, int S, int C, hipStream_t s);
// d(image) = conv1 input-gradient, times tanh' = 1 - img^2  ->  dpre
void launch_conv1_dgrad_tanh(int dt, const void* dv, const float* Wt, const float* img, float* dpre, int B, int S,
                             int C, hipStream_t s);
void launch_cls_fwd(int dt, const void* act, const float* wcp, const float* bc, float* logits, int B, int F, hipStream_t s);
void launch_cls_features(int dt, const void* act, float* feat, int B, int C, hipStream_t s);
// sigmoid + BCE (mean per segment) + d(logit); seg0 = first n0 rows with target y0, rest target y1
// gscale: d(logit) is multiplied by it (the fp16 gradient scale; every gradient downstream then carries it and the optimiser
// step divides it out again -- 1 for fp32 / bf16)
void launch_bce(const float* logits, int B, int n0, float y0, float y1, float* probs, float* dlogit,
                float* metrics, int is_g_step, hipStream_t s, float gscale = 1.0f, const float* parts = nullptr, int P = 0,
                const float* bc = nullptr);
// dv[n][hw][c] = dlogit[n] * wcp[hw*C+c] * leaky'(act) * noise[n][c]
// from the logits (rows < n0: target y0, the rest y1; each segment's mean): d(logit) is recomputed with k_bce's expression
void launch_cls_bwd(int dt, const float* logits, int n0, float y0, float y1, const float* wcp, const void* act, const float* noise,
                    float slope, void* dv, int B, int C, hipStream_t s, float gscale = 1.0f, float* bce_probs = nullptr,
                    float* bce_dlogit = nullptr, float* bce_metrics = nullptr, int bce_is_g = 0, bool with_bce = false,
                    const float* parts = nullptr, int P = 0, const float* bc = nullptr);
// dWc (torch order c*16+hw) and dbc
void launch_cls_wgrad(int dt, const float* dlogit, const void* act, float* dWc, float* dbc, int B, int C, hipStream_t s);
// dst[i] = (float)src[i] for a tensor of element type dt
void launch_to_f32(int dt, const void* src, float* dst, int64_t n, hipStream_t s);

// ---- spectral normalisation of the Discriminator (sn.hip) -------------------------------------------------------------
struct SnLayer {
    const float* W;              // weight_orig in the parameter arena, (rows, K) row-major = weight.view(Cout, -1)
    int rows, K;
    int u_off, v_off;            // offsets of this layer's u / v inside the flat u / v arenas (and of tbuf / wbuf)
    int64_t w_off;               // offset of the weight tensor inside the gradient arena
};
struct SnTable {
    static constexpr int MAXS = 8;          // conv blocks + classifier
    int n;
    SnLayer layer[MAXS];
    int pre_k[MAXS + 1], pre_r[MAXS + 1];   // prefix sums of ceil(K / 256) and ceil(rows / 4): workgroups of k_sn_wtu / k_sn_wv
    float *u, *v;                           // the caller's weight_u / weight_v buffers (flat, layer order)
    float *tbuf, *wbuf;                     // scratch: W^T u (v-sized), W v (u-sized)
    float* sig;                             // [3 slots][sigma | 1/sigma][MAXS]
    float *u_saved, *v_saved;               // [3 slots][u_total] / [3 slots][v_total]: each pass's (u, v)
    float* dots;                            // [2 passes][MAXS][64] partial <G_p, W>
    int u_total, v_total;
    int slot[2];                            // k_sn_combine: the sigma slot of pass 0 / pass 1
};
// sigma (+ one power iteration when training) of every layer for one pass, results into slot
void launch_sn_sigma(const SnTable& t, int training, int slot, float eps, hipStream_t s);
// out[arena] = sum over passes of the gradient w.r.t. weight_orig (through sigma) from the per-pass gradients w.r.t. the
// normalised weights g0 / g1 (whole-arena temporaries; non-weight entries are added plainly)
void launch_sn_combine(const SnTable& t, const float* g0, const float* g1, float* out, int64_t total, int npass, hipStream_t s);

// ---- optimiser ------------------------------------------------------------------------------
// launch_grad_sumsq: per-block partial sums of squares of the arena into `partial` (512 floats); whoever needs the norm adds
// them itself in one fixed order (k_adam_prepare when handed `sumsq_partial`, every block of the fused k_adam): no finalize launch.
// launch_adam_prepare: reads steps[0], writes steps[i] += 1 for every tensor, derives the Adam scalars; with clip_max_norm > 0
// (or check_finite) it needs launch_grad_sumsq first and its `partial` as sumsq_partial
void launch_grad_sumsq(const float* g, int64_t n, DevState* st, float* partial, hipStream_t s);
// check_finite (fp16 chains; needs launch_grad_sumsq first): a non-finite sum of squares marks the update as skipped
// (DevState::skip: k_adam returns at once, the step counts stay) and writes 1 to *metric_skipped, else 0
void launch_adam_prepare(DevState* st, float* steps, int ntensors, double lr, double beta1, double beta2,
                         float grad_scale, float clip_max_norm, float* metric_norm, hipStream_t s,
                         int check_finite = 0, float* metric_skipped = nullptr, const float* sumsq_partial = nullptr);
// beta^t for an integer-valued step count by repeated squaring in double: the SAME arithmetic on the host (fused path below)
// and on the device (k_adam_prepare), so the two paths give bit-identical bias corrections (within an ulp of libm's pow,
// which torch's Python-side `beta ** step` uses)
__host__ __device__ inline double pow_step(double b, double t) {
    unsigned long long n = (unsigned long long)t;
    double r = 1.0;
    while (n) { if (n & 1) r *= b; b *= b; n >>= 1; }
    return r;
}
// One-launch optimiser update (no k_adam_prepare in front): the caller knows the step count `t` (AFTER the increment) on the
// host and passes the bias-corrected scalars; the kernel derives the clip coefficient from DevState::sumsq itself, block 0
// writes steps[0..ntensors) = t, ticks the RNG epoch and stores the pre-clip norm.  Same update arithmetic as launch_adam.
void launch_adam_fused(float* p, float* g, float* m, float* v, int64_t n, DevState* st, float* steps, int ntensors, double t,
                       double lr, double beta1, double beta2, double eps, float grad_scale, float clip_max_norm,
                       float* metric_norm, const float* sumsq_partial, hipStream_t s, float* metric_skipped = nullptr);
void launch_adam(float* p, float* g, float* m, float* v, int64_t n, const DevState* st, double beta1,
                 double beta2, double eps, int write_back_grad, hipStream_t s);

// ---- the one-launch update that also rebuilds what the next pass derives from the arena ------------------------------
// launch_adam_fused plus launch_prepare's work in ONE launch: the arena is covered by a job table; a workgroup that owns a
// 16 x 16 x 16-tap tile of a convolution weight updates it and writes both MFMA packs from LDS (64-byte runs each way), the
// classifier / one-channel weights and the BatchNorm eval tables likewise.  Same update arithmetic, same pack values as the
// two launches it replaces (bitwise: tests/test_engine_gpu.py compares the execution modes).
enum ApType : int { AP_FLAT = 0, AP_CONV, AP_T16, AP_TAPS, AP_BN };
struct ApJob {
    int type;
    int A, Bc;                   // CONV: the weight is [A][Bc][4][4]; T16: A rows of 16; TAPS: A channels x Bc taps; BN: A channels, Bc = perm_c0
    int dt;                      // CONV: element type of the two packs
    long long off, n;            // first arena element, element count (BN: off = gamma)
    long long off2, n2;          // TAPS: a second flat range owned by the same workgroup (the layer's bias); BN: off2 = beta
    float* dst;                  // CONV: the pack whose unit is dim 0 (PREP_PACK_DOWN); T16 / TAPS: the permuted copy; BN: the table
    float* dst2;                 // CONV: the pack whose unit is dim 1 (PREP_PACK_UP)
    const float* rmean;          // BN: running statistics
    const float* rvar;
    int wait;                    // TAPS: riders that must have read this job's ranges before it writes them (ApRide), 0: none
};
struct ApTable {
    static constexpr int MAXJ = 24;
    int njobs, overflow;
    long long prefix[MAXJ + 1];  // prefix sums of the jobs' workgroup counts
    ApJob job[MAXJ];
};
void ap_add(ApTable& t, const ApJob& j);
// Riders: further workgroups of the same launch run the first Discriminator block's forward of B images with the block's
// UPDATED weights, which each of them derives for itself from the arena (1088 values); the workgroup that owns those ranges
// writes them only after every rider has read them (`counter`: one zero-initialised word, left at zero).
struct ApRide { const float* x; void* out; int B, S, dt; float slope; long long w_off, b_off; unsigned* counter; };
bool launch_adam_pack(const ApTable& t, float* p, float* g, float* m, float* v, DevState* st, float* steps, int ntensors,
                      double tstep, double lr, double beta1, double beta2, double eps, float grad_scale, float clip_max_norm,
                      float* metric_norm, const float* sumsq_partial, float* metric_skipped, float bn_eps,
                      const ApRide* ride, hipStream_t s);

// input pipeline: out[b] = lut[ resample(cache[index[b]]) ], (B,1,S,S) fp32 from an (N,S,S) uint8 cache (see k_augment)
void launch_augment(const uint8_t* cache, int64_t n_images, const int32_t* index, const int32_t* prm, const int16_t* tabs,
                    const float* lut, float* out, int B, int S, int augment, int fill, hipStream_t s);

}  // namespace siggan
