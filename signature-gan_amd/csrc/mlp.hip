// mlp.hip -- the fully-connected vanilla-GAN variant (include/siggan_mlp.h): a BUILD-DEFINED extension for BASELINE.json's
// configs[0] / configs[1]-as-worded; the reference has no such model, so this path is "parity unpinned" (checked against the
// build's own CPU restatement, oracle/mlp_oracle.py).  gfx950 only.
//
// Every dense product is one kernel, k_gemm<LAYOUT>, on v_mfma_f32_32x32x2_f32: 64 x 64 tiles, four waves of 32 x 32, K-tiles of
// 16 staged k-major in LDS ([k][64 + 4]: a lane's operand is one conflict-free ds_read_b32), operands zero-padded at the
// ragged edges (K = 100, N = 784 are not multiples of the tile).  The three layouts cover the forward product (NT: x W^T), the
// input gradient (NN: dy W) and the weight gradient (TN: dy^T x); bias and the activation ride in the epilogue.  BatchNorm1d,
// BCE and Adam are the conv engine's kernels (ops.hip).  These layers are tiny (<= 64 x 4096 x 512): the path is
// launch-bound, nothing here is tuned beyond being correct and matrix-core based.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <new>
#include <vector>

#include "../../include/siggan_mlp.h"
#include "ops.h"

using namespace siggan;

extern "C" const char* siggan_last_error(void);
int siggan_set_error(int code, const char* fmt, ...);     // siggan.hip
#define MFAIL(...) siggan_set_error(__VA_ARGS__)
#define MHIP(x)                                                                                     \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return MFAIL(SIGGAN_E_HIP, "%s -> %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

namespace {

enum Act : int { ACT_NONE = 0, ACT_LEAKY = 1, ACT_TANH = 2 };
struct GemmArgs {
    const float* A; const float* B; float* C;
    int M, N, K;
    const float* bias;      // [N] or nullptr
    int act; float slope;
};

// LAYOUT 0 NT: A (M,K), B (N,K);  1 NN: A (M,K), B (K,N);  2 TN: A (K,M), B (K,N)
template <int LAYOUT>
__global__ __launch_bounds__(256) void k_gemm(const GemmArgs g) {
    constexpr int BK = 16, LD = 64 + 4;
    __shared__ float sA[BK][LD], sB[BK][LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < g.K; k0 += BK) {
        // ---- stage A: element (m, k) ----
        if (LAYOUT == 2) {           // A (K, M): rows of m contiguous
            const int k = k0 + (tid >> 4), c = (tid & 15) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int m = m0 + c + j;
                sA[tid >> 4][c + j] = (k < g.K && m < g.M) ? g.A[(size_t)k * g.M + m] : 0.f;
            }
        } else {                     // A (M, K): rows of k contiguous
            const int m = m0 + (tid >> 2), kk = (tid & 3) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + kk + j;
                sA[kk + j][tid >> 2] = (k < g.K && m < g.M) ? g.A[(size_t)m * g.K + k] : 0.f;
            }
        }
        // ---- stage B: element (n, k) ----
        if (LAYOUT == 0) {           // B (N, K)
            const int n = n0 + (tid >> 2), kk = (tid & 3) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + kk + j;
                sB[kk + j][tid >> 2] = (k < g.K && n < g.N) ? g.B[(size_t)n * g.K + k] : 0.f;
            }
        } else {                     // B (K, N)
            const int k = k0 + (tid >> 4), c = (tid & 15) * 4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + c + j;
                sB[tid >> 4][c + j] = (k < g.K && n < g.N) ? g.B[(size_t)k * g.N + n] : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < BK / 2; ++s)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[2 * s + lh][wm * 32 + li], sB[2 * s + lh][wn * 32 + li], acc, 0, 0, 0);
        __syncthreads();
    }
    const int n = n0 + wn * 32 + li;
    if (n >= g.N) return;
    const float b = g.bias ? g.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= g.M) continue;
        float v = acc[r] + b;
        if (g.act == ACT_LEAKY) v = v > 0.f ? v : v * g.slope;
        else if (g.act == ACT_TANH) v = tanhf(v);
        g.C[(size_t)m * g.N + n] = v;
    }
}

void gemm(int layout, const float* A, const float* B, float* C, int M, int N, int K, const float* bias, int act, float slope,
          hipStream_t s) {
    GemmArgs g{A, B, C, M, N, K, bias, act, slope};
    const dim3 grid((N + 63) / 64, (M + 63) / 64);
    if (layout == 0) hipLaunchKernelGGL(k_gemm<0>, grid, dim3(256), 0, s, g);
    else if (layout == 1) hipLaunchKernelGGL(k_gemm<1>, grid, dim3(256), 0, s, g);
    else hipLaunchKernelGGL(k_gemm<2>, grid, dim3(256), 0, s, g);
}

// dpre = dy * leaky'(y) (y = stored activation) | dpre = dimg * (1 - img^2)
__global__ void k_act_bwd(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ out, int64_t n, int act,
                          float slope) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = y[i];
    out[i] = act == ACT_TANH ? dy[i] * (1.0f - v * v) : dy[i] * (v > 0.f ? 1.0f : slope);
}
// db[n] = sum_b d[b][n] (B <= a few hundred rows: one thread per column, coalesced across columns)
__global__ void k_colsum(const float* __restrict__ d, float* __restrict__ out, int B, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    float a = 0.f;
    for (int b = 0; b < B; ++b) a += d[(size_t)b * N + n];
    out[n] = a;
}
// eval-mode BatchNorm1d + ReLU from the running statistics
__global__ void k_bn_eval_relu(const float* __restrict__ y, float* __restrict__ a, int64_t n, int C, const float* __restrict__ gamma,
                               const float* __restrict__ beta, const float* __restrict__ rm, const float* __restrict__ rv, float eps) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const float sc = gamma[c] / sqrtf(rv[c] + eps);
    a[i] = fmaxf(fmaf(y[i] - rm[c], sc, beta[c]), 0.f);
}
inline unsigned blocks(int64_t n) { return (unsigned)((n + 255) / 256); }

}  // namespace

struct mlpgan_ctx {
    mlpgan_config cfg;
    int nh, P, Bm;                       // hidden layers, pixels per image, max batch
    int gdim[MLPGAN_MAX_HIDDEN + 2];     // latent, h_0 .. h_{nh-1}, P
    int ddim[MLPGAN_MAX_HIDDEN + 2];     // P, h_{nh-1} .. h_0, 1
    std::vector<int64_t> g_off, d_off;
    int64_t g_total, d_total, bn_total;
    int64_t bn_off[MLPGAN_MAX_HIDDEN];
    mlpgan_storage st; bool bound;
    char* ws;
    float *z, *img, *x2, *gy[MLPGAN_MAX_HIDDEN], *ga[MLPGAN_MAX_HIDDEN], *gda[MLPGAN_MAX_HIDDEN], *gbn[MLPGAN_MAX_HIDDEN];
    float *dh[MLPGAN_MAX_HIDDEN + 1], *ddv[MLPGAN_MAX_HIDDEN + 1], *dx, *dimg, *logits, *probs, *dlogit, *partial, *metrics;
    DevState* dev;
};

static const float MLP_BN_MOM = 0.1f, MLP_BN_EPS = 1e-5f;
// G tensor indices: hidden i -> 4i (w), 4i+1 (b), 4i+2 (bn w), 4i+3 (bn b); output 4nh, 4nh+1.  D layer j -> 2j, 2j+1.
#define MGP(c, i) ((c)->st.g_params + (c)->g_off[i])
#define MGG(c, i) ((c)->st.g_grads + (c)->g_off[i])
#define MDP(c, i) ((c)->st.d_params + (c)->d_off[i])
#define MDG(c, i) ((c)->st.d_grads + (c)->d_off[i])

extern "C" int mlpgan_create(const mlpgan_config* cfg, mlpgan_ctx** out) {
    if (!cfg || !out) return MFAIL(SIGGAN_E_INVALID, "null argument");
    if (cfg->n_hidden < 1 || cfg->n_hidden > MLPGAN_MAX_HIDDEN) return MFAIL(SIGGAN_E_INVALID, "n_hidden out of range");
    if (cfg->image_size < 4 || cfg->image_size > 128 || cfg->latent_dim < 1 || cfg->latent_dim > 4096 || cfg->max_batch < 1 ||
        cfg->max_batch > 4096)
        return MFAIL(SIGGAN_E_INVALID, "bad geometry");
    for (int i = 0; i < cfg->n_hidden; ++i)
        if (cfg->hidden[i] < 4 || cfg->hidden[i] > 8192 || (cfg->hidden[i] & 3)) return MFAIL(SIGGAN_E_INVALID, "hidden widths must be multiples of 4 in [4, 8192]");
    MHIP(hipSetDevice(cfg->device));
    mlpgan_ctx* c = new (std::nothrow) mlpgan_ctx();
    if (!c) return MFAIL(SIGGAN_E_NOMEM, "out of host memory");
    c->cfg = *cfg; c->nh = cfg->n_hidden; c->P = cfg->image_size * cfg->image_size; c->Bm = cfg->max_batch; c->bound = false;
    c->gdim[0] = cfg->latent_dim;
    for (int i = 0; i < c->nh; ++i) c->gdim[i + 1] = cfg->hidden[i];
    c->gdim[c->nh + 1] = c->P;
    c->ddim[0] = c->P;
    for (int i = 0; i < c->nh; ++i) c->ddim[i + 1] = cfg->hidden[c->nh - 1 - i];
    c->ddim[c->nh + 1] = 1;
    c->g_total = c->d_total = c->bn_total = 0;
    auto push = [](std::vector<int64_t>& o, int64_t& t, int64_t n) { o.push_back(t); t += n; };
    for (int i = 0; i < c->nh; ++i) {
        push(c->g_off, c->g_total, (int64_t)c->gdim[i + 1] * c->gdim[i]); push(c->g_off, c->g_total, c->gdim[i + 1]);
        push(c->g_off, c->g_total, c->gdim[i + 1]); push(c->g_off, c->g_total, c->gdim[i + 1]);
        c->bn_off[i] = c->bn_total; c->bn_total += c->gdim[i + 1];
    }
    push(c->g_off, c->g_total, (int64_t)c->P * c->gdim[c->nh]); push(c->g_off, c->g_total, c->P);
    for (int j = 0; j <= c->nh; ++j) { push(c->d_off, c->d_total, (int64_t)c->ddim[j + 1] * c->ddim[j]); push(c->d_off, c->d_total, c->ddim[j + 1]); }
    const int64_t Bm = c->Bm, B2 = 2 * Bm;
    size_t off = 0; char* base = nullptr;
    auto carve = [&](float** p, int64_t n) { if (base) *p = (float*)(base + off); off += ((size_t)n * 4 + 255) & ~(size_t)255; };
    for (int pass = 0; pass < 2; ++pass) {
        off = 0;
        carve(&c->z, Bm * c->gdim[0]); carve(&c->img, Bm * c->P); carve(&c->x2, B2 * c->P); carve(&c->dx, B2 * c->P); carve(&c->dimg, Bm * c->P);
        for (int i = 0; i < c->nh; ++i) {
            carve(&c->gy[i], Bm * c->gdim[i + 1]); carve(&c->ga[i], Bm * c->gdim[i + 1]); carve(&c->gda[i], Bm * c->gdim[i + 1]);
            carve(&c->gbn[i], 6 * (int64_t)c->gdim[i + 1]);
        }
        for (int j = 0; j <= c->nh; ++j) { carve(&c->dh[j], B2 * c->ddim[j + 1]); carve(&c->ddv[j], B2 * c->ddim[j + 1]); }
        carve(&c->logits, B2); carve(&c->probs, B2); carve(&c->dlogit, B2);
        carve(&c->partial, (int64_t)1 << 20); carve(&c->metrics, SIGGAN_M_COUNT);
        float* devp = nullptr; carve(&devp, 64);
        if (pass == 1) c->dev = (DevState*)devp;
        if (pass == 0) {
            hipError_t e = hipMalloc((void**)&base, off);
            if (e != hipSuccess) { delete c; return MFAIL(SIGGAN_E_NOMEM, "hipMalloc(%zu) -> %s", off, hipGetErrorString(e)); }
            c->ws = base;
            (void)hipMemset(base, 0, off);
        }
    }
    DevState h; memset(&h, 0, sizeof h); h.seed = cfg->seed; h.grad_mul = 1.f;
    MHIP(hipMemcpy(c->dev, &h, sizeof h, hipMemcpyHostToDevice));
    *out = c;
    return SIGGAN_OK;
}
extern "C" int mlpgan_destroy(mlpgan_ctx* c) {
    if (!c) return SIGGAN_OK;
    (void)hipSetDevice(c->cfg.device); (void)hipDeviceSynchronize();
    if (c->ws) (void)hipFree(c->ws);
    delete c;
    return SIGGAN_OK;
}
extern "C" int64_t mlpgan_param_count(const mlpgan_ctx* c, int which) { return !c ? -1 : (which == 0 ? c->g_total : c->d_total); }
extern "C" int32_t mlpgan_param_tensors(const mlpgan_ctx* c, int which) { return !c ? -1 : (int32_t)(which == 0 ? c->g_off.size() : c->d_off.size()); }
extern "C" int64_t mlpgan_bn_count(const mlpgan_ctx* c) { return c ? c->bn_total : -1; }
extern "C" int mlpgan_bind(mlpgan_ctx* c, const mlpgan_storage* st) {
    if (!c || !st) return MFAIL(SIGGAN_E_INVALID, "null argument");
    if (!st->g_params || !st->d_params || !st->g_bn_running_mean || !st->g_bn_running_var) return MFAIL(SIGGAN_E_INVALID, "mlpgan_bind: parameters and BatchNorm buffers are required");
    c->st = *st; c->bound = true;
    return SIGGAN_OK;
}
extern "C" int mlpgan_seed(mlpgan_ctx* c, uint64_t seed, uint64_t offset) {
    if (!c) return MFAIL(SIGGAN_E_INVALID, "null context");
    unsigned long long v[2] = {seed, offset};
    MHIP(hipMemcpy(c->dev, v, sizeof v, hipMemcpyHostToDevice));
    return SIGGAN_OK;
}
static int mcheck(mlpgan_ctx* c, int B) {
    if (!c) return MFAIL(SIGGAN_E_INVALID, "null context");
    if (!c->bound) return MFAIL(SIGGAN_E_STATE, "mlpgan_bind has not been called");
    if (B < 1 || B > c->Bm) return MFAIL(SIGGAN_E_INVALID, "batch %d outside [1, max_batch=%d]", B, c->Bm);
    hipError_t e = hipSetDevice(c->cfg.device);
    if (e != hipSuccess) return MFAIL(SIGGAN_E_HIP, "hipSetDevice -> %s", hipGetErrorString(e));
    return SIGGAN_OK;
}

// Generator forward; training keeps y_i / a_i and the BatchNorm tables for the backward pass
static void g_fwd(mlpgan_ctx* c, const float* z, int B, bool training, float* img, hipStream_t s) {
    const float* x = z;
    for (int i = 0; i < c->nh; ++i) {
        const int N = c->gdim[i + 1], K = c->gdim[i];
        gemm(0, x, MGP(c, 4 * i), c->gy[i], B, N, K, MGP(c, 4 * i + 1), ACT_NONE, 0.f, s);
        if (training) {
            launch_bn_train_stats(DT_F32, c->gy[i], B, N, MGP(c, 4 * i + 2), MGP(c, 4 * i + 3), c->st.g_bn_running_mean + c->bn_off[i],
                                  c->st.g_bn_running_var + c->bn_off[i], c->st.g_bn_batches ? c->st.g_bn_batches + i : nullptr, c->gbn[i],
                                  c->partial, 0, MLP_BN_MOM, MLP_BN_EPS, s);
            launch_bn_relu(DT_F32, c->gy[i], c->ga[i], B, N, c->gbn[i], s);
        } else {
            hipLaunchKernelGGL(k_bn_eval_relu, dim3(blocks((int64_t)B * N)), dim3(256), 0, s, c->gy[i], c->ga[i], (int64_t)B * N, N,
                               MGP(c, 4 * i + 2), MGP(c, 4 * i + 3), c->st.g_bn_running_mean + c->bn_off[i],
                               c->st.g_bn_running_var + c->bn_off[i], MLP_BN_EPS);
        }
        x = c->ga[i];
    }
    gemm(0, x, MGP(c, 4 * c->nh), img, B, c->P, c->gdim[c->nh], MGP(c, 4 * c->nh + 1), ACT_TANH, 0.f, s);
}
// Discriminator forward over R rows of x (R = B or 2B): activations kept in dh[j], logits in c->logits
static void d_fwd(mlpgan_ctx* c, const float* x, int R, hipStream_t s) {
    const float* in = x;
    for (int j = 0; j < c->nh; ++j) {
        gemm(0, in, MDP(c, 2 * j), c->dh[j], R, c->ddim[j + 1], c->ddim[j], MDP(c, 2 * j + 1), ACT_LEAKY, c->cfg.leaky_slope, s);
        in = c->dh[j];
    }
    gemm(0, in, MDP(c, 2 * c->nh), c->logits, R, 1, c->ddim[c->nh], MDP(c, 2 * c->nh + 1), ACT_NONE, 0.f, s);
}
// Discriminator backward from c->dlogit (R rows); wgrad: fill the D gradient arena; dimage: leave d(image) in c->dx
static void d_bwd(mlpgan_ctx* c, const float* x, int R, bool wgrad, bool dimage, hipStream_t s) {
    const float* dpre = c->dlogit;                       // (R, 1)
    for (int j = c->nh; j >= 0; --j) {
        const int N = c->ddim[j + 1], K = c->ddim[j];
        const float* in = j == 0 ? x : c->dh[j - 1];
        if (wgrad) {
            gemm(2, dpre, in, MDG(c, 2 * j), N, K, R, nullptr, ACT_NONE, 0.f, s);          // dW (N, K) = dpre^T in
            hipLaunchKernelGGL(k_colsum, dim3(blocks(N)), dim3(256), 0, s, dpre, MDG(c, 2 * j + 1), R, N);
        }
        if (j == 0 && !dimage) break;
        float* dxin = j == 0 ? c->dx : c->ddv[j - 1];
        gemm(1, dpre, MDP(c, 2 * j), dxin, R, K, N, nullptr, ACT_NONE, 0.f, s);            // d(input) (R, K) = dpre W
        if (j > 0) {
            hipLaunchKernelGGL(k_act_bwd, dim3(blocks((int64_t)R * K)), dim3(256), 0, s, dxin, c->dh[j - 1], dxin, (int64_t)R * K,
                               (int)ACT_LEAKY, c->cfg.leaky_slope);
            dpre = dxin;
        }
    }
}
static void adam(mlpgan_ctx* c, int which, const siggan_hyper* hp, float* mt, hipStream_t s) {
    float* p = which == 0 ? c->st.g_params : c->st.d_params; float* g = which == 0 ? c->st.g_grads : c->st.d_grads;
    float* m = which == 0 ? c->st.g_exp_avg : c->st.d_exp_avg; float* v = which == 0 ? c->st.g_exp_avg_sq : c->st.d_exp_avg_sq;
    float* steps = which == 0 ? c->st.g_adam_steps : c->st.d_adam_steps;
    const int64_t n = which == 0 ? c->g_total : c->d_total;
    const int nt = (int)(which == 0 ? c->g_off.size() : c->d_off.size());
    const bool clip = hp->clip_max_norm > 0.f;
    const float gs = hp->grad_scale > 0.f ? hp->grad_scale : 1.0f;
    if (clip) launch_grad_sumsq(g, n, c->dev, c->partial, s);
    launch_adam_prepare(c->dev, steps, nt, hp->lr, hp->beta1, hp->beta2, gs, clip ? hp->clip_max_norm : 0.f,
                        mt + (which == 0 ? SIGGAN_M_G_GRAD_NORM : SIGGAN_M_D_GRAD_NORM), s, 0, nullptr, clip ? c->partial : nullptr);
    launch_adam(p, g, m, v, n, c->dev, hp->beta1, hp->beta2, hp->eps, (clip || gs != 1.0f) ? 1 : 0, s);
}
static int need_train_arenas(const mlpgan_ctx* c, int which) {
    const float* a[] = {which ? c->st.d_grads : c->st.g_grads, which ? c->st.d_exp_avg : c->st.g_exp_avg,
                        which ? c->st.d_exp_avg_sq : c->st.g_exp_avg_sq, which ? c->st.d_adam_steps : c->st.g_adam_steps};
    for (const float* p : a) if (!p) return MFAIL(SIGGAN_E_STATE, "gradient / Adam arenas were not bound");
    return SIGGAN_OK;
}

extern "C" int mlpgan_g_forward(mlpgan_ctx* c, const float* z, int32_t B, int32_t training, float* img, void* stream) {
    int rc = mcheck(c, B); if (rc) return rc;
    if (!z || !img) return MFAIL(SIGGAN_E_INVALID, "null tensor");
    g_fwd(c, z, B, training != 0, img, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? SIGGAN_OK : MFAIL(SIGGAN_E_HIP, "kernel launch failed");
}
extern "C" int mlpgan_d_forward(mlpgan_ctx* c, const float* x, int32_t B, float* probs, void* stream) {
    int rc = mcheck(c, B); if (rc) return rc;
    if (!x || !probs) return MFAIL(SIGGAN_E_INVALID, "null tensor");
    hipStream_t s = (hipStream_t)stream;
    d_fwd(c, x, B, s);
    launch_bce(c->logits, B, B, 0.f, 0.f, probs, nullptr, nullptr, 0, s);
    return hipGetLastError() == hipSuccess ? SIGGAN_OK : MFAIL(SIGGAN_E_HIP, "kernel launch failed");
}
extern "C" int mlpgan_d_step(mlpgan_ctx* c, const float* real, int32_t B, const float* z, const siggan_hyper* hp, float* mt, void* stream) {
    int rc = mcheck(c, B); if (rc) return rc;
    if (!real || !hp) return MFAIL(SIGGAN_E_INVALID, "null argument");
    if ((rc = need_train_arenas(c, 1))) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (!mt) mt = c->metrics;
    const size_t ib = (size_t)B * c->P * sizeof(float);
    if (z) MHIP(hipMemcpyAsync(c->z, z, (size_t)B * c->gdim[0] * sizeof(float), hipMemcpyDeviceToDevice, s));
    else launch_randn(c->z, (int64_t)B * c->gdim[0], c->dev, 1, s);
    MHIP(hipMemcpyAsync(c->x2, real, ib, hipMemcpyDeviceToDevice, s));             // rows [0, B): real
    g_fwd(c, c->z, B, false, c->x2 + (size_t)B * c->P, s);                          // rows [B, 2B): G.eval()(z), no grad
    d_fwd(c, c->x2, 2 * B, s);                                                      // no BatchNorm in D: one 2B-row pass
    launch_bce(c->logits, 2 * B, B, hp->label_smoothing, 0.f, c->probs, c->dlogit, mt, 0, s);
    d_bwd(c, c->x2, 2 * B, true, false, s);
    adam(c, 1, hp, mt, s);
    return hipGetLastError() == hipSuccess ? SIGGAN_OK : MFAIL(SIGGAN_E_HIP, "kernel launch failed");
}
extern "C" int mlpgan_g_step(mlpgan_ctx* c, int32_t B, const float* z, const siggan_hyper* hp, float* mt, void* stream) {
    int rc = mcheck(c, B); if (rc) return rc;
    if (!hp) return MFAIL(SIGGAN_E_INVALID, "null argument");
    if ((rc = need_train_arenas(c, 0))) return rc;
    hipStream_t s = (hipStream_t)stream;
    if (!mt) mt = c->metrics;
    if (z) MHIP(hipMemcpyAsync(c->z, z, (size_t)B * c->gdim[0] * sizeof(float), hipMemcpyDeviceToDevice, s));
    else launch_randn(c->z, (int64_t)B * c->gdim[0], c->dev, 2, s);
    g_fwd(c, c->z, B, true, c->img, s);                                             // G.train(): BatchNorm batch statistics
    d_fwd(c, c->img, B, s);
    launch_bce(c->logits, B, B, 1.0f, 1.0f, c->probs, c->dlogit, mt, 1, s);
    d_bwd(c, c->img, B, false, true, s);                                            // through D into the image; no D weight grads
    hipLaunchKernelGGL(k_act_bwd, dim3(blocks((int64_t)B * c->P)), dim3(256), 0, s, c->dx, c->img, c->dimg, (int64_t)B * c->P, (int)ACT_TANH, 0.f);
    const int nh = c->nh;
    const float* dpre = c->dimg;                                                    // (B, P): d(pre-tanh)
    gemm(2, dpre, c->ga[nh - 1], MGG(c, 4 * nh), c->P, c->gdim[nh], B, nullptr, ACT_NONE, 0.f, s);
    hipLaunchKernelGGL(k_colsum, dim3(blocks(c->P)), dim3(256), 0, s, dpre, MGG(c, 4 * nh + 1), B, c->P);
    gemm(1, dpre, MGP(c, 4 * nh), c->gda[nh - 1], B, c->gdim[nh], c->P, nullptr, ACT_NONE, 0.f, s);
    for (int i = nh - 1; i >= 0; --i) {
        const int N = c->gdim[i + 1], K = c->gdim[i];
        launch_bn_bwd(DT_F32, c->gda[i], c->gy[i], B, N, c->gbn[i], c->partial, MGG(c, 4 * i + 2), MGG(c, 4 * i + 3), 0, s);   // gda[i] <- d(y_i)
        gemm(2, c->gda[i], i == 0 ? c->z : c->ga[i - 1], MGG(c, 4 * i), N, K, B, nullptr, ACT_NONE, 0.f, s);
        hipLaunchKernelGGL(k_colsum, dim3(blocks(N)), dim3(256), 0, s, c->gda[i], MGG(c, 4 * i + 1), B, N);
        if (i > 0) gemm(1, c->gda[i], MGP(c, 4 * i), c->gda[i - 1], B, K, N, nullptr, ACT_NONE, 0.f, s);
    }
    adam(c, 0, hp, mt, s);
    return hipGetLastError() == hipSuccess ? SIGGAN_OK : MFAIL(SIGGAN_E_HIP, "kernel launch failed");
}
extern "C" int mlpgan_op_gemm(int32_t device, int32_t layout, const float* a, const float* b, float* cc, int32_t m, int32_t n, int32_t k,
                              void* stream) {
    if (!a || !b || !cc || m < 1 || n < 1 || k < 1 || layout < 0 || layout > 2) return MFAIL(SIGGAN_E_INVALID, "bad argument");
    MHIP(hipSetDevice(device));
    gemm(layout, a, b, cc, m, n, k, nullptr, ACT_NONE, 0.f, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? SIGGAN_OK : MFAIL(SIGGAN_E_HIP, "kernel launch failed");
}
