// siggan.hip -- C ABI (include/siggan.h) and step orchestration of the MI355X signature-GAN engine.
//
// One context = one GPU.  The caller (PyTorch-ROCm tensors, through ctypes) owns parameters,
// gradients, Adam moments and BatchNorm buffers as flat fp32 arenas in the reference's
// parameters() order; the library owns only the workspace allocated here (NHWC activations,
// packed weight copies, split-K slabs, reduction partials, device-side step state).
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>

#include "../../include/siggan.h"
#include "gconv.h"
#include "ops.h"

using namespace siggan;

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
// (mlp.hip reports through the same thread-local message)
int siggan_set_error(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return fail(SIGGAN_E_HIP, "%s -> %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)
#define LAUNCHCHK()                                                                                 \
    do {                                                                                            \
        hipError_t e_ = hipGetLastError();                                                          \
        if (e_ != hipSuccess) return fail(SIGGAN_E_HIP, "kernel launch -> %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// Entry points run on the context's device and put the caller's current device back on return (a process whose
// torch current device is another GPU must not find it switched behind its back).
struct DevGuard {
    int prev = -1, dev;
    hipError_t err = hipSuccess;
    explicit DevGuard(int d) : dev(d) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev);
    }
    ~DevGuard() { if (prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
};

static const float BN_MOMENTUM = 0.1f, BN_EPS = 1e-5f;   // nn.BatchNorm defaults (generator_vanilla_gan.py:58,126)
static const int MAXL = 6;
static const int64_t PARTIAL_FLOATS = (int64_t)2 << 20;   // each of the three reduction-partial regions (partial, partial_b, partial_c)
// Events that order the library's own lanes against each other on ONE device carry no system-scope fence: with the default
// flags every record writes back / invalidates caches for the host and for peer devices, which showed up as a ~6 us hole
// behind each fork on the recording lane (-1.5 % step time at fp32, -3.8 % at bf16 without it).  Kernel boundaries
// still release at agent scope, which is all a kernel on another lane of the same device needs.  The events in front of an
// RCCL collective (peer devices read the gradient arena over xGMI) keep the fence: while a communicator exists the lanes use
// the fenced ring (ev_fenced), and ev_sys / ev_ar are always fenced.
#define EV_FLAGS (hipEventDisableTiming | hipEventDisableSystemFence)
#define ENTER(c)                                                     \
    if (!(c)) return fail(SIGGAN_E_INVALID, "null context");          \
    DevGuard dg_((c)->cfg.device);                                    \
    HIPCHK(dg_.err)

static int ilog2i(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// ------------------------------------------------------------------------------------------
// RCCL (the collective of the data-parallel step).  The library does not link librccl: the
// process that loads it normally holds one already (torch bundles its own copy), and a second HIP/RCCL runtime in one
// process is what SURVEY 7 warns about.  The five entry points used are resolved with dlsym from the copy that is
// loaded (RTLD_NOLOAD), else from the first librccl the loader finds.  Declarations follow the stable NCCL C API.
// ------------------------------------------------------------------------------------------
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId_t;
enum { NCCL_SUCCESS = 0, NCCL_CHAR = 0, NCCL_FLOAT32 = 7, NCCL_SUM = 0 };
struct Rccl {
    void* h = nullptr;
    int (*GetUniqueId)(ncclUniqueId_t*) = nullptr;
    int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId_t, int) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    const char* err = nullptr;
};
static Rccl* rccl() {
    static Rccl r;
    if (r.h || r.err) return &r;
    const char* names[] = {"librccl.so.1", "librccl.so"};
    for (const char* n : names) if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);      // the copy torch loaded
    for (const char* n : names) if (!r.h) r.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!r.h) { r.err = "librccl.so was not found (load it, or import torch, before siggan_comm_*)"; return &r; }
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.h, "ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.h, "ncclCommDestroy");
    r.AllReduce = (decltype(r.AllReduce))dlsym(r.h, "ncclAllReduce");
    r.Broadcast = (decltype(r.Broadcast))dlsym(r.h, "ncclBroadcast");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.h, "ncclGetErrorString");
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllReduce || !r.Broadcast || !r.GetErrorString)
        r.err = "librccl.so lacks one of ncclGetUniqueId / CommInitRank / CommDestroy / AllReduce / Broadcast / GetErrorString";
    return &r;
}
#define NCCLCHK(x)                                                                                  \
    do {                                                                                            \
        int e_ = (x);                                                                               \
        if (e_ != NCCL_SUCCESS) return fail(SIGGAN_E_HIP, "%s -> %s (%s:%d)", #x, rccl()->GetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct PhaseKey {
    int phase, B, has_z, has_masks, g_dirty, d_dirty, spec_g, has_zg, pre_real, variant, coll;   // coll: an apply call follows (collectives may start early)
    float* mt;   // where the phase writes its metrics (caller's buffer, or the workspace one)
    double lr, beta1, beta2, eps;
    double fused_t;   // > 0: the step count (after the increment) of a one-launch optimiser update; 0: k_adam_prepare path
    int pack;         // apply: that launch also rebuilds the weight packs / eval tables (k_adam_pack)
    int ride;         // D apply: ... and runs the first Discriminator block of the pending G step's `ride` images (its riders)
    int rode;         // G grads: the preceding D apply did
    float ls, clip, gs;
    bool operator==(const PhaseKey& o) const {
        return phase == o.phase && B == o.B && has_z == o.has_z && has_masks == o.has_masks && g_dirty == o.g_dirty &&
               d_dirty == o.d_dirty && spec_g == o.spec_g && has_zg == o.has_zg && pre_real == o.pre_real && variant == o.variant && coll == o.coll && mt == o.mt && lr == o.lr && beta1 == o.beta1 && beta2 == o.beta2 && eps == o.eps && ls == o.ls &&
               clip == o.clip && gs == o.gs && fused_t == o.fused_t && pack == o.pack && ride == o.ride && rode == o.rode;
    }
};

struct siggan_ctx {
    siggan_config cfg;
    int S, latent, Lg, Ld, Bm;
    int dt;             // element type of the activation / gradient tensors and of the MFMA weight packs (act.h)
    size_t es;          // its size in bytes
    bool sn;            // spectral normalisation of every Discriminator weight (sn.hip)
    SnTable snt;        // its layer table (pointers into the bound arenas are refreshed by siggan_bind)
    float *sn_tbuf, *sn_wbuf, *sn_sig, *sn_us, *sn_vs, *sn_dots, *sn_g[2], *d_w1s;
    int variant;        // SIGGAN_STEP_TRAINER / SIGGAN_STEP_ABLATION (siggan_set_step_variant)
    bool abl_masks;     // ablation step: the D half was given explicit masks, the third set waits in mask_stage for the G half
    bool fc_fused;      // Generator.fc runs as the one-launch MFMA kernels of fc.hip (max_batch <= 256, latent % 4 == 0)
    float gscale;       // gradient scale of the backward chains (fp16: keeps small gradients out of the subnormals; else 1)
    int gC[MAXL + 1];   // generator channel chain gC[0..Lg]  (generator_vanilla_gan.py:131-149)
    int dC[MAXL + 1];   // discriminator chain dC[0]=1, dC[1..Ld] (discriminator_vanilla_gan.py:131-194)
    int F;              // gC[0]*16
    // flat-arena spans (offset, numel) in parameters() order
    std::vector<int64_t> g_off, g_num, d_off, d_num;
    int64_t g_total, d_total, bn_total;
    int64_t g_bn_off[MAXL + 1];
    siggan_storage st;
    bool bound;
    bool g_dirty, d_dirty;
    // workspace
    char* ws; size_t ws_bytes;
    // (element type dt: fc_y, g_y, g_a, g_da, d_a, d_dv and the MFMA weight packs g_up, g_dn, d_dn, d_up; fp32: the rest)
    float *z, *g_bn[MAXL + 1], *g_bne[MAXL + 1];
    char *fc_y, *g_y[MAXL + 1], *g_a[MAXL + 1], *g_da[MAXL + 1];
    char *g_ae[MAXL + 1];     // 16-bit contexts: activations of the EVAL-mode Generator forward (it runs beside the training forward)
    float *img, *dpre;
    char *d_a[MAXL + 1], *d_dv[MAXL + 1];
    float *d_noise[MAXL + 1];
    float *logits, *probs, *dlogit;
    float* lparts;       // partial classifier dot products left by the last block's split-K epilogue: lP[h] per image of workspace half h
    int lP[2];           //   (0: that half's logits are in `logits`, written by k_cls_fwd)
    char *g_up[MAXL + 1], *g_dn[MAXL + 1], *d_dn[MAXL + 1], *d_up[MAXL + 1];
    float *wcp;
    float *slab, *slab_k, *slab_k2, *slab_k3, *partial, *partial_b, *partial_c, *z_g, *img_g, *metrics, *zeros, *wfc_t, *wfin_t, *d_w1t, *real_stage, *mask_stage;
    char *op_pack;
    int64_t slab_floats, slab_k_floats;
    DevState* dev;
    // last *_grads call (for *_apply)
    int pending;   // 0 none, 1 D, 2 G
    float* metrics_last;   // metrics target of the last *_grads call
    // lanes / graphs
    static constexpr int NEV = 96;
    int mode;
    hipStream_t s_m, s_a, s_b, s_c, s_n;      // s_n: the lane of an all-reduce issued while the backward pass is still running
    hipEvent_t ev_gfwd, ev_dreal, ev_ar, ev_sys[2];
    bool early_ar;       // the last D block's weight gradient is already being all-reduced on s_n (ev_ar marks its end)
    bool dreal_orphan;   // a D(real) forward enqueued on lane c was abandoned: the next enqueue waits for ev_dreal first
    hipError_t lane_err; // first failed event record / wait of a fork or join (checked after every phase)
    // data-parallel communicator (siggan_comm_init): world 1 = none
    ncclComm_t comm; int comm_rank, comm_world; int comm_err;
    int staged_B;        // batch of a real batch staged for the NEXT D step by siggan_stage_real (0: none)
    bool gfwd_joined;    // the pipelined Generator forward's lane was already joined into the caller's stream (end of the D grads phase)
    bool dreal_joined;   // the lane of that early forward was already joined into the caller's stream (end of the G step)
    bool dreal_noise2;   // its launch also drew the dropout tables of the D(fake) pass
    float* slab_b;       // second weight-gradient slab region (a weight gradient on the main lane beside lane a's)
    const float* staged_src;   // where that batch lies: the caller's own tensor (borrowed until the D step that consumes it)
    int dreal_B;         // batch whose D(real) forward siggan_g_grads already enqueued on lane c (0: none)
    int zg_stash;        // batch of an explicit G-step z handed to siggan_step_begin when the forward was not pipelined
    int g_fwd_pending;   // batch of a Generator training forward already enqueued by siggan_step_begin (0: none)
    int conv1_rode;      // batch whose first-Discriminator-block forward (workspace rows [B, 2B), updated weights) the last D apply ran
    float* ride_ctr;     // k_adam_pack's rider count (one word, zero between launches)
    double adam_t[2];    // step count of the network's Adam state as the HOST knows it ([0] G, [1] D); valid while adam_t_known
    bool adam_t_known[2];// (reset by siggan_bind / siggan_params_changed: the caller may have written the step tensors)
    int g_r0;            // first workspace row of the Discriminator pass of the last G step (0, or B: beside an early D(real))
    int ga_last_B;       // batch of the last TRAINING Generator forward: the last block's activation was not materialised
                         // (siggan_debug_tensor("g_a", Lg) forms it on demand)
    hipEvent_t ev[NEV], ev_fenced[NEV], ev_bridge[2];   // ev_fenced: the same ring with the system-scope fence (used while a communicator exists)
    int evi;
    std::vector<std::pair<PhaseKey, hipGraphExec_t>> graphs;
};

// parameter tensor indices
static inline int gi_fc_w() { return 0; }
static inline int gi_fc_b() { return 1; }
static inline int gi_bn0_w() { return 2; }
static inline int gi_bn0_b() { return 3; }
static inline int gi_up_w(int l) { return 4 + 3 * (l - 1); }       // l = 1..Lg
static inline int gi_bn_w(int l) { return l == 0 ? 2 : 5 + 3 * (l - 1); }
static inline int gi_bn_b(int l) { return l == 0 ? 3 : 6 + 3 * (l - 1); }
static inline int gi_fin_w(const siggan_ctx* c) { return 4 + 3 * c->Lg; }
static inline int gi_fin_b(const siggan_ctx* c) { return 5 + 3 * c->Lg; }
static inline int di_w(int l) { return 2 * (l - 1); }              // l = 1..Ld
static inline int di_b(int l) { return 2 * (l - 1) + 1; }
static inline int di_cls_w(const siggan_ctx* c) { return 2 * c->Ld; }
static inline int di_cls_b(const siggan_ctx* c) { return 2 * c->Ld + 1; }

#define GP(c, i) ((c)->st.g_params + (c)->g_off[i])
#define GG(c, i) ((c)->st.g_grads + (c)->g_off[i])
#define DP(c, i) ((c)->st.d_params + (c)->d_off[i])
#define DG(c, i) ((c)->st.d_grads + (c)->d_off[i])

extern "C" int siggan_abi_version(void) { return SIGGAN_ABI_VERSION; }
extern "C" const char* siggan_last_error(void) { return g_err; }

static void build_layout(siggan_ctx* c) {
    static const int g64[] = {256, 128, 64, 32, 32}, g128[] = {512, 256, 128, 64, 32, 32};
    static const int d64[] = {1, 64, 128, 256, 512}, d128[] = {1, 64, 128, 256, 512, 512};
    if (c->S == 64) { c->Lg = 4; c->Ld = 4; memcpy(c->gC, g64, sizeof g64); memcpy(c->dC, d64, sizeof d64); }
    else            { c->Lg = 5; c->Ld = 5; memcpy(c->gC, g128, sizeof g128); memcpy(c->dC, d128, sizeof d128); }
    c->F = c->gC[0] * 16;
    auto push = [](std::vector<int64_t>& off, std::vector<int64_t>& num, int64_t& tot, int64_t n) {
        off.push_back(tot); num.push_back(n); tot += n;
    };
    c->g_total = 0;
    push(c->g_off, c->g_num, c->g_total, (int64_t)c->F * c->latent);
    push(c->g_off, c->g_num, c->g_total, c->F);
    push(c->g_off, c->g_num, c->g_total, c->F);
    push(c->g_off, c->g_num, c->g_total, c->F);
    for (int l = 1; l <= c->Lg; ++l) {
        push(c->g_off, c->g_num, c->g_total, (int64_t)c->gC[l - 1] * c->gC[l] * 16);
        push(c->g_off, c->g_num, c->g_total, c->gC[l]);
        push(c->g_off, c->g_num, c->g_total, c->gC[l]);
    }
    push(c->g_off, c->g_num, c->g_total, (int64_t)c->gC[c->Lg] * 9);
    push(c->g_off, c->g_num, c->g_total, 1);
    c->d_total = 0;
    for (int l = 1; l <= c->Ld; ++l) {
        push(c->d_off, c->d_num, c->d_total, (int64_t)c->dC[l] * c->dC[l - 1] * 16);
        push(c->d_off, c->d_num, c->d_total, c->dC[l]);
    }
    push(c->d_off, c->d_num, c->d_total, (int64_t)c->dC[c->Ld] * 16);
    push(c->d_off, c->d_num, c->d_total, 1);
    c->bn_total = 0;
    c->g_bn_off[0] = 0; c->bn_total = c->F;
    for (int l = 1; l <= c->Lg; ++l) { c->g_bn_off[l] = c->bn_total; c->bn_total += c->gC[l]; }
}

extern "C" int siggan_create(const siggan_config* cfg, siggan_ctx** out) {
    if (!cfg || !out) return fail(SIGGAN_E_INVALID, "null argument");
    if (cfg->image_size != 64 && cfg->image_size != 128)
        return fail(SIGGAN_E_INVALID, "output_size must be 64 or 128, got %d", cfg->image_size);
    if (cfg->image_channels != 1) return fail(SIGGAN_E_INVALID, "only image_channels == 1 is built, got %d", cfg->image_channels);
    if (cfg->latent_dim < 1 || cfg->latent_dim > 4096) return fail(SIGGAN_E_INVALID, "latent_dim out of range: %d", cfg->latent_dim);
    if (cfg->max_batch < 1 || cfg->max_batch > 4096) return fail(SIGGAN_E_INVALID, "max_batch out of range: %d", cfg->max_batch);
    if (!(cfg->dropout >= 0.f && cfg->dropout < 1.f)) return fail(SIGGAN_E_INVALID, "dropout must be in [0,1)");
    if (cfg->dtype != SIGGAN_DTYPE_F32 && cfg->dtype != SIGGAN_DTYPE_BF16 && cfg->dtype != SIGGAN_DTYPE_F16)
        return fail(SIGGAN_E_INVALID, "dtype must be SIGGAN_DTYPE_F32, _BF16 or _F16, got %d", cfg->dtype);
    if (!(cfg->f16_grad_scale >= 0.f) || cfg->f16_grad_scale > 65536.f)
        return fail(SIGGAN_E_INVALID, "f16_grad_scale must be in [0, 65536] (0 = default)");
    DevGuard dg(cfg->device);
    HIPCHK(dg.err);
    siggan_ctx* c = new (std::nothrow) siggan_ctx();
    if (!c) return fail(SIGGAN_E_NOMEM, "out of host memory");
    c->cfg = *cfg; c->S = cfg->image_size; c->latent = cfg->latent_dim; c->Bm = cfg->max_batch;
    c->dt = cfg->dtype; c->es = dt_size(c->dt);
    c->fc_fused = cfg->max_batch <= 256 && (cfg->latent_dim & 3) == 0;
    c->variant = SIGGAN_STEP_TRAINER; c->abl_masks = false;
    c->sn = cfg->spectral_norm != 0;
    // fp16 stores activation gradients of order 1e-7..1e-3: a power-of-two scale (exact to apply and to remove) lifts
    // them clear of the fp16 subnormals; bf16 has fp32's exponent range and needs none
    c->gscale = c->dt == DT_F16 ? (cfg->f16_grad_scale > 0.f ? cfg->f16_grad_scale : 1024.f) : 1.0f;
    c->bound = false; c->g_dirty = c->d_dirty = true; c->pending = 0; c->metrics_last = nullptr;
    build_layout(c);

    // ---- workspace carve (two passes: size, then assign) ----------------------------------
    const int64_t Bm = c->Bm, Bd = 2 * Bm;
    size_t off = 0;
    char* base = nullptr;
    auto carve = [&](float** p, int64_t nfloats) {
        if (base) *p = (float*)(base + off);
        off += ((size_t)nfloats * sizeof(float) + 255) & ~(size_t)255;
    };
    auto carve_t = [&](char** p, int64_t nelem) {          // a tensor of the context's element type
        if (base) *p = base + off;
        off += ((size_t)nelem * c->es + 255) & ~(size_t)255;
    };
    c->slab_floats = (int64_t)16 << 20;
    c->slab_k_floats = (int64_t)16 << 20;
    for (int pass = 0; pass < 2; ++pass) {
        off = 0;
        carve(&c->z, Bm * c->latent);
        carve_t(&c->fc_y, Bm * c->F);
        for (int l = 0; l <= c->Lg; ++l) {
            const int64_t H = 4 << l, n = Bm * H * H * c->gC[l];
            if (l == 0) c->g_y[0] = c->fc_y; else carve_t(&c->g_y[l], n);
            carve_t(&c->g_a[l], n);
            if (c->dt != DT_F32) carve_t(&c->g_ae[l], n); else if (pass == 1) c->g_ae[l] = c->g_a[l];
            carve_t(&c->g_da[l], n);
            carve(&c->g_bn[l], 6 * (int64_t)(l == 0 ? c->F : c->gC[l]));
            carve(&c->g_bne[l], 4 * (int64_t)(l == 0 ? c->F : c->gC[l]));   // eval-mode [scale|shift|mean|rstd]
        }
        if (pass == 1) c->g_y[0] = c->fc_y;
        carve(&c->img, Bm * c->S * c->S);
        carve(&c->dpre, Bm * c->S * c->S);
        for (int l = 1; l <= c->Ld; ++l) {
            const int64_t H = c->S >> l, n = Bd * H * H * c->dC[l];
            carve_t(&c->d_a[l], n);
            carve_t(&c->d_dv[l], n);
            carve(&c->d_noise[l], Bd * c->dC[l]);
        }
        carve(&c->logits, Bd); carve(&c->probs, Bd); carve(&c->dlogit, Bd);
        carve(&c->lparts, (int64_t)Bd * 16);
        for (int l = 1; l <= c->Lg; ++l) {
            carve_t(&c->g_up[l], (int64_t)c->gC[l - 1] * c->gC[l] * 16);
            carve_t(&c->g_dn[l], (int64_t)c->gC[l - 1] * c->gC[l] * 16);
        }
        for (int l = 2; l <= c->Ld; ++l) {
            carve_t(&c->d_dn[l], (int64_t)c->dC[l - 1] * c->dC[l] * 16);
            carve_t(&c->d_up[l], (int64_t)c->dC[l - 1] * c->dC[l] * 16);
        }
        carve(&c->wcp, (int64_t)c->dC[c->Ld] * 16);
        carve(&c->slab, c->slab_floats);
        carve(&c->slab_k, c->slab_k_floats);
        if (c->dt != DT_F32) carve(&c->slab_b, c->slab_floats); else if (pass == 1) c->slab_b = c->slab;
        carve(&c->slab_k2, c->slab_k_floats);
        if (c->dt != DT_F32) carve(&c->slab_k3, c->slab_k_floats); else if (pass == 1) c->slab_k3 = c->slab_k2;
        carve(&c->partial, PARTIAL_FLOATS);
        carve(&c->partial_b, PARTIAL_FLOATS);
        carve(&c->partial_c, PARTIAL_FLOATS);
        carve(&c->z_g, Bm * c->latent);
        carve(&c->img_g, Bm * c->S * c->S);
        carve(&c->real_stage, Bm * c->S * c->S);
        { int64_t sumC = 0; for (int l = 1; l <= c->Ld; ++l) sumC += c->dC[l]; carve(&c->mask_stage, 3 * Bm * sumC); }
        if (c->sn) {
            int64_t ut = 1, vt = (int64_t)c->dC[c->Ld] * 16;
            for (int l = 1; l <= c->Ld; ++l) { ut += c->dC[l]; vt += (int64_t)c->dC[l - 1] * 16; }
            carve(&c->sn_tbuf, vt); carve(&c->sn_wbuf, ut);
            carve(&c->sn_sig, 3 * 2 * SnTable::MAXS);
            carve(&c->sn_us, 3 * ut); carve(&c->sn_vs, 3 * vt);
            carve(&c->sn_dots, 2 * SnTable::MAXS * 64);
            carve(&c->sn_g[0], c->d_total); carve(&c->sn_g[1], c->d_total);
            carve(&c->d_w1s, (int64_t)c->dC[1] * 16);
        }
        carve(&c->metrics, SIGGAN_M_COUNT);
        carve_t(&c->op_pack, (int64_t)512 * 512 * 16);
        carve(&c->zeros, 64);
        carve(&c->wfc_t, (int64_t)c->F * c->latent);
        carve(&c->wfin_t, (int64_t)9 * c->gC[c->Lg]);
        carve(&c->d_w1t, (int64_t)16 * c->dC[1]);
        carve(&c->ride_ctr, 64);
        float* devp = nullptr;
        carve(&devp, 64);
        if (pass == 1) c->dev = (DevState*)devp;
        if (pass == 0) {
            c->ws_bytes = off;
            hipError_t e = hipMalloc((void**)&base, off);
            if (e != hipSuccess) { delete c; return fail(SIGGAN_E_NOMEM, "hipMalloc(%zu) -> %s", off, hipGetErrorString(e)); }
            c->ws = base;
        }
    }
    HIPCHK(hipMemset(c->ws, 0, c->ws_bytes));
    DevState h; memset(&h, 0, sizeof h);
    h.seed = cfg->seed; h.rng_ctr = 0; h.grad_mul = 1.f;
    HIPCHK(hipMemcpy(c->dev, &h, sizeof h, hipMemcpyHostToDevice));
    c->mode = SIGGAN_MODE_OVERLAP;   // hipGraph replay measured slower than eager launches on ROCm 7
    c->evi = 0;
    HIPCHK(hipStreamCreateWithFlags(&c->s_m, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->s_a, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->s_b, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->s_c, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&c->s_n, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&c->ev_gfwd, EV_FLAGS));
    HIPCHK(hipEventCreateWithFlags(&c->ev_dreal, EV_FLAGS));
    HIPCHK(hipEventCreateWithFlags(&c->ev_ar, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) HIPCHK(hipEventCreateWithFlags(&c->ev_sys[i], hipEventDisableTiming));
    c->early_ar = false;
    c->staged_B = c->dreal_B = 0; c->staged_src = nullptr; c->dreal_joined = c->dreal_noise2 = c->gfwd_joined = false;
    c->dreal_orphan = false; c->lane_err = hipSuccess;
    c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1; c->comm_err = 0;
    c->g_fwd_pending = 0; c->conv1_rode = 0;
    c->zg_stash = 0;
    c->ga_last_B = 0; c->g_r0 = 0;
    c->adam_t_known[0] = c->adam_t_known[1] = false;
    for (int i = 0; i < siggan_ctx::NEV; ++i) HIPCHK(hipEventCreateWithFlags(&c->ev[i], EV_FLAGS));
    for (int i = 0; i < siggan_ctx::NEV; ++i) HIPCHK(hipEventCreateWithFlags(&c->ev_fenced[i], hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) HIPCHK(hipEventCreateWithFlags(&c->ev_bridge[i], hipEventDisableTiming));
    *out = c;
    return SIGGAN_OK;
}

extern "C" int siggan_destroy(siggan_ctx* c) {
    if (!c) return SIGGAN_OK;
    DevGuard dg(c->cfg.device);
    (void)hipDeviceSynchronize();
    if (c->comm) { (void)rccl()->CommDestroy(c->comm); c->comm = nullptr; }
    for (auto& e : c->graphs) (void)hipGraphExecDestroy(e.second);
    for (int i = 0; i < siggan_ctx::NEV; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
    for (int i = 0; i < siggan_ctx::NEV; ++i) if (c->ev_fenced[i]) (void)hipEventDestroy(c->ev_fenced[i]);
    for (int i = 0; i < 2; ++i) if (c->ev_bridge[i]) (void)hipEventDestroy(c->ev_bridge[i]);
    if (c->s_m) (void)hipStreamDestroy(c->s_m);
    if (c->s_a) (void)hipStreamDestroy(c->s_a);
    if (c->s_b) (void)hipStreamDestroy(c->s_b);
    if (c->s_c) (void)hipStreamDestroy(c->s_c);
    if (c->s_n) (void)hipStreamDestroy(c->s_n);
    if (c->ev_ar) (void)hipEventDestroy(c->ev_ar);
    for (int i = 0; i < 2; ++i) if (c->ev_sys[i]) (void)hipEventDestroy(c->ev_sys[i]);
    if (c->ev_gfwd) (void)hipEventDestroy(c->ev_gfwd);
    if (c->ev_dreal) (void)hipEventDestroy(c->ev_dreal);
    if (c->ws) (void)hipFree(c->ws);
    delete c;
    return SIGGAN_OK;
}

extern "C" int64_t siggan_param_count(const siggan_ctx* c, int which) { return !c ? -1 : (which == 0 ? c->g_total : c->d_total); }
extern "C" int32_t siggan_param_tensors(const siggan_ctx* c, int which) {
    return !c ? -1 : (int32_t)(which == 0 ? c->g_off.size() : c->d_off.size());
}
extern "C" int siggan_param_span(const siggan_ctx* c, int which, int32_t idx, int64_t* offset, int64_t* numel) {
    if (!c || !offset || !numel) return fail(SIGGAN_E_INVALID, "null argument");
    const auto& o = which == 0 ? c->g_off : c->d_off;
    const auto& n = which == 0 ? c->g_num : c->d_num;
    if (idx < 0 || idx >= (int32_t)o.size()) return fail(SIGGAN_E_INVALID, "tensor index %d out of range", idx);
    *offset = o[idx]; *numel = n[idx];
    return SIGGAN_OK;
}
extern "C" int64_t siggan_bn_count(const siggan_ctx* c) { return c ? c->bn_total : -1; }
extern "C" int64_t siggan_sn_count(const siggan_ctx* c, int which) {
    if (!c) return -1;
    int64_t ut = 1, vt = (int64_t)c->dC[c->Ld] * 16;
    for (int l = 1; l <= c->Ld; ++l) { ut += c->dC[l]; vt += (int64_t)c->dC[l - 1] * 16; }
    return which == 0 ? ut : vt;
}
extern "C" int32_t siggan_bn_layers(const siggan_ctx* c) { return c ? c->Lg + 1 : -1; }
extern "C" int64_t siggan_workspace_bytes(const siggan_ctx* c) { return c ? (int64_t)c->ws_bytes : -1; }

// A D(real) forward that siggan_g_grads started ahead of time on lane c (siggan_stage_real) and that no D step will
// consume still reads the staged batch, the D weight packs and slab_k2 and writes activation rows [0,B): whoever
// abandons it marks it orphaned, and the next call that enqueues anything first makes its stream wait for that lane.
static void drop_dreal(siggan_ctx* c) {
    if (c->dreal_B) { c->dreal_orphan = true; c->dreal_B = 0; }
}
static int settle(siggan_ctx* c, hipStream_t s) {
    if (c->dreal_orphan) {
        HIPCHK(hipStreamWaitEvent(s, c->ev_dreal, 0));
        c->dreal_orphan = false;
    }
    return SIGGAN_OK;
}

extern "C" int siggan_bind(siggan_ctx* c, const siggan_storage* st) {
    if (!c || !st) return fail(SIGGAN_E_INVALID, "null argument");
    const void* need[] = {st->g_params, st->g_bn_running_mean, st->g_bn_running_var, st->d_params};
    for (const void* p : need)
        if (!p) return fail(SIGGAN_E_INVALID, "siggan_bind: parameters and BatchNorm buffers are required");
    const void* all[] = {st->g_params, st->g_grads, st->g_exp_avg, st->g_exp_avg_sq, st->d_params, st->d_grads,
                         st->d_exp_avg, st->d_exp_avg_sq};
    for (const void* p : all)
        if (((uintptr_t)p & 15) != 0) return fail(SIGGAN_E_INVALID, "siggan_bind: arenas must be 16-byte aligned");
    c->st = *st;
    if (c->sn) {
        if (!st->d_sn_u || !st->d_sn_v) return fail(SIGGAN_E_INVALID, "siggan_bind: a spectral-norm context needs d_sn_u / d_sn_v");
        SnTable& t = c->snt; memset(&t, 0, sizeof t);
        t.n = c->Ld + 1;
        int uo = 0, vo = 0;
        t.pre_k[0] = t.pre_r[0] = 0;
        for (int i = 0; i < t.n; ++i) {
            SnLayer& L = t.layer[i];
            const bool cls = i == c->Ld;
            const int pi = cls ? di_cls_w(c) : di_w(i + 1);
            L.W = st->d_params + c->d_off[pi]; L.w_off = c->d_off[pi];
            L.rows = cls ? 1 : c->dC[i + 1]; L.K = cls ? c->dC[c->Ld] * 16 : c->dC[i] * 16;
            L.u_off = uo; L.v_off = vo; uo += L.rows; vo += L.K;
            t.pre_k[i + 1] = t.pre_k[i] + (L.K + 255) / 256;
            t.pre_r[i + 1] = t.pre_r[i] + (L.rows + 3) / 4;
        }
        t.u = st->d_sn_u; t.v = st->d_sn_v; t.tbuf = c->sn_tbuf; t.wbuf = c->sn_wbuf; t.sig = c->sn_sig;
        t.u_saved = c->sn_us; t.v_saved = c->sn_vs; t.dots = c->sn_dots; t.u_total = uo; t.v_total = vo;
    }
    c->bound = true; c->g_dirty = c->d_dirty = true; c->pending = 0; c->staged_B = 0; c->conv1_rode = 0;
    c->adam_t_known[0] = c->adam_t_known[1] = false;
    c->early_ar = false;               // (an early all-reduce whose apply never ran is abandoned with the gradients it covered)
    drop_dreal(c);
    return SIGGAN_OK;
}
extern "C" int siggan_params_changed(siggan_ctx* c) {
    if (!c) return fail(SIGGAN_E_INVALID, "null context");
    c->g_dirty = c->d_dirty = true;
    c->conv1_rode = 0;                 // (a first-block forward the last D apply ran ahead used the old weights too)
    c->adam_t_known[0] = c->adam_t_known[1] = false;      // (optimizer.load_state_dict writes the step tensors)
    c->early_ar = false;
    drop_dreal(c);                     // a D(real) forward started ahead of time used the old weights
    return SIGGAN_OK;
}
extern "C" int siggan_seed(siggan_ctx* c, uint64_t seed, uint64_t offset) {
    if (!c) return fail(SIGGAN_E_INVALID, "null context");
    DevGuard dg_(c->cfg.device); HIPCHK(dg_.err);
    unsigned long long v[2] = {seed, offset};
    HIPCHK(hipMemcpy(c->dev, v, sizeof v, hipMemcpyHostToDevice));   // seed, rng_ctr are the first two fields
    drop_dreal(c);                     // its dropout tables were drawn from the old stream
    return SIGGAN_OK;
}

extern "C" int siggan_rng_state(siggan_ctx* c, uint64_t* seed, uint64_t* offset) {
    ENTER(c);
    if (!seed || !offset) return fail(SIGGAN_E_INVALID, "null argument");
    HIPCHK(hipDeviceSynchronize());
    unsigned long long v[2];
    HIPCHK(hipMemcpy(v, c->dev, sizeof v, hipMemcpyDeviceToHost));
    *seed = v[0]; *offset = v[1];
    return SIGGAN_OK;
}

// ------------------------------------------------------------------------------------------
// internal passes
// ------------------------------------------------------------------------------------------
// A training-mode Generator forward of ONE sample: the fc block's BatchNorm1d has a single value per channel, which torch
// refuses (torch/nn/functional.py _verify_batch_size, reached from generator_vanilla_gan.py:112) -- same refusal, same text
// (the binding raises ValueError for SIGGAN_E_INVALID, as torch does).  Eval mode and the Discriminator take one sample.
static int check_bn_batch(const siggan_ctx* c, int batch) {
    if (batch == 1)
        return fail(SIGGAN_E_INVALID, "Expected more than 1 value per channel when training, got input size torch.Size([1, %d])", c->F);
    return SIGGAN_OK;
}
static int check_call(siggan_ctx* c, int batch, bool need_bound = true) {
    if (!c) return fail(SIGGAN_E_INVALID, "null context");
    if (need_bound && !c->bound) return fail(SIGGAN_E_STATE, "siggan_bind has not been called");
    if (batch < 1 || batch > c->Bm) return fail(SIGGAN_E_INVALID, "batch %d outside [1, max_batch=%d]", batch, c->Bm);
    return SIGGAN_OK;
}

// One launch per network rebuilds everything derived from its arena: GEMM-friendly weight copies and
// (for G) the BatchNorm eval-mode scale/shift tables.  sg / sd: the lanes the two launches go to.
// conv1_x != nullptr (G step, no spectral norm): the D table's launch also runs the first-block forward of conv1_B images at
// conv1_x into workspace rows [conv1_r0, ...) -- it reads the raw block-1 weights, not a pack (launch_prepare_conv1)
static void repack(siggan_ctx* c, hipStream_t sg, hipStream_t sd, bool do_g, bool do_d, int sn_slot = 0, const float* conv1_x = nullptr,
                   int conv1_r0 = 0, int conv1_B = 0) {
    if (do_g) {
        PrepTable t; t.njobs = 0; t.overflow = 0;
        PrepJob j; memset(&j, 0, sizeof j);
        if (!c->fc_fused) {                                            // the generic fc kernel reads a k-major copy
            j.type = PREP_FC_T; j.O = c->latent; j.I = c->gC[0]; j.src = GP(c, gi_fc_w()); j.dst = c->wfc_t;
            prep_add(t, j, (long long)c->latent * c->F);
        }
        for (int l = 1; l <= c->Lg; ++l) {
            const long long n = (long long)c->gC[l - 1] * c->gC[l] * 16;
            memset(&j, 0, sizeof j);
            j.src = GP(c, gi_up_w(l)); j.dt = c->dt;                   // (Cin, Cout, 4, 4)
            j.type = PREP_PACK_UP; j.I = c->gC[l - 1]; j.O = c->gC[l]; j.dst = (float*)c->g_up[l];   // forward: contract Cin
            prep_add(t, j, n);
            j.type = PREP_PACK_DOWN; j.O = c->gC[l - 1]; j.I = c->gC[l]; j.dst = (float*)c->g_dn[l]; // input-gradient: out = Cin
            prep_add(t, j, n);
        }
        for (int l = 0; l <= c->Lg; ++l) {
            const int C = l == 0 ? c->F : c->gC[l];
            memset(&j, 0, sizeof j);
            j.type = PREP_BN_EVAL; j.O = C; j.perm = l == 0 ? c->gC[0] : 0;
            j.src = GP(c, gi_bn_w(l)); j.src2 = GP(c, gi_bn_b(l));
            j.src3 = c->st.g_bn_running_mean + c->g_bn_off[l]; j.src4 = c->st.g_bn_running_var + c->g_bn_off[l];
            j.dst = c->g_bne[l];
            prep_add(t, j, C);
        }
        memset(&j, 0, sizeof j);                                       // final conv: [tap][c] for the strip kernels
        j.type = PREP_TAPS; j.I = 9; j.O = c->gC[c->Lg]; j.src = GP(c, gi_fin_w(c)); j.dst = c->wfin_t;
        prep_add(t, j, 9 * j.O);
        if (!launch_prepare(t, BN_EPS, sg)) c->lane_err = hipErrorInvalidValue;   // table overflow: reported by the caller
    }
    if (do_d) {
        PrepTable t; t.njobs = 0; t.overflow = 0;
        PrepJob j;
        for (int l = 2; l <= c->Ld; ++l) {
            const long long n = (long long)c->dC[l - 1] * c->dC[l] * 16;
            memset(&j, 0, sizeof j);
            j.src = DP(c, di_w(l)); j.dt = c->dt;                      // (Cout, Cin, 4, 4)
            if (c->sn) j.mul = c->sn_sig + (sn_slot * 2 + 1) * SnTable::MAXS + (l - 1);   // W / sigma of this pass
            j.type = PREP_PACK_DOWN; j.O = c->dC[l]; j.I = c->dC[l - 1]; j.dst = (float*)c->d_dn[l];  // forward
            prep_add(t, j, n);
            j.type = PREP_PACK_UP; j.I = c->dC[l]; j.O = c->dC[l - 1]; j.dst = (float*)c->d_up[l];    // input-gradient: contract Cout
            prep_add(t, j, n);
        }
        memset(&j, 0, sizeof j);
        j.type = PREP_CLS; j.O = c->dC[c->Ld]; j.src = DP(c, di_cls_w(c)); j.dst = c->wcp;
        if (c->sn) j.mul = c->sn_sig + (sn_slot * 2 + 1) * SnTable::MAXS + c->Ld;
        prep_add(t, j, (long long)c->dC[c->Ld] * 16);
        if (c->sn) {                                                   // block 1 is not an MFMA kernel: an fp32 scaled copy
            memset(&j, 0, sizeof j);
            j.type = PREP_SCALE; j.O = c->dC[1] * 16; j.src = DP(c, di_w(1)); j.dst = c->d_w1s;
            j.mul = c->sn_sig + (sn_slot * 2 + 1) * SnTable::MAXS;
            prep_add(t, j, j.O);
        }
        memset(&j, 0, sizeof j);                                       // block 1 as [tap][co] for its input-gradient kernel
        j.type = PREP_TAPS; j.I = 16; j.O = c->dC[1]; j.src = DP(c, di_w(1)); j.dst = c->d_w1t;
        if (c->sn) j.mul = c->sn_sig + (sn_slot * 2 + 1) * SnTable::MAXS;
        prep_add(t, j, j.O * 16);
        bool ok;
        if (conv1_x && !c->sn) {
            const int64_t H = c->S >> 1;
            char* out = c->d_a[1] + (size_t)((int64_t)conv1_r0 * H * H * c->dC[1]) * c->es;
            ok = launch_prepare_conv1(t, BN_EPS, c->dt, conv1_x, DP(c, di_w(1)), DP(c, di_b(1)), c->cfg.leaky_slope, out, conv1_B, c->S, sd);
        } else {
            ok = launch_prepare(t, BN_EPS, sd);
        }
        if (!ok) c->lane_err = hipErrorInvalidValue;
    }
}

// The job tables of the one-launch update (launch_adam_pack): the whole arena, tensor by tensor, with what launch_prepare
// would derive from each.  false: this context keeps the two-launch path (the generic fc kernel's k-major copy, channel
// counts the 16 x 16 tiles do not divide).
static bool ap_table_g(siggan_ctx* c, ApTable& t) {
    if (!c->fc_fused || c->gC[c->Lg] * 9 > 1024) return false;
    for (int l = 0; l <= c->Lg; ++l) if (c->gC[l] % 16) return false;
    t.njobs = 0; t.overflow = 0;
    ApJob j;
    memset(&j, 0, sizeof j);                                           // fc weight + bias: adjacent in the arena, nothing derived
    j.type = AP_FLAT; j.off = c->g_off[gi_fc_w()]; j.n = c->g_num[gi_fc_w()] + c->g_num[gi_fc_b()];
    ap_add(t, j);
    for (int l = 0; l <= c->Lg; ++l) {
        if (l > 0) {
            memset(&j, 0, sizeof j);                                   // (Cin, Cout, 4, 4): forward contracts Cin, input-gradient Cout
            j.type = AP_CONV; j.A = c->gC[l - 1]; j.Bc = c->gC[l]; j.dt = c->dt; j.off = c->g_off[gi_up_w(l)];
            j.dst = (float*)c->g_dn[l]; j.dst2 = (float*)c->g_up[l];
            ap_add(t, j);
        }
        memset(&j, 0, sizeof j);
        j.type = AP_BN; j.A = l == 0 ? c->F : c->gC[l]; j.Bc = l == 0 ? c->gC[0] : 0;
        j.off = c->g_off[gi_bn_w(l)]; j.off2 = c->g_off[gi_bn_b(l)];
        j.rmean = c->st.g_bn_running_mean + c->g_bn_off[l]; j.rvar = c->st.g_bn_running_var + c->g_bn_off[l];
        j.dst = c->g_bne[l];
        ap_add(t, j);
    }
    memset(&j, 0, sizeof j);                                           // final conv: [tap][c] for the strip kernels, and its bias
    j.type = AP_TAPS; j.A = c->gC[c->Lg]; j.Bc = 9; j.off = c->g_off[gi_fin_w(c)]; j.n = 9 * c->gC[c->Lg];
    j.off2 = c->g_off[gi_fin_b(c)]; j.n2 = 1; j.dst = c->wfin_t;
    ap_add(t, j);
    return !t.overflow;
}
static bool ap_table_d(siggan_ctx* c, ApTable& t, int nride) {
    if (c->sn || c->dC[1] * 16 > 1024 || c->dC[1] > 256) return false;     // (spectral norm: the packs follow sigma, not the update)
    for (int l = 1; l <= c->Ld; ++l) if (c->dC[l] % 16) return false;
    t.njobs = 0; t.overflow = 0;
    ApJob j;
    memset(&j, 0, sizeof j);                                           // block 1 (weight + bias: what the riders read) first
    j.type = AP_TAPS; j.A = c->dC[1]; j.Bc = 16; j.off = c->d_off[di_w(1)]; j.n = 16 * c->dC[1];
    j.off2 = c->d_off[di_b(1)]; j.n2 = c->dC[1]; j.dst = c->d_w1t; j.wait = nride;
    ap_add(t, j);
    for (int l = 2; l <= c->Ld; ++l) {
        memset(&j, 0, sizeof j);                                       // (Cout, Cin, 4, 4): forward contracts Cin, input-gradient Cout
        j.type = AP_CONV; j.A = c->dC[l]; j.Bc = c->dC[l - 1]; j.dt = c->dt; j.off = c->d_off[di_w(l)];
        j.dst = (float*)c->d_dn[l]; j.dst2 = (float*)c->d_up[l];
        ap_add(t, j);
        memset(&j, 0, sizeof j);
        j.type = AP_FLAT; j.off = c->d_off[di_b(l)]; j.n = c->dC[l];
        ap_add(t, j);
    }
    memset(&j, 0, sizeof j);
    j.type = AP_T16; j.A = c->dC[c->Ld]; j.off = c->d_off[di_cls_w(c)]; j.dst = c->wcp;
    ap_add(t, j);
    memset(&j, 0, sizeof j);
    j.type = AP_FLAT; j.off = c->d_off[di_cls_b(c)]; j.n = 1;
    ap_add(t, j);
    return !t.overflow;
}

// ------------------------------------------------------------------------------------------
// lanes: where a phase enqueues its kernels.  m is the main lane; a and b are side lanes that run
// independent work (weight gradients, bias / BatchNorm reductions) beside the main chain.  With
// overlap off all three are the same stream and fork/join are no-ops.  Forks and joins are plain
// event record / wait pairs, so the same code runs eagerly and under stream capture (hipGraph).
// ------------------------------------------------------------------------------------------
static int lane_check(siggan_ctx* c) {
    if (c->comm_err)      // sticky: the communicator is unusable and the replicas have diverged; cleared by siggan_comm_destroy
        return fail(SIGGAN_E_HIP, "ncclAllReduce of the gradient bucket failed (the optimiser update was skipped): %s",
                    rccl()->GetErrorString(c->comm_err));
    if (c->lane_err == hipSuccess) return SIGGAN_OK;
    const hipError_t e = c->lane_err; c->lane_err = hipSuccess;
    return fail(SIGGAN_E_HIP, "an event record / stream wait / prepare table of the step failed: %s", hipGetErrorString(e));
}
struct Lanes {
    siggan_ctx* c;
    hipStream_t m, a, b;
    // ext: a fork may ride on the producing kernel's own completion signal (ops.h, SIGGAN_LAUNCH_EV) instead of a marker packet
    // behind it -- eager launches only (not under stream capture, not with the profiler's own events on the launch)
    bool ext;
    hipEvent_t tail_wait;        // d_backward_pass: one more event the main lane waits for where it is idle anyway (before its last join)
    // the event to hand the producing launch as `done` (nullptr: fork_after records one the classic way)
    hipEvent_t fork_event(hipStream_t to) { return (ext && to != m) ? next() : nullptr; }
    void fork_after(hipStream_t to, hipEvent_t done) {      // `to` waits for the launch that was given `done` (and all before it on m)
        if (to == m) return;
        if (done) note(hipStreamWaitEvent(to, done, 0)); else fork(to);
    }
    // with a communicator every lane event keeps its system-scope fence: peers read this device's gradient arena over xGMI
    hipEvent_t next() { hipEvent_t e = (c->comm ? c->ev_fenced : c->ev)[c->evi]; c->evi = (c->evi + 1) % siggan_ctx::NEV; return e; }
    void fork(hipStream_t to) {          // `to` waits for everything enqueued on m so far
        if (to == m) return;
        hipEvent_t e = next();
        note(hipEventRecord(e, m)); note(hipStreamWaitEvent(to, e, 0));
    }
    void join(hipStream_t from) {        // m waits for everything enqueued on `from`
        if (from == m) return;
        hipEvent_t e = next();
        note(hipEventRecord(e, from)); note(hipStreamWaitEvent(m, e, 0));
    }
    // a failed record / wait would silently drop an ordering edge: remember the first one, run_phase reports it
    void note(hipError_t e) { if (e != hipSuccess && c->lane_err == hipSuccess) c->lane_err = e; }
    void record(hipEvent_t e, hipStream_t s) { note(hipEventRecord(e, s)); }
    void wait(hipStream_t s, hipEvent_t e) { note(hipStreamWaitEvent(s, e, 0)); }
};

static GConvArgs gconv_args(siggan_ctx* c) {
    GConvArgs a; memset(&a, 0, sizeof a);
    a.dt = c->dt;
    a.slab = c->slab_k; a.slab_floats = c->slab_k_floats; a.zeros = c->zeros;
    return a;
}

// Generator.forward (generator_vanilla_gan.py:189-209).  training: BN batch stats (+ running
// update) and raw pre-BN outputs kept for the backward pass; eval: BN folded into the epilogue.
// z == nullptr: the latent batch is drawn inside the fc kernel from RNG stream rng_sid and left in z_out.
static void g_forward_pass(siggan_ctx* c, const float* z, int B, bool training, float* img, hipStream_t s,
                           float* partial = nullptr, float* slab_k = nullptr, uint32_t rng_sid = 0, float* z_out = nullptr,
                           hipEvent_t done = nullptr) {     // done: completion event of the pass' last launch
    if (!partial) partial = c->partial;
    char* const* const A = training ? c->g_a : c->g_ae;     // (fp32: the same buffers)
    // fc + BatchNorm1d + ReLU: one MFMA launch (fc.hip) whenever the shape allows, else the generic kernels
    if (c->fc_fused && launch_fc_fwd_fused(c->dt, z, GP(c, gi_fc_w()), GP(c, gi_fc_b()), c->fc_y, A[0], GP(c, gi_bn0_w()),
                                          GP(c, gi_bn0_b()), c->st.g_bn_running_mean, c->st.g_bn_running_var, c->st.g_bn_batches,
                                          c->g_bn[0], training ? nullptr : c->g_bne[0], z_out, c->dev, rng_sid, B, c->latent,
                                          c->gC[0], BN_MOMENTUM, BN_EPS, s)) {
    } else if (!training) {     // eval: BatchNorm1d + ReLU folded into the fc epilogue
        launch_fc_fwd(c->dt, z, c->wfc_t, GP(c, gi_fc_b()), A[0], B, c->latent, c->gC[0], s, c->g_bne[0], c->dev, rng_sid, z_out);
    } else {
        launch_fc_fwd(c->dt, z, c->wfc_t, GP(c, gi_fc_b()), c->fc_y, B, c->latent, c->gC[0], s, nullptr, c->dev, rng_sid, z_out);
        launch_bn_train_stats(c->dt, c->fc_y, B, c->F, GP(c, gi_bn0_w()), GP(c, gi_bn0_b()), c->st.g_bn_running_mean,
                              c->st.g_bn_running_var, c->st.g_bn_batches, c->g_bn[0], partial, c->gC[0], BN_MOMENTUM,
                              BN_EPS, s);
        launch_bn_relu(c->dt, c->fc_y, A[0], B, c->F, c->g_bn[0], s);
    }
    for (int l = 1; l <= c->Lg; ++l) {
        const int Hi = 4 << (l - 1), Ci = c->gC[l - 1], Co = c->gC[l];
        GConvArgs a = gconv_args(c);
        if (slab_k) a.slab = slab_k;
        a.in = A[l - 1]; a.wp = c->g_up[l]; a.B = B; a.Hi = Hi; a.Wi = Hi; a.Ci = Ci; a.Co = Co;
        a.lgHr = ilog2i(Hi); a.lgWr = a.lgHr; a.Ho = 2 * Hi; a.Wo = 2 * Hi; a.form = 1; a.M = B * Hi * Hi;
        const int C = Co;
        const int64_t off = c->g_bn_off[l];
        if (training) {
            a.out = c->g_y[l]; a.epi = EPI_RAW;
            launch_gconv(a, s);
            const int64_t R = (int64_t)B * 4 * Hi * Hi;
            // the LAST block's activation has two readers, the final conv here and its weight gradient in the backward
            // pass: both re-derive it from y and this table, so it is never written (33.5 MB each way at batch 64)
            launch_bn_train_stats(c->dt, c->g_y[l], R, C, GP(c, gi_bn_w(l)), GP(c, gi_bn_b(l)), c->st.g_bn_running_mean + off,
                                  c->st.g_bn_running_var + off, c->st.g_bn_batches + l, c->g_bn[l], partial, 0,
                                  BN_MOMENTUM, BN_EPS, s);
            if (l < c->Lg) launch_bn_relu(c->dt, c->g_y[l], A[l], R, C, c->g_bn[l], s);
        } else {
            a.out = A[l]; a.epi = EPI_AFFINE_RELU; a.scale = c->g_bne[l]; a.shift = c->g_bne[l] + C;
            launch_gconv(a, s);
        }
    }
    if (training)
        launch_final_fwd(c->dt, c->g_y[c->Lg], c->wfin_t, GP(c, gi_fin_b(c)), img, B, c->S, c->gC[c->Lg], s, c->g_bn[c->Lg], done);
    else
        launch_final_fwd(c->dt, A[c->Lg], c->wfin_t, GP(c, gi_fin_b(c)), img, B, c->S, c->gC[c->Lg], s, nullptr, done);
    c->ga_last_B = training ? B : 0;
}

// Discriminator conv blocks + classifier logits for nB images written to workspace rows
// [r0, r0 + nB) (the D step runs D(real) into rows [0,B) on a side lane while the Generator
// produces the fakes, then D(fake) into rows [B,2B); backward treats the 2B rows as one batch).
static void d_forward_rows(siggan_ctx* c, const float* x, int r0, int nB, bool dropout, hipStream_t s, float* slab_k,
                           bool fuse_cls = false, bool conv1_done = false) {
    const float slope = c->cfg.leaky_slope;
    auto act = [&](int l) { const int64_t H = c->S >> l; return c->d_a[l] + (size_t)((int64_t)r0 * H * H * c->dC[l]) * c->es; };
    auto nz = [&](int l) { return dropout ? c->d_noise[l] + (int64_t)r0 * c->dC[l] : nullptr; };
    if (!conv1_done)          // (done: it rode in the launch that re-packed D's weights, see repack)
        launch_conv1_fwd(c->dt, x, nB, x, c->sn ? c->d_w1s : DP(c, di_w(1)), DP(c, di_b(1)), nz(1), slope, act(1), nB, c->S, c->dC[1], s);
    for (int l = 2; l <= c->Ld; ++l) {
        const int Hi = c->S >> (l - 1), Ho = Hi / 2;
        GConvArgs a = gconv_args(c);
        a.slab = slab_k;
        a.in = act(l - 1); a.wp = c->d_dn[l]; a.out = act(l);
        a.B = nB; a.Hi = Hi; a.Wi = Hi; a.Ci = c->dC[l - 1]; a.Co = c->dC[l];
        a.lgHr = ilog2i(Ho); a.lgWr = a.lgHr; a.Ho = Ho; a.Wo = Ho; a.form = 0; a.M = nB * Ho * Ho;
        a.epi = EPI_BIAS_LRELU_DROP; a.bias = DP(c, di_b(l)); a.noise = nz(l); a.slope = slope;
        // fuse_cls (the step phases, whose k_bce / k_cls_bwd read either form): when the last block ends in a split-K epilogue,
        // the classifier's dot product rides there as P partials per image and k_cls_fwd is not launched
        const int P_max = c->dC[c->Ld] * 16 / 1024;
        if (l == c->Ld && fuse_cls && P_max >= 1 && P_max <= 16) { a.cls_w = c->wcp; a.cls_part = c->lparts + (size_t)r0 * P_max; }
        const int P = launch_gconv(a, s);
        if (l == c->Ld) c->lP[r0 == 0 ? 0 : 1] = a.cls_w ? P : 0;
    }
    if (c->lP[r0 == 0 ? 0 : 1] == 0)
        launch_cls_fwd(c->dt, act(c->Ld), c->wcp, DP(c, di_cls_b(c)), c->logits + r0, nB, c->dC[c->Ld] * 16, s);
}

// Backward through the Discriminator from d(logit).  want_wgrad: fill the D gradient arena
// (D step); want_dimage: continue to d(pre-tanh image) (G step).  The chain of input-gradients
// runs on lane m; each block's weight gradient (lane a) and bias gradient (lane b) only need that
// block's d(pre-activation) and run beside the rest of the chain.
struct BceSpec { int n0; float y0, y1; float* mt; int is_g; };   // rows < n0: target y0, the rest y1 (each segment's mean)

// r0 / garena (spectral norm: one pass at a time): the Bd rows start at workspace row r0 and the gradients go to garena
// (an arena-shaped temporary) instead of the bound arena.
static void d_backward_pass(siggan_ctx* c, Lanes& L, const float* x0, int n0, const float* x1, int Bd, bool dropout,
                            bool want_wgrad, bool want_dimage, const BceSpec& bce, int r0 = 0, float* garena = nullptr,
                            bool early_allreduce = false) {
    const float slope = c->cfg.leaky_slope;
    const int Ld = c->Ld;
    float* const ga = garena ? garena : c->st.d_grads;
    auto G_ = [&](int i) { return ga + c->d_off[i]; };
    auto act = [&](int l) { const int64_t H = c->S >> l; return c->d_a[l] + (size_t)((int64_t)r0 * H * H * c->dC[l]) * c->es; };
    auto dvp = [&](int l) { const int64_t H = c->S >> l; return c->d_dv[l] + (size_t)((int64_t)r0 * H * H * c->dC[l]) * c->es; };
    auto nz = [&](int l) { return dropout ? c->d_noise[l] + (int64_t)r0 * c->dC[l] : nullptr; };
    // sigmoid + BCE: losses / means into the metrics, d(logit) for the classifier's weight gradient -- on lane b; the
    // chain below recomputes d(logit) from the logits and does not wait for it
    // Lanes: an event record on the main lane delays the kernel behind it by ~6 us and a join whose event
    // completes just before it is waited for costs ~15 us, so a side lane must carry enough work to pay for its fork and
    // join.  D step: the weight gradients (lane a: they overlap the input-gradient chain, putting them on the main lane costs
    // 3 %) and the small reductions (lane b).  G step: nothing here is worth a lane -- the 5 us loss kernel stays on m.
    hipStream_t const sb = want_wgrad ? L.b : L.m;
    // the logits of the rows this pass covers: stored, or P partial dot products per image (d_forward_rows).  Both halves of a
    // 2B-row pass were produced with the same batch size, hence in the same form
    const int hP = c->lP[r0 == 0 ? 0 : 1];
    const int P = (r0 == 0 && Bd > bce.n0 && bce.n0 > 0 && c->lP[1] != hP) ? -1 : hP;
    if (P < 0) { c->lane_err = hipErrorInvalidValue; return; }       // (cannot happen: reported, not computed wrongly)
    const float* const parts = P ? c->lparts + (size_t)r0 * P : nullptr;
    const float* const bc = DP(c, di_cls_b(c));
    if (sb != L.m) {
        L.fork(L.b);
        launch_bce(c->logits + r0, Bd, bce.n0, bce.y0, bce.y1, c->probs + r0, c->dlogit + r0, bce.mt, bce.is_g, sb, c->gscale, parts, P, bc);
    }
    // (G step: the loss kernel's work is one more block of the classifier's input-gradient kernel -- one launch, not two)
    launch_cls_bwd(c->dt, c->logits + r0, bce.n0, bce.y0, bce.y1, c->wcp, act(Ld), nz(Ld), slope, dvp(Ld), Bd,
                   c->dC[Ld], L.m, c->gscale, c->probs + r0, c->dlogit + r0, bce.mt, bce.is_g, sb == L.m, parts, P, bc);
    if (want_wgrad)
        launch_cls_wgrad(c->dt, c->dlogit + r0, act(Ld), G_(di_cls_w(c)), G_(di_cls_b(c)), Bd, c->dC[Ld], sb);
    for (int l = Ld; l >= 2; --l) {
        const int Ho = c->S >> l, Hi = 2 * Ho, Co = c->dC[l], Ci = c->dC[l - 1];
        // 16-bit contexts (launch-latency-bound): the LAST block's weight gradient runs behind the input-gradient chain on the
        // main lane, own slab, so that lane a ends before the main lane does (measured with the three 16-bit lane rules
        // together: bf16 0.718 -> 0.707 ms; at fp32 each is within noise and the plain structure stays)
        const bool w_main = want_wgrad && l == 2 && c->dt != DT_F32 && garena == nullptr;
        auto wgrad_l = [&](hipStream_t st, float* slab) {
            WgradArgs w; memset(&w, 0, sizeof w); w.zeros = c->zeros; w.dt = c->dt;
            w.S = dvp(l); w.L = act(l - 1); w.slab = slab; w.dw = G_(di_w(l)); w.B = Bd; w.Cs = Co; w.Cl = Ci;
            w.lgHs = ilog2i(Ho); w.lgWs = w.lgHs; w.lgCl = ilog2i(Ci); w.K = Bd * Ho * Ho;
            w.db = G_(di_b(l));                              // bias gradient = column sums of d(pre-activation): rides in the same kernel
            const int max_splits = (int)(c->slab_floats / ((int64_t)Co * (16 * Ci + 1)));
            launch_wgrad(w, max_splits, st);
        };
        if (want_wgrad && !w_main) {
            L.fork(L.a);                                   // d_dv[l] is complete on m here
            wgrad_l(L.a, c->slab);
            if (early_allreduce && l == Ld && c->comm && !c->comm_err && garena == nullptr) {
                // data parallel: the tail of the Discriminator's bucket -- the LAST block's weight (76 % of the bucket) and bias
                // gradient (this kernel, lane a) and the classifier's (k_cls_wgrad, lane b) -- is complete first: its
                // all-reduce starts now, on a lane of its own, under the rest of the backward pass; *_apply reduces the head of
                // the arena and waits for this one (same sums: an all-reduce is elementwise)
                hipEvent_t ea = c->ev_sys[0], eb = c->ev_sys[1];     // (system-scope release: the collective's peers read this arena)
                L.record(ea, L.a); L.record(eb, sb);
                L.wait(c->s_n, ea); L.wait(c->s_n, eb);
                const int64_t o = c->d_off[di_w(l)];
                const int rc = rccl()->AllReduce(ga + o, ga + o, (size_t)(c->d_total - o), NCCL_FLOAT32, NCCL_SUM, c->comm, c->s_n);
                if (rc != NCCL_SUCCESS) c->comm_err = rc;
                else { L.record(c->ev_ar, c->s_n); c->early_ar = true; }
            }
        }
        // input gradient ("up" form): contract Cout, produce Cin at (Hi x Hi); fused leaky'/dropout of block l-1
        GConvArgs a = gconv_args(c);
        a.in = dvp(l); a.wp = c->d_up[l]; a.out = dvp(l - 1);
        a.B = Bd; a.Hi = Ho; a.Wi = Ho; a.Ci = Co; a.Co = Ci;
        a.lgHr = ilog2i(Ho); a.lgWr = a.lgHr; a.Ho = Hi; a.Wo = Hi; a.form = 1; a.M = Bd * Ho * Ho;
        a.epi = EPI_LRELU_BWD; a.aref = act(l - 1); a.noise = nz(l - 1); a.slope = slope;
        launch_gconv(a, L.m);
        if (w_main) {
            L.fork(L.b);                                   // (the first-block reductions start here, beside the weight gradient)
            launch_conv1_wgrad(c->dt, dvp(1), x0, n0, x1, G_(di_w(1)), G_(di_b(1)), c->partial_b, Bd, c->S, c->dC[1], L.b);
            wgrad_l(L.m, c->slab_b);
        }
    }
    if (want_wgrad && c->dt != DT_F32 && garena == nullptr) {
        L.join(L.a);
        L.join(L.b);
    } else if (want_wgrad) {
        // fp32: lane a is still busy with the last (largest-K) weight gradient when the input-gradient chain ends, so the main
        // lane is idle here: block 1's reductions run on it, and the waits for the lanes that ended earlier (lane b, the
        // pipelined Generator forward) are processed in that window too -- every wait is a barrier packet of its own (~5 us
        // each, one after the other), and only the one for lane a is left behind the last kernel
        launch_conv1_wgrad(c->dt, dvp(1), x0, n0, x1, G_(di_w(1)), G_(di_b(1)), c->partial_b, Bd, c->S, c->dC[1], L.m);
        L.join(L.b);
        if (L.tail_wait) L.wait(L.m, L.tail_wait);
        L.join(L.a);
    }
    if (want_dimage)
        launch_conv1_dgrad_tanh(c->dt, dvp(1), c->d_w1t, x0, c->dpre, Bd, c->S, c->dC[1], L.m);
}

// Backward through the Generator from d(pre-tanh) in c->dpre; fills the G gradient arena.  Lane m:
// BatchNorm backward and the input-gradient chain; lane a: the weight gradients.
static void g_backward_pass(siggan_ctx* c, Lanes& L, const float* z, int B) {
    const int Lg = c->Lg, S = c->S;
    int pre_rows = 0;              // partial rows of block l's BatchNorm-backward sums left by the input-gradient GEMM of block l+1
    for (int l = Lg; l >= 1; --l) {
        const int Hi = 4 << (l - 1), Ho = 2 * Hi, Ci = c->gC[l - 1], Co = c->gC[l];
        const int64_t R = (int64_t)B * Ho * Ho;
        // dy[l] is complete behind this block's BatchNorm backward: lane a's fork rides on that launch (four marker packets
        // fewer on the lane every kernel of this pass waits on)
        hipEvent_t const ef = L.fork_event(L.a);
        if (l == Lg) {   // final conv's input-gradient folded into this block's BatchNorm backward; its weight gradient rides in
                         // the same pass over y and its row sums stay on this lane (a 5 us kernel does not pay for a fork + join)
            launch_final_bwd_reduce(c->dt, c->dpre, c->wfin_t, c->g_y[l], B, S, Co, c->g_bn[l], c->partial, c->partial_b, L.m);
            launch_final_bn_bwd_apply(c->dt, c->dpre, c->wfin_t, c->g_y[l], c->g_da[l], B, S, Co, c->g_bn[l], c->partial, c->partial_b,
                                      GG(c, gi_fin_w(c)), GG(c, gi_fin_b(c)), GG(c, gi_bn_w(l)), GG(c, gi_bn_b(l)), L.m, ef);
        } else
            launch_bn_bwd(c->dt, c->g_da[l], c->g_y[l], R, Co, c->g_bn[l], c->partial, GG(c, gi_bn_w(l)), GG(c, gi_bn_b(l)), 0, L.m, pre_rows, ef);
        L.fork_after(L.a, ef);
        // weight gradient: small = block input a[l-1] (Hi), large = dy[l] (Ho)
        // (one fork per block; per two blocks measured the same, weight gradients on the main lane 2.5 % slower: DESIGN 4)
        WgradArgs w; memset(&w, 0, sizeof w); w.zeros = c->zeros; w.dt = c->dt;
        w.S = c->g_a[l - 1]; w.L = c->g_da[l]; w.slab = c->slab; w.dw = GG(c, gi_up_w(l)); w.B = B; w.Cs = Ci; w.Cl = Co;
        w.lgHs = ilog2i(Hi); w.lgWs = w.lgHs; w.lgCl = ilog2i(Co); w.K = B * Hi * Hi;
        const int max_splits = (int)(c->slab_floats / ((int64_t)Ci * (16 * Co + 1)));
        launch_wgrad(w, max_splits, L.a);
        // input gradient ("down" form): out = Cin at Hi, contract Cout over 16 taps
        GConvArgs a = gconv_args(c);
        a.in = c->g_da[l]; a.wp = c->g_dn[l]; a.out = c->g_da[l - 1];
        a.B = B; a.Hi = Ho; a.Wi = Ho; a.Ci = Co; a.Co = Ci;
        a.lgHr = ilog2i(Hi); a.lgWr = a.lgHr; a.Ho = Hi; a.Wo = Hi; a.form = 0; a.M = B * Hi * Hi; a.epi = EPI_RAW;
        if (l >= 2) {              // its output is d(relu output) of block l-1: that block's BatchNorm-backward sums ride in the epilogue
            a.epi = EPI_BN_BWD_STATS; a.aref = c->g_y[l - 1]; a.bnp = c->g_bn[l - 1]; a.stat0 = c->partial; a.stat_cap = PARTIAL_FLOATS;
        }
        pre_rows = launch_gconv(a, L.m);
    }
    if (!(c->fc_fused && launch_fc_bwd_fused(c->dt, c->g_da[0], c->fc_y, z, c->g_bn[0], GG(c, gi_fc_w()), GG(c, gi_fc_b()),
                                             GG(c, gi_bn0_w()), GG(c, gi_bn0_b()), B, c->latent, c->gC[0], L.m))) {
        launch_bn_bwd(c->dt, c->g_da[0], c->fc_y, B, c->F, c->g_bn[0], c->partial, GG(c, gi_bn0_w()), GG(c, gi_bn0_b()),
                      c->gC[0], L.m);
        launch_fc_wgrad(c->dt, c->g_da[0], z, GG(c, gi_fc_w()), GG(c, gi_fc_b()), B, c->latent, c->gC[0], L.m);
    }
    L.join(L.a);
}

// dropout multiplier tables of passes [p0, p1) (pass 0 = rows [0,B) = D(real), pass 1 = rows [B,2B) = D(fake)).
// ctr_add: draw as if the step counter were that much further (a pass generated ahead of its step).
static void make_noise(siggan_ctx* c, const float* masks, int B, int p0, int p1, hipStream_t s, uint32_t ctr_add = 0) {
    const float keep = 1.0f - c->cfg.dropout;
    int64_t sumC = 0;
    for (int l = 1; l <= c->Ld; ++l) sumC += c->dC[l];
    if (!masks) {
        float* out[8]; int64_t n[8], e0[8]; uint32_t sid[8];
        for (int l = 1; l <= c->Ld; ++l) {
            const int64_t nl = (int64_t)B * c->dC[l];
            out[l - 1] = c->d_noise[l] + p0 * nl; n[l - 1] = nl * (p1 - p0); e0[l - 1] = p0 * nl; sid[l - 1] = 16 + l;
        }
        launch_dropnoise_multi(c->Ld, out, n, e0, sid, keep, c->dev, s, ctr_add);
        return;
    }
    int64_t pre = 0;
    for (int l = 1; l <= c->Ld; ++l) {
        const int64_t n = (int64_t)B * c->dC[l];
        for (int p = p0; p < p1; ++p)
            launch_mask_to_noise(masks + (int64_t)p * B * sumC + (int64_t)B * pre, c->d_noise[l] + p * n, n, keep, s);
        pre += c->dC[l];
    }
}

static int check_hyper(const siggan_hyper* hp) {
    if (!hp) return fail(SIGGAN_E_INVALID, "null hyper-parameters");
    if (!(hp->lr >= 0.0) || !(hp->beta1 >= 0.0 && hp->beta1 < 1.0) || !(hp->beta2 >= 0.0 && hp->beta2 < 1.0) || !(hp->eps >= 0.0))
        return fail(SIGGAN_E_INVALID, "invalid Adam hyper-parameters");
    return SIGGAN_OK;
}

// ------------------------------------------------------------------------------------------
// the four step phases (inputs already staged in the workspace: c->real_stage, c->z, c->mask_stage)
// ------------------------------------------------------------------------------------------
static const float SN_EPS = 1e-12f;        // torch.nn.utils.spectral_norm's eps

// D step with a spectrally normalised Discriminator: every training forward runs one power iteration, so the real and the
// fake pass see different effective weights W / sigma_p.  The passes therefore run one after the other on the main lane
// (sigma -> weight packs -> forward), the backward is done per pass with that pass's packs into two arena-shaped
// temporaries, and k_sn_combine forms the gradient w.r.t. weight_orig through both sigmas.
static void phase_d_grads_sn(siggan_ctx* c, Lanes& L, const PhaseKey& k) {
    const int B = k.B;
    const bool drop = c->cfg.dropout > 0.f;
    repack(c, L.m, L.m, k.g_dirty != 0, false);
    if (k.pre_real)
        L.note(hipMemcpyAsync(c->real_stage, c->staged_src, (size_t)B * c->S * c->S * sizeof(float), hipMemcpyDeviceToDevice, L.m));
    if (drop) make_noise(c, k.has_masks ? c->mask_stage : nullptr, B, 0, 2, L.m);
    const float* fake = c->img;
    if (k.variant == SIGGAN_STEP_ABLATION) {
        g_forward_pass(c, k.has_zg ? c->z_g : nullptr, B, true, c->img_g, L.m, nullptr, nullptr, 2, c->z_g);
        L.record(c->ev_gfwd, L.m);
        fake = c->img_g;
    } else {
        g_forward_pass(c, k.has_z ? c->z : nullptr, B, false, c->img, L.m, nullptr, nullptr, 1, c->z);
    }
    launch_sn_sigma(c->snt, 1, 0, SN_EPS, L.m);                       // D(real): power iteration 1
    repack(c, L.m, L.m, false, true, 0);
    d_forward_rows(c, c->real_stage, 0, B, drop, L.m, c->slab_k);
    launch_sn_sigma(c->snt, 1, 1, SN_EPS, L.m);                       // D(fake): power iteration 2
    repack(c, L.m, L.m, false, true, 1);
    d_forward_rows(c, fake, B, B, drop, L.m, c->slab_k);
    launch_bce(c->logits, 2 * B, B, k.ls, 0.f, c->probs, c->dlogit, k.mt, 0, L.m, c->gscale);     // the step's metrics
    d_backward_pass(c, L, fake, B, fake, B, drop, true, false, BceSpec{B, 0.f, 0.f, nullptr, 0}, B, c->sn_g[1]);
    repack(c, L.m, L.m, false, true, 0);
    d_backward_pass(c, L, c->real_stage, B, c->real_stage, B, drop, true, false, BceSpec{B, k.ls, k.ls, nullptr, 0}, 0, c->sn_g[0]);
    c->snt.slot[0] = 0; c->snt.slot[1] = 1;
    launch_sn_combine(c->snt, c->sn_g[0], c->sn_g[1], c->st.d_grads, c->d_total, 2, L.m);
}

static void phase_d_grads(siggan_ctx* c, Lanes& L, const PhaseKey& k) {
    c->early_ar = false;             // set again by this pass when (and only when) it starts the tail's all-reduce itself
    if (c->sn) return phase_d_grads_sn(c, L, k);
    const int B = k.B;
    const bool drop = c->cfg.dropout > 0.f;
    // A staged step whose D(real) forward already ran has nothing for lane a: the real batch is read where it lies (no copy:
    // the borrowed tensor is valid until this call returns), the dropout tables of BOTH passes were drawn when that forward
    // was launched -- no fork, no join, two launches fewer in the step's first 100 us (+0.4 %)
    const bool nolane = k.pre_real == 2 && !k.d_dirty && (!drop || c->dreal_noise2);
    const float* xreal = c->real_stage;
    if (nolane) {
        repack(c, L.m, L.m, k.g_dirty != 0, false);
        xreal = c->staged_src;
    } else {
    L.fork(L.a);                                                     // lane a: D's packs, dropout tables, D(real)
    repack(c, L.m, L.a, k.g_dirty != 0, k.d_dirty != 0);
    if (k.pre_real)      // staged batch -> this step's real batch (the D backward reads it again)
        L.note(hipMemcpyAsync(c->real_stage, c->staged_src, (size_t)B * c->S * c->S * sizeof(float), hipMemcpyDeviceToDevice, L.a));
    if (drop && !(k.pre_real == 2 && c->dreal_noise2))
        make_noise(c, k.has_masks ? c->mask_stage : nullptr, B, k.pre_real == 2 ? 1 : 0, 2, L.a);
    // D(real) beside the Generator (train...py:309) -- unless the previous siggan_g_grads already ran it
    // (siggan_stage_real) beside its Generator backward; then bce only has to wait for that lane
    if (k.pre_real != 2) d_forward_rows(c, c->real_stage, 0, B, drop, L.a, c->slab_k2, true);
    }
    const float* fake = c->img;
    const bool spec_fwd = k.spec_g && k.variant != SIGGAN_STEP_ABLATION;
    // siggan_step_begin: the G step's training forward depends on nothing the D step changes.  16-bit contexts (every kernel
    // is a few microseconds: the step is a chain of launch latencies) start it HERE, beside the eval forward, with the eval
    // forward's activations in buffers of their own: bf16 batch 64 0.747 -> 0.723 ms.  fp32 keeps it behind D(fake) (below):
    // there the early start measured 0.7 % slower (round 3) / 0.1 % faster with fence-free events (round 4: noise) -- its
    // BatchNorm / fc kernels take matrix-pipe time from the eval forward, which is on the step's critical lane.
    const bool spec_early = spec_fwd && c->dt != DT_F32;
    hipEvent_t e_early = nullptr, e_eval = nullptr;
    if (spec_early) { e_early = L.next(); L.record(e_early, L.m); }
    if (k.variant == SIGGAN_STEP_ABLATION) {
        // ablation_vanilla_gan_signatures.py:397-448: both nets in train mode and ONE Generator forward per iteration --
        // BatchNorm batch statistics (+ running update), activations kept: the D half sees fake.detach(), the G half
        // back-propagates through the very same forward
        g_forward_pass(c, k.has_zg ? c->z_g : nullptr, B, true, c->img_g, L.m, nullptr, nullptr, 2, c->z_g);
        L.record(c->ev_gfwd, L.m);
        fake = c->img_g;
    } else {
        // (nolane: the pipelined forward below needs nothing but this pass -- its fork rides on the last kernel here)
        if (spec_fwd && !spec_early && nolane) e_eval = L.fork_event(c->s_c);
        g_forward_pass(c, k.has_z ? c->z : nullptr, B, false, c->img, L.m, nullptr, nullptr, 1, c->z, e_eval);   // G.eval(), no grad (train...py:314-315)
    }
    if (spec_early) {       // (enqueued behind the eval forward: the critical lane's kernels reach the dispatcher first)
        L.wait(c->s_c, e_early);
        g_forward_pass(c, k.has_zg ? c->z_g : nullptr, B, true, c->img_g, c->s_c, c->partial_c, c->slab_k3, 2, c->z_g);
        L.record(c->ev_gfwd, c->s_c);
    }
    if (!nolane) L.join(L.a);
    // siggan_step_begin: the G step's training forward depends on nothing the D step changes, so it
    // runs on its own lane beside D(fake) and the D step's backward (after the eval forward above: it
    // moves the BatchNorm running statistics and reuses the activation buffers; its own image / z /
    // scratch).  It forks HERE but is enqueued after D(fake), whose kernels the dispatcher should see first.
    hipEvent_t e_spec = nullptr;
    if (spec_fwd && !spec_early) { e_spec = e_eval; if (!e_spec) { e_spec = L.next(); L.record(e_spec, L.m); } }
    // the staged D(real) launch (lane c, previous G step) also drew THIS pass' dropout tables (dreal_noise2): the main lane
    // must be behind that lane before D(fake) reads them, not only before the backward pass reads the rows
    const bool join_early = k.pre_real == 2 && !c->dreal_joined && drop && c->dreal_noise2;
    if (join_early) L.wait(L.m, c->ev_dreal);
    d_forward_rows(c, fake, B, B, drop, L.m, c->slab_k, true);       // D(fake) into rows [B, 2B)
    if (k.pre_real == 2 && !c->dreal_joined && !join_early) L.wait(L.m, c->ev_dreal);
    c->dreal_joined = c->dreal_noise2 = false;
    if (spec_fwd && !spec_early) {
        L.wait(c->s_c, e_spec);
        g_forward_pass(c, k.has_zg ? c->z_g : nullptr, B, true, c->img_g, c->s_c, c->partial_c, c->slab_k2, 2, c->z_g);
        L.record(c->ev_gfwd, c->s_c);
    }
    // the pipelined forward has ended by the time the weight gradients have: the main lane waits for it inside the backward
    // pass' tail (fp32) / next to the two joins of this phase (16-bit), not between the optimiser and the G step's first kernel
    c->gfwd_joined = false;
    if (spec_fwd && c->dt == DT_F32) { L.tail_wait = c->ev_gfwd; c->gfwd_joined = true; }
    d_backward_pass(c, L, xreal, B, fake, 2 * B, drop, true, false, BceSpec{B, k.ls, 0.f, k.mt, 0}, 0, nullptr, k.coll != 0);
    L.tail_wait = nullptr;
    if (spec_fwd && c->dt != DT_F32) { L.wait(L.m, c->ev_gfwd); c->gfwd_joined = true; }
}

static void phase_g_grads(siggan_ctx* c, Lanes& L, const PhaseKey& k) {
    const int B = k.B;
    const float* zg; float* img;
    const bool d_pack = !c->sn && k.d_dirty != 0;                    // (spectral norm: the packs follow sigma, below)
    bool conv1_done = false;
    if (k.spec_g) {                                                  // forward already enqueued by siggan_step_begin
        if (!c->gfwd_joined) L.wait(L.m, c->ev_gfwd);
        c->gfwd_joined = false;
        // trainer step: the first-block forward of the new images (rows [B, 2B), no dropout in this pass) rides in the launch
        // that re-packs D's weights -- it reads the raw block-1 weights, and the two would stand back to back on this lane
        // (k.rode: both already happened in the D update's launch, k_adam_pack)
        conv1_done = k.rode || (d_pack && k.variant != SIGGAN_STEP_ABLATION);
        repack(c, L.m, L.m, false, d_pack, 0, (conv1_done && !k.rode) ? c->img_g : nullptr, B, B);
        zg = c->z_g; img = c->img_g;
    } else {
        L.fork(L.a);
        repack(c, L.m, L.a, k.g_dirty != 0, d_pack);
        g_forward_pass(c, k.has_z ? c->z : nullptr, B, true, c->img, L.m, nullptr, nullptr, 2, c->z);   // G.train(): BN batch stats (train...py:349)
        L.join(L.a);
        zg = c->z; img = c->img;
    }
    // trainer step: D.eval() -- dropout off, target 1.0 (train...py:350,360).  Ablation step: D stays in train mode -- a
    // fresh set of dropout masks -- and the target is the smoothed real label (ablation...py:441-442)
    const bool abl = k.variant == SIGGAN_STEP_ABLATION;
    const bool gdrop = abl && c->cfg.dropout > 0.f;
    if (c->sn) {        // D.eval() (trainer step): sigma from the stored u, v; D.train() (ablation step): a third power iteration
        launch_sn_sigma(c->snt, abl ? 1 : 0, 2, SN_EPS, L.m);
        repack(c, L.m, L.m, false, true, 2);
    }
    if (gdrop) {
        int64_t sumC = 0;
        for (int l = 1; l <= c->Ld; ++l) sumC += c->dC[l];
        make_noise(c, k.has_masks ? c->mask_stage + (int64_t)2 * B * sumC : nullptr, B, 0, 1, L.m);
    }
    const float gy = abl ? k.ls : 1.0f;
    // The G step's Discriminator pass runs in workspace rows [B, 2B) -- free since the D step's backward -- so that the NEXT D
    // step's D(real) forward (rows [0, B), siggan_stage_real) can start right here, beside this pass's forward and
    // input-gradient chain (one B-sized GEMM at a time leaves half the chip's wave slots idle), not only beside the Generator
    // backward: +1.3 % (round 3).  (ablation step: its dropout tables are drawn for rows [0, B); spectral norm: no staging)
    const int r0g = (!abl && !c->sn) ? B : 0;
    c->g_r0 = r0g;
    const bool real_early = k.pre_real && r0g != 0;
    if (real_early) {
        const bool drop = c->cfg.dropout > 0.f;
        L.fork(c->s_c);                                               // D's packs are complete on m here
        if (drop) make_noise(c, nullptr, B, 0, 2, c->s_c, 1);       // the D(fake) pass' tables too (this G step's pass has no dropout)
        c->dreal_noise2 = drop;
        d_forward_rows(c, c->staged_src, 0, B, drop, c->s_c, c->slab_k2, true);
        L.record(c->ev_dreal, c->s_c);
    }
    d_forward_rows(c, img, r0g, B, gdrop, L.m, c->slab_k, true, conv1_done);
    d_backward_pass(c, L, img, B, img, B, gdrop, false, true, BceSpec{B, gy, gy, k.mt, 1}, r0g);   // through D into the image; no D weight grads
    if (k.pre_real && !real_early) {
        // siggan_stage_real: the NEXT D step's D(real) forward needs the Discriminator as it is now (its
        // update is behind us) and the activation rows this step is done with: run it on lane c beside the
        // Generator backward.  Its dropout tables are drawn for the step counter that step will see (+1:
        // the Generator update in between ticks once).
        const bool drop = c->cfg.dropout > 0.f;
        L.fork(c->s_c);
        if (drop) make_noise(c, nullptr, B, 0, 1, c->s_c, 1);
        d_forward_rows(c, c->staged_src, 0, B, drop, c->s_c, c->slab_k2, true);
        L.record(c->ev_dreal, c->s_c);
    }
    g_backward_pass(c, L, zg, B);
    // 16-bit contexts: join the early D(real) lane HERE, next to the join of the weight-gradient lane, so that the next D step
    // does not stop for it behind D(fake)
    if (real_early && c->dt != DT_F32) { L.wait(L.m, c->ev_dreal); c->dreal_joined = true; }
}

static void phase_apply(siggan_ctx* c, Lanes& L, const PhaseKey& k) {
    const int which = k.phase == 2 ? 1 : 0;                          // phase 2 = D apply, 3 = G apply
    float* p = which == 0 ? c->st.g_params : c->st.d_params;
    float* g = which == 0 ? c->st.g_grads : c->st.d_grads;
    float* m = which == 0 ? c->st.g_exp_avg : c->st.d_exp_avg;
    float* v = which == 0 ? c->st.g_exp_avg_sq : c->st.d_exp_avg_sq;
    float* steps = which == 0 ? c->st.g_adam_steps : c->st.d_adam_steps;
    const int64_t n = which == 0 ? c->g_total : c->d_total;
    const int nt = (int)(which == 0 ? c->g_off.size() : c->d_off.size());
    const bool clip = k.clip > 0.f;
    // the arena holds gscale x the gradient (fp16 chains; 1 otherwise): the optimiser's multiplier takes it out again, and
    // the gradient is written back unscaled (as torch leaves a clipped .grad)
    float gs = k.gs / c->gscale;
    if (c->comm) {
        // the data-parallel exchange: ONE sum all-reduce of the network's flat gradient bucket over RCCL, in place, on the
        // step's own stream (whatever runs on the side lanes -- the pipelined Generator forward behind the D bucket, the
        // staged D(real) forward behind the G bucket -- overlaps it); the mean is taken by the optimiser's multiplier
        // a failed (or earlier failed) exchange leaves this rank's arena unreduced: the optimiser must not step on it --
        // the update is skipped and lane_check hands the error to the caller (sticky: every later call returns it until
        // siggan_comm_destroy)
        if (c->comm_err) return;
        if (which == 1 && c->early_ar) {
            // the tail of the arena (last block + classifier) went ahead (d_backward_pass): the head now
            const int64_t o = c->d_off[di_w(c->Ld)];
            const int e = rccl()->AllReduce(g, g, (size_t)o, NCCL_FLOAT32, NCCL_SUM, c->comm, L.m);
            c->early_ar = false;
            if (e != NCCL_SUCCESS) { c->comm_err = e; return; }
            L.wait(L.m, c->ev_ar);
        } else {
        const int e = rccl()->AllReduce(g, g, (size_t)n, NCCL_FLOAT32, NCCL_SUM, c->comm, L.m);
        if (e != NCCL_SUCCESS) { c->comm_err = e; return; }
        }
        gs *= 1.0f / (float)c->comm_world;
    }
    const bool guard = c->dt == DT_F16;                              // static gradient scale: skip the update on an overflow
    if (clip || guard) launch_grad_sumsq(g, n, c->dev, c->partial, L.m);
    if (k.fused_t > 0.0 && k.pack) {
        // ... and the same launch rebuilds what the next pass derives from the arena (weight packs, eval tables): no
        // k_prepare between the update and the forward pass that follows it on the step's critical lane
        ApTable t;
        ApRide rd; memset(&rd, 0, sizeof rd);
        if (k.ride) {      // the pending G step's first Discriminator block (siggan_step_begin's images; rows [B, 2B)) rides along
            if (!c->gfwd_joined) { L.wait(L.m, c->ev_gfwd); c->gfwd_joined = true; }
            const int64_t H = c->S >> 1;
            rd.x = c->img_g; rd.out = c->d_a[1] + (size_t)((int64_t)k.ride * H * H * c->dC[1]) * c->es; rd.B = k.ride; rd.S = c->S;
            rd.dt = c->dt; rd.slope = c->cfg.leaky_slope; rd.w_off = c->d_off[di_w(1)]; rd.b_off = c->d_off[di_b(1)];
            rd.counter = (unsigned*)c->ride_ctr;
        }
        const bool ok = (which == 0 ? ap_table_g(c, t) : ap_table_d(c, t, k.ride ? k.ride * (c->S / 4) : 0)) &&
            launch_adam_pack(t, p, g, m, v, c->dev, steps, nt, k.fused_t, k.lr, k.beta1, k.beta2, k.eps, gs, k.clip,
                             k.mt + (which == 0 ? SIGGAN_M_G_GRAD_NORM : SIGGAN_M_D_GRAD_NORM), clip ? c->partial : nullptr,
                             k.mt + (which == 0 ? SIGGAN_M_G_SKIPPED : SIGGAN_M_D_SKIPPED), BN_EPS, k.ride ? &rd : nullptr, L.m);
        if (!ok) c->lane_err = hipErrorInvalidValue;                    // (apply_common checked the table: not reached)
        return;
    }
    if (k.fused_t > 0.0) {
        // ONE launch: the host knows the step count (apply_common), so the bias corrections are kernel arguments and
        // k_adam_prepare (a 5 us kernel plus a kernel boundary on the step's critical lane, twice per step) is not needed
        launch_adam_fused(p, g, m, v, n, c->dev, steps, nt, k.fused_t, k.lr, k.beta1, k.beta2, k.eps, gs, k.clip,
                          k.mt + (which == 0 ? SIGGAN_M_G_GRAD_NORM : SIGGAN_M_D_GRAD_NORM), clip ? c->partial : nullptr, L.m,
                          k.mt + (which == 0 ? SIGGAN_M_G_SKIPPED : SIGGAN_M_D_SKIPPED));
        return;
    }
    launch_adam_prepare(c->dev, steps, nt, k.lr, k.beta1, k.beta2, gs, k.clip,
                        k.mt + (which == 0 ? SIGGAN_M_G_GRAD_NORM : SIGGAN_M_D_GRAD_NORM), L.m, guard ? 1 : 0,
                        k.mt + (which == 0 ? SIGGAN_M_G_SKIPPED : SIGGAN_M_D_SKIPPED), (clip || guard) ? c->partial : nullptr);
    launch_adam(p, g, m, v, n, c->dev, k.beta1, k.beta2, k.eps, (clip || gs != 1.0f) ? 1 : 0, L.m);
}

static void run_phase_body(siggan_ctx* c, Lanes& L, const PhaseKey& k) {
    if (k.phase == 0) phase_d_grads(c, L, k);
    else if (k.phase == 1) phase_g_grads(c, L, k);
    else phase_apply(c, L, k);
}

// Enqueue one phase behind everything on the caller's stream u.  Eager: directly on u (side lanes
// forked from it).  Graph mode: the phase is captured once per distinct key on the library's own
// main lane (a caller stream may be the legacy default stream, which cannot be captured) and replayed.
static int run_phase(siggan_ctx* c, const PhaseKey& k, hipStream_t u) {
    const bool overlap = (c->mode & SIGGAN_MODE_OVERLAP) != 0;
    const bool graph = (c->mode & SIGGAN_MODE_GRAPH) != 0 && g_prof == nullptr;
    if (graph && c->comm && k.phase >= 2) return fail(SIGGAN_E_STATE, "SIGGAN_MODE_GRAPH is not available with a communicator");
    if (graph && c->sn) return fail(SIGGAN_E_STATE, "SIGGAN_MODE_GRAPH is not available with spectral normalisation");
    if (!graph) {
        const bool ext = overlap && g_prof == nullptr;
        Lanes L{c, u, overlap ? c->s_a : u, overlap ? c->s_b : u, ext, nullptr};
        run_phase_body(c, L, k);
        LAUNCHCHK();
        return lane_check(c);
    }
    hipGraphExec_t exec = nullptr;
    for (auto& e : c->graphs)
        if (e.first == k) { exec = e.second; break; }
    if (!exec) {
        Lanes L{c, c->s_m, overlap ? c->s_a : c->s_m, overlap ? c->s_b : c->s_m, false, nullptr};
        hipGraph_t g = nullptr;
        HIPCHK(hipStreamBeginCapture(c->s_m, hipStreamCaptureModeRelaxed));
        run_phase_body(c, L, k);
        hipError_t e1 = hipStreamEndCapture(c->s_m, &g);
        if (e1 != hipSuccess || !g) return fail(SIGGAN_E_HIP, "stream capture failed: %s", hipGetErrorString(e1));
        hipError_t e2 = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e2 != hipSuccess) return fail(SIGGAN_E_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e2));
        if (c->graphs.size() >= 64) { (void)hipGraphExecDestroy(c->graphs.front().second); c->graphs.erase(c->graphs.begin()); }
        c->graphs.emplace_back(k, exec);
    }
    hipEvent_t e_in = c->ev_bridge[0], e_out = c->ev_bridge[1];
    HIPCHK(hipEventRecord(e_in, u));
    HIPCHK(hipStreamWaitEvent(c->s_m, e_in, 0));
    HIPCHK(hipGraphLaunch(exec, c->s_m));
    HIPCHK(hipEventRecord(e_out, c->s_m));
    HIPCHK(hipStreamWaitEvent(u, e_out, 0));
    return SIGGAN_OK;
}

static PhaseKey make_key(siggan_ctx* c, int phase, int B, bool has_z, bool has_masks, const siggan_hyper* hp,
                         float* metrics_dev) {
    PhaseKey k; memset(&k, 0, sizeof k);
    k.phase = phase; k.B = B; k.has_z = has_z; k.has_masks = has_masks;
    k.mt = metrics_dev ? metrics_dev : c->metrics;
    if (phase <= 1) { k.g_dirty = c->g_dirty; k.d_dirty = c->d_dirty; k.ls = hp->label_smoothing; }
    else {
        k.lr = hp->lr; k.beta1 = hp->beta1; k.beta2 = hp->beta2; k.eps = hp->eps;
        k.clip = hp->clip_max_norm > 0.f ? hp->clip_max_norm : 0.f;
        k.gs = hp->grad_scale > 0.f ? hp->grad_scale : 1.0f;
    }
    return k;
}

static int finish_metrics(siggan_ctx* c, float* metrics_dev, float* metrics_host, hipStream_t s) {
    if (metrics_host) {
        HIPCHK(hipMemcpyAsync(metrics_host, metrics_dev ? metrics_dev : c->metrics, SIGGAN_M_COUNT * sizeof(float),
                              hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    return SIGGAN_OK;
}

// ------------------------------------------------------------------------------------------
// public passes
// ------------------------------------------------------------------------------------------
extern "C" int siggan_set_step_variant(siggan_ctx* c, int32_t variant) {
    if (!c) return fail(SIGGAN_E_INVALID, "null context");
    if (variant != SIGGAN_STEP_TRAINER && variant != SIGGAN_STEP_ABLATION) return fail(SIGGAN_E_INVALID, "unknown step variant %d", variant);
    if (c->g_fwd_pending || c->pending) return fail(SIGGAN_E_STATE, "a step is in flight");
    c->variant = variant;
    return SIGGAN_OK;
}

extern "C" int siggan_set_mode(siggan_ctx* c, int32_t mode) {
    if (!c) return fail(SIGGAN_E_INVALID, "null context");
    c->mode = mode;
    return SIGGAN_OK;
}

extern "C" int siggan_g_forward(siggan_ctx* c, const float* z_dev, int32_t batch, int32_t training, float* images_dev,
                                void* stream) {
    ENTER(c);
    int rc = check_call(c, batch);
    if (rc) return rc;
    if (training && (rc = check_bn_batch(c, batch))) return rc;
    if (!z_dev || !images_dev) return fail(SIGGAN_E_INVALID, "null tensor");
    hipStream_t s = (hipStream_t)stream;
    if ((rc = settle(c, s))) return rc;
    repack(c, s, s, c->g_dirty, !c->sn && c->d_dirty);
    c->g_dirty = false; if (!c->sn) c->d_dirty = false;
    g_forward_pass(c, z_dev, batch, training != 0, images_dev, s);
    if (training) c->g_dirty = true;   // running statistics moved: the eval-mode tables are stale
    LAUNCHCHK();
    return lane_check(c);
}

extern "C" int siggan_d_forward(siggan_ctx* c, const float* x_dev, int32_t batch, int32_t training, const float* masks_dev,
                                float* probs_dev, float* features_dev, void* stream) {
    ENTER(c);
    int rc = check_call(c, batch);
    if (rc) return rc;
    if (!x_dev || (!probs_dev && !features_dev)) return fail(SIGGAN_E_INVALID, "null tensor");
    hipStream_t s = (hipStream_t)stream;
    drop_dreal(c);                     // the activation rows of a D(real) forward started ahead of time are overwritten
    c->conv1_rode = 0;                 // ... and those of a first-block forward the last D apply ran ahead
    if ((rc = settle(c, s))) return rc;
    if (c->sn) launch_sn_sigma(c->snt, training != 0, 2, SN_EPS, s);   // train(): one power iteration (u, v move), as torch's hook
    repack(c, s, s, c->g_dirty, c->sn || c->d_dirty, 2);
    c->g_dirty = c->d_dirty = false;
    const bool drop = training != 0 && c->cfg.dropout > 0.f;
    if (drop) { if (!masks_dev) launch_tick(c->dev, s); make_noise(c, masks_dev, batch, 0, 1, s); }
    d_forward_rows(c, x_dev, 0, batch, drop, s, c->slab_k);
    if (probs_dev) launch_bce(c->logits, batch, batch, 0.f, 0.f, probs_dev, nullptr, nullptr, 0, s);
    if (features_dev) launch_cls_features(c->dt, c->d_a[c->Ld], features_dev, batch, c->dC[c->Ld], s);
    LAUNCHCHK();
    return lane_check(c);
}

static int d_grads_common(siggan_ctx* c, const float* real_dev, int32_t batch, const float* z_dev, const float* masks_dev,
                          const siggan_hyper* hp, float* metrics_dev, void* stream, bool spec_g, const float* zg_dev,
                          bool apply_follows = false) {
    ENTER(c);
    int rc = check_call(c, batch);
    if (rc) return rc;
    if ((rc = check_hyper(hp))) return rc;
    if (!real_dev && c->staged_B != batch)
        return fail(SIGGAN_E_INVALID, "null real batch (and no batch of %d images staged by siggan_stage_real)", batch);
    if (!c->st.d_grads) return fail(SIGGAN_E_STATE, "gradient arena was not bound");
    if (c->g_fwd_pending) return fail(SIGGAN_E_STATE, "siggan_step_begin must be followed by siggan_g_grads before the next D step");
    // the speculative forward needs its own lane: without overlap (or under graph replay) it is skipped
    if (spec_g && ((c->mode & SIGGAN_MODE_OVERLAP) == 0 || (c->mode & SIGGAN_MODE_GRAPH) != 0 || g_prof != nullptr)) spec_g = false;
    const bool abl = c->variant == SIGGAN_STEP_ABLATION;
    // spectral norm, trainer step: phase_d_grads_sn runs its passes one after the other on the main lane and has no
    // pipelined Generator forward -- siggan_g_grads then runs the training forward itself (an explicit zg_dev waits in
    // z_g, zg_stash)
    if (c->sn && !abl) spec_g = false;
    // one sample: the reference's D step runs (G.eval()) and its G step then refuses the batch -- no forward ahead of that
    if (batch == 1 && !abl) spec_g = false;
    if (abl) {                                 // one z per iteration: it belongs to the (single, training-mode) Generator forward
        if ((rc = check_bn_batch(c, batch))) return rc;
        if (zg_dev) return fail(SIGGAN_E_INVALID, "the ablation step has one latent batch per iteration: pass it as z_dev");
        zg_dev = z_dev; z_dev = nullptr; spec_g = true;
    }
    hipStream_t s = (hipStream_t)stream;
    const int B = batch;
    const size_t img_bytes = (size_t)B * c->S * c->S * sizeof(float);
    // stage the caller's tensors into fixed workspace slots (captured phases must see fixed addresses)
    int pre_real = 0;                          // 1: the staged batch is this step's real batch; 2: and its D(real) forward is done
    if (!real_dev) pre_real = (c->dreal_B == B && !masks_dev) ? 2 : 1;
    if (pre_real == 2) c->dreal_B = 0;         // consumed: the phase waits for ev_dreal where it needs the rows
    else drop_dreal(c);                        // D(real) is redone (explicit batch / masks): the early one is abandoned
    if ((rc = settle(c, s))) return rc;
    if (real_dev && real_dev != c->real_stage) HIPCHK(hipMemcpyAsync(c->real_stage, real_dev, img_bytes, hipMemcpyDeviceToDevice, s));
    c->staged_B = 0;
    if (z_dev && z_dev != c->z) HIPCHK(hipMemcpyAsync(c->z, z_dev, (size_t)B * c->latent * sizeof(float), hipMemcpyDeviceToDevice, s));
    int64_t sumC = 0;
    for (int l = 1; l <= c->Ld; ++l) sumC += c->dC[l];
    if (masks_dev) HIPCHK(hipMemcpyAsync(c->mask_stage, masks_dev, (size_t)(abl ? 3 : 2) * B * sumC * sizeof(float), hipMemcpyDeviceToDevice, s));
    c->abl_masks = abl && masks_dev != nullptr;
    if (zg_dev) HIPCHK(hipMemcpyAsync(c->z_g, zg_dev, (size_t)B * c->latent * sizeof(float), hipMemcpyDeviceToDevice, s));
    c->zg_stash = (!spec_g && zg_dev) ? B : 0;
    PhaseKey k = make_key(c, 0, B, z_dev != nullptr, masks_dev != nullptr, hp, metrics_dev);
    c->metrics_last = k.mt;
    k.spec_g = spec_g; k.has_zg = spec_g && zg_dev != nullptr; k.pre_real = pre_real; k.variant = c->variant;
    k.coll = apply_follows && c->comm != nullptr && (c->mode & SIGGAN_MODE_GRAPH) == 0;   // (no collective inside a captured phase)
    if ((rc = run_phase(c, k, s))) return rc;
    c->g_dirty = c->d_dirty = false;
    if (spec_g) { c->g_fwd_pending = B; c->g_dirty = true; }   // running statistics moved
    c->pending = 1;
    return SIGGAN_OK;
}

extern "C" int siggan_stage_real(siggan_ctx* c, const float* real_dev, int32_t batch, void* stream) {
    ENTER(c);
    int rc = check_call(c, batch);
    if (rc) return rc;
    if (!real_dev) return fail(SIGGAN_E_INVALID, "null real batch");
    drop_dreal(c);                     // an early D(real) forward of a previously staged batch may still be reading that batch
    if ((rc = settle(c, (hipStream_t)stream))) return rc;
    // borrowed, not copied (a 1 MB copy on the step's critical lane): the caller keeps the tensor alive and unmodified until the
    // D step that consumes it has returned (include/siggan.h); that step copies it into the workspace on a side lane
    c->staged_src = real_dev;
    c->staged_B = batch;
    return SIGGAN_OK;
}

extern "C" int siggan_d_grads(siggan_ctx* c, const float* real_dev, int32_t batch, const float* z_dev, const float* masks_dev,
                              const siggan_hyper* hp, float* metrics_dev, void* stream) {
    return d_grads_common(c, real_dev, batch, z_dev, masks_dev, hp, metrics_dev, stream, false, nullptr);
}

extern "C" int siggan_step_begin(siggan_ctx* c, const float* real_dev, int32_t batch, const float* z_dev, const float* masks_dev,
                                 const float* zg_dev, const siggan_hyper* hp, float* metrics_dev, void* stream) {
    return d_grads_common(c, real_dev, batch, z_dev, masks_dev, hp, metrics_dev, stream, true, zg_dev, true);
}

static int apply_common(siggan_ctx* c, int which, const siggan_hyper* hp, float* metrics_dev, float* metrics_host, void* stream) {
    ENTER(c);
    int rc = check_call(c, 1);
    if (rc) return rc;
    if ((rc = check_hyper(hp))) return rc;
    if (c->pending != (which == 1 ? 1 : 2))
        return fail(SIGGAN_E_STATE, "siggan_%c_apply without a preceding siggan_%c_grads", which ? 'd' : 'g', which ? 'd' : 'g');
    const float* need[] = {which ? c->st.d_grads : c->st.g_grads, which ? c->st.d_exp_avg : c->st.g_exp_avg,
                           which ? c->st.d_exp_avg_sq : c->st.g_exp_avg_sq, which ? c->st.d_adam_steps : c->st.g_adam_steps};
    for (const float* p : need)
        if (!p) return fail(SIGGAN_E_STATE, "gradient / Adam arenas were not bound");
    hipStream_t s = (hipStream_t)stream;
    if ((rc = settle(c, s))) return rc;
    // metrics of a step live in ONE buffer: the one the *_grads call named (its losses are already there)
    if (!metrics_dev) metrics_dev = c->metrics_last != c->metrics ? c->metrics_last : nullptr;
    else if (c->metrics_last && metrics_dev != c->metrics_last)
        HIPCHK(hipMemcpyAsync(metrics_dev, c->metrics_last, SIGGAN_M_COUNT * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    PhaseKey k = make_key(c, which == 1 ? 2 : 3, 0, false, false, hp, metrics_dev);
    // One-launch update whenever the step count can live on the host: not under graph replay (arguments are baked into the
    // captured launch) and not with the fp16 overflow guard (a skipped update leaves the device-side count behind).  The count
    // is read back once after siggan_bind / siggan_params_changed and tracked from then on.
    const int wi = which == 1 ? 1 : 0;
    if ((c->mode & SIGGAN_MODE_GRAPH) == 0 && c->dt != DT_F16) {
        if (!c->adam_t_known[wi]) {
            float t0 = 0.f;
            HIPCHK(hipStreamSynchronize(s));
            HIPCHK(hipMemcpy(&t0, which == 1 ? c->st.d_adam_steps : c->st.g_adam_steps, sizeof t0, hipMemcpyDeviceToHost));
            c->adam_t[wi] = (double)t0; c->adam_t_known[wi] = true;
        }
        k.fused_t = c->adam_t[wi] + 1.0;
        ApTable probe;
        k.pack = (which == 0 ? ap_table_g(c, probe) : ap_table_d(c, probe, 0)) ? 1 : 0;
        // trainer step with the G step's training forward already enqueued (siggan_step_begin): its first Discriminator block
        // needs nothing but the updated block-1 weights -- it rides in the update's launch (phase_g_grads then skips it).
        // fp32 only: 1.4201 -> 1.4105 ms; at bf16 the separate launch measured better (0.6216 vs 0.6245 ms)
        if (which == 1 && k.pack && c->g_fwd_pending && c->variant != SIGGAN_STEP_ABLATION && g_prof == nullptr && c->dt == DT_F32)
            k.ride = c->g_fwd_pending;
    } else {
        c->adam_t_known[wi] = false;
    }
    if ((rc = run_phase(c, k, s))) return rc;
    if (k.fused_t > 0.0) c->adam_t[wi] = k.fused_t;
    // (k.pack: the launch that updated the arena also rebuilt what is derived from it, with the running statistics as the
    // G step's training forward left them)
    if (which == 0) c->g_dirty = !k.pack; else c->d_dirty = !k.pack;
    if (which == 1) c->conv1_rode = k.ride;
    c->pending = 0;
    return finish_metrics(c, metrics_dev, metrics_host, s);
}

extern "C" int siggan_d_apply(siggan_ctx* c, const siggan_hyper* hp, float* metrics_dev, float* metrics_host, void* stream) {
    return apply_common(c, 1, hp, metrics_dev, metrics_host, stream);
}

extern "C" int siggan_d_step(siggan_ctx* c, const float* real_dev, int32_t batch, const float* z_dev, const float* masks_dev,
                             const siggan_hyper* hp, float* metrics_dev, float* metrics_host, void* stream) {
    int rc = d_grads_common(c, real_dev, batch, z_dev, masks_dev, hp, metrics_dev, stream, false, nullptr, true);
    if (rc) return rc;
    return siggan_d_apply(c, hp, metrics_dev, metrics_host, stream);
}

extern "C" int siggan_g_grads(siggan_ctx* c, int32_t batch, const float* z_dev, const siggan_hyper* hp, float* metrics_dev,
                              void* stream) {
    ENTER(c);
    int rc = check_call(c, batch);
    if (rc) return rc;
    if ((rc = check_bn_batch(c, batch))) return rc;
    if ((rc = check_hyper(hp))) return rc;
    if (!c->st.g_grads) return fail(SIGGAN_E_STATE, "gradient arena was not bound");
    hipStream_t s = (hipStream_t)stream;
    if ((rc = settle(c, s))) return rc;
    const int B = batch;
    const bool spec = c->g_fwd_pending != 0;
    if (c->variant == SIGGAN_STEP_ABLATION && !spec)
        return fail(SIGGAN_E_STATE, "ablation step: siggan_g_grads follows the siggan_d_grads / siggan_d_apply of the same iteration");
    if (spec && (c->g_fwd_pending != B || z_dev))
        return fail(SIGGAN_E_STATE, "siggan_g_grads after siggan_step_begin must use the same batch and no explicit z (pass it to step_begin)");
    if (!spec && !z_dev && c->zg_stash == B) z_dev = c->z_g;          // z given to a step_begin that could not pipeline
    c->zg_stash = 0;
    if (!spec && z_dev && z_dev != c->z) HIPCHK(hipMemcpyAsync(c->z, z_dev, (size_t)B * c->latent * sizeof(float), hipMemcpyDeviceToDevice, s));
    PhaseKey k = make_key(c, 1, B, z_dev != nullptr, false, hp, metrics_dev);
    c->metrics_last = k.mt;
    k.spec_g = spec; k.variant = c->variant; k.has_masks = c->abl_masks;
    k.rode = (spec && c->conv1_rode == B && !c->d_dirty) ? 1 : 0;
    c->conv1_rode = 0;
    c->abl_masks = false;
    // a staged next batch: start its D(real) forward beside this Generator backward (own lane: eager overlap mode only)
    k.pre_real = c->staged_B == B && c->dreal_B == 0 && (c->mode & SIGGAN_MODE_OVERLAP) != 0 && (c->mode & SIGGAN_MODE_GRAPH) == 0 &&
                 g_prof == nullptr && c->pending == 0 && !c->sn;
    if ((rc = run_phase(c, k, s))) return rc;
    if (k.pre_real) c->dreal_B = B;
    c->g_fwd_pending = 0;
    c->d_dirty = false;
    c->g_dirty = true;                 // the training forward moved the BatchNorm running statistics
    c->pending = 2;
    return SIGGAN_OK;
}

extern "C" int siggan_g_apply(siggan_ctx* c, const siggan_hyper* hp, float* metrics_dev, float* metrics_host, void* stream) {
    return apply_common(c, 0, hp, metrics_dev, metrics_host, stream);
}

extern "C" int siggan_g_step(siggan_ctx* c, int32_t batch, const float* z_dev, const siggan_hyper* hp, float* metrics_dev,
                             float* metrics_host, void* stream) {
    int rc = siggan_g_grads(c, batch, z_dev, hp, metrics_dev, stream);
    if (rc) return rc;
    return siggan_g_apply(c, hp, metrics_dev, metrics_host, stream);
}

// ------------------------------------------------------------------------------------------
// operator-level entry points
// ------------------------------------------------------------------------------------------
static bool pow2(int v) { return v > 0 && (v & (v - 1)) == 0; }

extern "C" int siggan_op_conv4x4s2(siggan_ctx* c, int32_t form, const void* in_dev, const float* w_dev, void* out_dev,
                                   int32_t batch, int32_t h_in, int32_t c_in, int32_t c_out, void* stream) {
    if (!c) return fail(SIGGAN_E_INVALID, "null context");
    if (!in_dev || !w_dev || !out_dev) return fail(SIGGAN_E_INVALID, "null tensor");
    if (form != 0 && form != 1) return fail(SIGGAN_E_INVALID, "form must be 0 (down) or 1 (up)");
    if (!pow2(h_in) || !pow2(c_in) || !pow2(c_out) || c_in < 32 || c_out < 32 || c_in > 512 || c_out > 512 || batch < 1)
        return fail(SIGGAN_E_INVALID, "shape not in the model family (pow2 dims, 32 <= C <= 512)");
    if (form == 0 && h_in < 2) return fail(SIGGAN_E_INVALID, "down form needs h_in >= 2");
    DevGuard dg_(c->cfg.device); HIPCHK(dg_.err);
    hipStream_t s = (hipStream_t)stream;
    GConvArgs a = gconv_args(c);
    a.in = in_dev; a.wp = c->op_pack; a.out = out_dev; a.B = batch; a.Hi = h_in; a.Wi = h_in; a.Ci = c_in; a.Co = c_out;
    a.form = form; a.epi = EPI_RAW;
    PrepTable t; t.njobs = 0; t.overflow = 0;
    PrepJob j; memset(&j, 0, sizeof j);
    j.src = w_dev; j.dst = (float*)c->op_pack; j.dt = c->dt;
    if (form == 0) {
        j.type = PREP_PACK_DOWN; j.O = c_out; j.I = c_in;
        a.Ho = a.Wo = h_in / 2; a.lgHr = a.lgWr = ilog2i(h_in / 2); a.M = batch * (h_in / 2) * (h_in / 2);
    } else {
        j.type = PREP_PACK_UP; j.I = c_in; j.O = c_out;
        a.Ho = a.Wo = 2 * h_in; a.lgHr = a.lgWr = ilog2i(h_in); a.M = batch * h_in * h_in;
    }
    prep_add(t, j, (long long)c_in * c_out * 16);
    if (!launch_prepare(t, BN_EPS, s)) return fail(SIGGAN_E_STATE, "prepare table overflow");
    launch_gconv(a, s);
    LAUNCHCHK();
    return SIGGAN_OK;
}

extern "C" int siggan_op_conv4x4s2_wgrad(siggan_ctx* c, const void* small_dev, const void* large_dev, float* dw_dev,
                                         int32_t batch, int32_t h_small, int32_t c_small, int32_t c_large, void* stream) {
    if (!c) return fail(SIGGAN_E_INVALID, "null context");
    if (!small_dev || !large_dev || !dw_dev) return fail(SIGGAN_E_INVALID, "null tensor");
    if (!pow2(h_small) || !pow2(c_small) || !pow2(c_large) || c_small < 32 || c_large < 32 || c_small > 512 || c_large > 512 ||
        batch < 1)
        return fail(SIGGAN_E_INVALID, "shape not in the model family (pow2 dims, 32 <= C <= 512)");
    DevGuard dg_(c->cfg.device); HIPCHK(dg_.err);
    hipStream_t s = (hipStream_t)stream;
    WgradArgs w; memset(&w, 0, sizeof w); w.zeros = c->zeros; w.dt = c->dt;
    w.S = small_dev; w.L = large_dev; w.slab = c->slab; w.dw = dw_dev; w.B = batch; w.Cs = c_small; w.Cl = c_large;
    w.lgHs = w.lgWs = ilog2i(h_small); w.lgCl = ilog2i(c_large); w.K = batch * h_small * h_small;
    const int max_splits = (int)(c->slab_floats / ((int64_t)c_small * (16 * c_large + 1)));
    launch_wgrad(w, max_splits, s);
    LAUNCHCHK();
    return SIGGAN_OK;
}

extern "C" int siggan_op_adam(siggan_ctx* c, float* p, float* g, float* m, float* v, int64_t n, int32_t step,
                              const siggan_hyper* hp, void* stream) {
    if (!c) return fail(SIGGAN_E_INVALID, "null context");
    int rc = check_hyper(hp);
    if (rc) return rc;
    if (!p || !g || !m || !v || n < 1 || step < 1) return fail(SIGGAN_E_INVALID, "bad argument");
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return fail(SIGGAN_E_INVALID, "arenas must be 16-byte aligned");
    DevGuard dg_(c->cfg.device); HIPCHK(dg_.err);
    hipStream_t s = (hipStream_t)stream;
    float* steps = c->partial + ((2 << 20) - 64);          // scratch slot: step count before the increment
    const float prev = (float)(step - 1);
    HIPCHK(hipMemcpyAsync(steps, &prev, sizeof prev, hipMemcpyHostToDevice, s));
    const bool clip = hp->clip_max_norm > 0.f;
    const float gs = hp->grad_scale > 0.f ? hp->grad_scale : 1.0f;
    if (clip) launch_grad_sumsq(g, n, c->dev, c->partial, s);
    launch_adam_prepare(c->dev, steps, 1, hp->lr, hp->beta1, hp->beta2, gs, hp->clip_max_norm, nullptr, s, 0, nullptr,
                        clip ? c->partial : nullptr);
    launch_adam(p, g, m, v, n, c->dev, hp->beta1, hp->beta2, hp->eps, (clip || gs != 1.0f) ? 1 : 0, s);
    LAUNCHCHK();
    return SIGGAN_OK;
}

// ------------------------------------------------------------------------------------------
// data-parallel communicator
// ------------------------------------------------------------------------------------------
extern "C" int siggan_comm_unique_id(void* id_out) {
    if (!id_out) return fail(SIGGAN_E_INVALID, "null argument");
    Rccl* r = rccl();
    if (r->err) return fail(SIGGAN_E_STATE, "%s", r->err);
    ncclUniqueId_t id;
    NCCLCHK(r->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return SIGGAN_OK;
}
extern "C" int siggan_comm_init(siggan_ctx* c, int32_t rank, int32_t world, const void* id) {
    ENTER(c);
    if (!id || world < 1 || rank < 0 || rank >= world) return fail(SIGGAN_E_INVALID, "bad rank / world / id");
    if (c->comm) return fail(SIGGAN_E_STATE, "a communicator is already initialised");
    Rccl* r = rccl();
    if (r->err) return fail(SIGGAN_E_STATE, "%s", r->err);
    ncclUniqueId_t uid;
    memcpy(&uid, id, sizeof uid);
    NCCLCHK(r->CommInitRank(&c->comm, world, uid, rank));
    c->comm_rank = rank; c->comm_world = world;
    return SIGGAN_OK;
}
extern "C" int siggan_comm_destroy(siggan_ctx* c) {
    ENTER(c);
    if (!c->comm) return SIGGAN_OK;
    HIPCHK(hipDeviceSynchronize());
    NCCLCHK(rccl()->CommDestroy(c->comm));
    c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1; c->comm_err = 0; c->early_ar = false;
    return SIGGAN_OK;
}
extern "C" int32_t siggan_comm_world(const siggan_ctx* c) { return c ? c->comm_world : -1; }
extern "C" int siggan_comm_broadcast(siggan_ctx* c, void* buf_dev, int64_t bytes, int32_t root, void* stream) {
    ENTER(c);
    if (!buf_dev || bytes < 1 || root < 0 || root >= c->comm_world) return fail(SIGGAN_E_INVALID, "bad argument");
    if (!c->comm) return SIGGAN_OK;                      // a single replica already holds the root's bytes
    NCCLCHK(rccl()->Broadcast(buf_dev, buf_dev, (size_t)bytes, NCCL_CHAR, root, c->comm, (hipStream_t)stream));
    return SIGGAN_OK;
}

extern "C" int siggan_device_info(int32_t device, int32_t* compute_units, int32_t* clock_khz, int64_t* hbm_bytes) {
    if (!compute_units || !clock_khz || !hbm_bytes) return fail(SIGGAN_E_INVALID, "null argument");
    hipDeviceProp_t p;
    HIPCHK(hipGetDeviceProperties(&p, device));
    *compute_units = p.multiProcessorCount; *clock_khz = p.clockRate; *hbm_bytes = (int64_t)p.totalGlobalMem;
    return SIGGAN_OK;
}

static Prof g_prof_store;
extern "C" int siggan_prof_enable(siggan_ctx* c, int32_t on) {
    if (!c) return fail(SIGGAN_E_INVALID, "null context");
    DevGuard dg_(c->cfg.device); HIPCHK(dg_.err);
    HIPCHK(hipDeviceSynchronize());
    g_prof_store.clear();
    g_prof = on ? &g_prof_store : nullptr;
    return SIGGAN_OK;
}
extern "C" int32_t siggan_prof_slots(void) { return Prof::NID - 1; }
extern "C" int siggan_prof_read(siggan_ctx* c, int32_t idx, char* name, int32_t name_cap, int64_t* launches, double* ms,
                                double* flops, double* bytes) {
    if (!c || !name || !launches || !ms || !flops || !bytes || idx < 0 || idx >= Prof::NID - 1) return fail(SIGGAN_E_INVALID, "bad argument");
    DevGuard dg_(c->cfg.device); HIPCHK(dg_.err);
    HIPCHK(hipDeviceSynchronize());
    snprintf(name, name_cap, "%s", Prof::name(idx));
    *launches = 0; *ms = 0.0; *flops = 0.0; *bytes = 0.0;
    for (auto& r : g_prof_store.recs) {
        if (r.id != idx) continue;
        float t = 0.f;
        HIPCHK(hipEventElapsedTime(&t, r.e0, r.e1));
        *launches += 1; *ms += t; *flops += r.flops; *bytes += r.bytes;
    }
    return SIGGAN_OK;
}

extern "C" int siggan_debug_tensor(siggan_ctx* c, const char* name, int32_t idx, float* out_dev, int64_t n, void* stream) {
    if (!c || !name || !out_dev || n < 1) return fail(SIGGAN_E_INVALID, "bad argument");
    const void* src = nullptr; int64_t capv = 0;
    const void** ptr = &src; int64_t* cap = &capv;
    bool typed = true;                          // element type dt (converted to fp32 on the way out) vs. plain fp32
    const int64_t Bm = c->Bm, Bd = 2 * Bm, SS = (int64_t)c->S * c->S;
    auto gsz = [&](int l) { const int64_t H = 4 << l; return Bm * H * H * c->gC[l]; };
    auto dsz = [&](int l) { const int64_t H = c->S >> l; return Bd * H * H * c->dC[l]; };
    const bool gl = idx >= 0 && idx <= c->Lg, dl = idx >= 1 && idx <= c->Ld;
    if (!strcmp(name, "g_y") && gl) { *ptr = c->g_y[idx]; *cap = gsz(idx); }
    else if (!strcmp(name, "g_a") && gl) { *ptr = c->g_a[idx]; *cap = gsz(idx); }
    else if (!strcmp(name, "g_da") && gl) { *ptr = c->g_da[idx]; *cap = gsz(idx); }
    else if (!strcmp(name, "d_a") && dl) { *ptr = c->d_a[idx]; *cap = dsz(idx); }
    else if (!strcmp(name, "d_dv") && dl) { *ptr = c->d_dv[idx]; *cap = dsz(idx); }
    else if (!strcmp(name, "d_a_g") && dl) {     // the Discriminator activations of the last G step (they may start at row B)
        const int64_t H = c->S >> idx, off = (int64_t)c->g_r0 * H * H * c->dC[idx];
        *ptr = c->d_a[idx] + (size_t)off * c->es; *cap = dsz(idx) - off;
    }
    else if (!strcmp(name, "z")) { *ptr = c->z; *cap = Bm * c->latent; typed = false; }
    else if (!strcmp(name, "z_g")) { *ptr = c->z_g; *cap = Bm * c->latent; typed = false; }
    else if (!strcmp(name, "img")) { *ptr = c->img; *cap = Bm * SS; typed = false; }
    else if (!strcmp(name, "dpre")) { *ptr = c->dpre; *cap = Bm * SS; typed = false; }
    else if (!strcmp(name, "logits")) { *ptr = c->logits; *cap = Bd; typed = false; }
    else if (!strcmp(name, "probs")) { *ptr = c->probs; *cap = Bd; typed = false; }
    else if (!strcmp(name, "dlogit")) { *ptr = c->dlogit; *cap = Bd; typed = false; }
    else return fail(SIGGAN_E_INVALID, "unknown debug tensor %s[%d]", name, idx);
    if (n > capv) return fail(SIGGAN_E_INVALID, "debug tensor %s[%d] holds %lld floats, %lld asked", name, idx, (long long)capv, (long long)n);
    DevGuard dg_(c->cfg.device); HIPCHK(dg_.err);
    if (!strcmp(name, "g_a") && idx == c->Lg && c->ga_last_B) {
        // after a training forward the last block's activation exists only as y + the BatchNorm table: form it now
        const int64_t H = c->S;
        HIPCHK(hipDeviceSynchronize());
        launch_bn_relu(c->dt, c->g_y[idx], c->g_a[idx], (int64_t)c->ga_last_B * H * H, c->gC[idx], c->g_bn[idx], (hipStream_t)stream);
        LAUNCHCHK();
    }
    if (typed && c->dt != DT_F32) { launch_to_f32(c->dt, src, out_dev, n, (hipStream_t)stream); LAUNCHCHK(); }
    else HIPCHK(hipMemcpyAsync(out_dev, src, (size_t)n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return SIGGAN_OK;
}

extern "C" int siggan_augment_batch(int32_t device, const uint8_t* cache_dev, int64_t n_images, const int32_t* index_dev,
                                    const int32_t* params_dev, const int16_t* tables_dev, const float* lut_dev, float* out_dev,
                                    int32_t batch, int32_t size, int32_t augment, int32_t fill, void* stream) {
    if (!cache_dev || !index_dev || !lut_dev || !out_dev) return fail(SIGGAN_E_INVALID, "null tensor");
    if (augment && (!params_dev || !tables_dev)) return fail(SIGGAN_E_INVALID, "augment needs params and tables");
    if (batch < 1 || n_images < 1 || size < 1 || size > 1024) return fail(SIGGAN_E_INVALID, "bad batch / cache / image size");
    if (fill < 0 || fill > 255) return fail(SIGGAN_E_INVALID, "fill must be a byte value");
    DevGuard dg_(device); HIPCHK(dg_.err);
    launch_augment(cache_dev, n_images, index_dev, params_dev, tables_dev, lut_dev, out_dev, batch, size, augment != 0, fill, (hipStream_t)stream);
    LAUNCHCHK();
    return SIGGAN_OK;
}

extern "C" int siggan_op_randn(siggan_ctx* c, float* out_dev, int64_t n, void* stream) {
    if (!c || !out_dev || n < 1) return fail(SIGGAN_E_INVALID, "bad argument");
    DevGuard dg_(c->cfg.device); HIPCHK(dg_.err);
    hipStream_t s = (hipStream_t)stream;
    launch_tick(c->dev, s);
    launch_randn(out_dev, n, c->dev, 3, s);
    LAUNCHCHK();
    return SIGGAN_OK;
}
