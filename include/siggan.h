/*
 * siggan.h -- C ABI of the MI355X (gfx950) signature-GAN engine.
 *
 * The reference (Nobita421/signature-Gan) has no FFI/plugin interface: its G+D train step and
 * its generation path sit behind Python classes.  This header is the boundary a maintainer
 * binds (ctypes) to route exactly that path to hand-written HIP kernels:
 *
 *   entry point              replaces (file:line under /root/reference/src)
 *   -----------------------  ---------------------------------------------------------------
 *   siggan_g_forward         Generator.forward            generator_vanilla_gan.py:189-209
 *                            (VanillaGAN.generate vanilla_gan_model.py:338-371,
 *                             generate_signatures_batch utils/inference.py:171-182)
 *   siggan_d_forward         Discriminator.forward / forward_features
 *                                                        discriminator_vanilla_gan.py:241-274
 *   siggan_d_step            VanillaGAN.train_discriminator_step vanilla_gan_model.py:180-252
 *                            == GANTrainer._train_discriminator  train_vanilla_gan_signatures.py:281-337
 *   siggan_g_step            VanillaGAN.train_generator_step    vanilla_gan_model.py:254-306
 *                            == GANTrainer._train_generator     train_vanilla_gan_signatures.py:339-376
 *   siggan_d_grads/_apply,   the same two steps cut at the point where a data-parallel run
 *   siggan_g_grads/_apply    averages the flat gradient bucket across ranks: with a communicator
 *                            (siggan_comm_init) *_apply itself issues the ncclAllReduce on the
 *                            step's stream ahead of the optimiser; without one, a host may reduce
 *                            the bound gradient arena between the two halves
 *   siggan_op_*              single kernels of the path (operator-level tests / profiling)
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer named *_dev is DEVICE memory of the context's
 *     GPU, fp32 unless stated; images are (B,1,S,S) contiguous (NCHW == NHWC for one channel).
 *   - ownership: the CALLER owns every tensor it passes (parameters, gradients, Adam moments,
 *     BatchNorm buffers, inputs, outputs) and their lifetime; the library owns only its
 *     workspace.  Parameter storage is BORROWED by siggan_bind: four flat arenas per network
 *     laid out in the reference's parameters() order, so the torch-side state_dict()/optimizer
 *     state stay authoritative and checkpoints need no conversion.
 *   - every call enqueues on `stream` (a hipStream_t passed as void*); no hidden device-wide
 *     synchronisation except where a HOST output pointer (metrics_host) is given.
 *   - return value: 0 = OK, negative = SIGGAN_E_*; never throws, never aborts.
 *     siggan_last_error() returns a thread-local message for the last failure.
 *   - one context per device per process; a context is not re-entrant.
 */
#ifndef SIGGAN_H
#define SIGGAN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SIGGAN_ABI_VERSION 3   /* 2: siggan_stage_real, siggan_augment_batch; 3: siggan_config.dtype, siggan_rng_state, siggan_comm_* */

enum {
    SIGGAN_OK = 0,
    SIGGAN_E_INVALID = -1,     /* bad argument (maps to ValueError in the Python shim) */
    SIGGAN_E_STATE = -2,       /* call sequence / binding missing                      */
    SIGGAN_E_HIP = -3,         /* HIP runtime error                                    */
    SIGGAN_E_NOMEM = -4
};

typedef struct siggan_ctx siggan_ctx;

/* Model geometry: Generator(latent_dim, output_size) / Discriminator(input_size)
 * (generator_vanilla_gan.py:97-121, discriminator_vanilla_gan.py:111-128). */
typedef struct siggan_config {
    int32_t device;         /* HIP device ordinal */
    int32_t latent_dim;     /* z dimension (reference default 100) */
    int32_t image_size;     /* 64 or 128 -- anything else is SIGGAN_E_INVALID (ValueError) */
    int32_t image_channels; /* only 1 (grayscale signatures) is built */
    int32_t max_batch;      /* largest per-call batch the workspace is sized for */
    float   dropout;        /* Discriminator dropout p (default 0.25) */
    float   leaky_slope;    /* LeakyReLU slope (default 0.2) */
    uint64_t seed;          /* seed of the library's counter-based RNG (z, dropout masks) */
    int32_t dtype;          /* SIGGAN_DTYPE_*: storage type of the library-owned activations / activation gradients and of
                             * the weight copies the MFMA kernels read.  F32 (0) is the reference's arithmetic and the parity
                             * path.  BF16 / F16 are build-defined narrow variants (BASELINE.json configs[2] / configs[4]; the
                             * reference has no reduced-precision mode, vanilla_gan_model.py:107-120): 16-bit storage, 16-bit
                             * MFMA operands, fp32 accumulation, fp32 master weights / BatchNorm statistics / losses / weight
                             * gradients / Adam.  Every caller-visible tensor stays fp32 in all modes. */
    float   f16_grad_scale; /* F16 only: power-of-two factor carried by the backward chains so that small activation
                             * gradients stay above the fp16 subnormals; removed again inside the optimiser step (the bound
                             * *_grads arenas hold scale x gradient between *_grads and *_apply).  0 = default (1024).
                             * The scale is STATIC (no dynamic loss scaling).  Protection against an overflow: in F16 mode
                             * *_apply takes the gradient arena's sum of squares first, and when it is not finite (an inf / NaN
                             * activation gradient reached a weight gradient) the whole update is SKIPPED -- parameters, moments
                             * and step counts untouched -- and SIGGAN_M_D_SKIPPED / SIGGAN_M_G_SKIPPED is set to 1 (else 0). */
    int32_t spectral_norm;  /* != 0: torch.nn.utils.spectral_norm on every Discriminator conv and on the classifier
                             * (Discriminator(use_spectral_norm=True), discriminator_vanilla_gan.py:60-62,200-202): the d_params
                             * arena then holds weight_orig, siggan_storage.d_sn_u / d_sn_v the weight_u / weight_v buffers; every
                             * training-mode Discriminator forward runs one power iteration (u, v updated in place), every forward
                             * divides the weights by sigma = u . (W v), and the D step's gradients flow through sigma. */
} siggan_config;
enum { SIGGAN_DTYPE_F32 = 0, SIGGAN_DTYPE_BF16 = 1, SIGGAN_DTYPE_F16 = 2 };

/* Borrowed storage.  *_params / *_grads / *_exp_avg / *_exp_avg_sq: flat fp32 arenas holding
 * the network's parameters() in reference order (sizes: siggan_param_count).  *_adam_steps: one
 * fp32 per parameter TENSOR (torch.optim.Adam keeps `step` per tensor).  g_bn_*: BatchNorm
 * running statistics of all G BatchNorm layers concatenated in module order (fc.1, then
 * upsample_blocks.i.block.1); g_bn_batches: one int64 per BatchNorm layer. */
typedef struct siggan_storage {
    float *g_params, *g_grads, *g_exp_avg, *g_exp_avg_sq, *g_adam_steps;
    float *g_bn_running_mean, *g_bn_running_var;
    int64_t *g_bn_batches;
    float *d_params, *d_grads, *d_exp_avg, *d_exp_avg_sq, *d_adam_steps;
    /* spectral norm only (else NULL): weight_u of all layers (conv blocks in order, then the classifier) concatenated --
     * sizes Cout_l, 1 -- and weight_v likewise -- sizes Cin_l*16, 512*16 (siggan_sn_count) */
    float *d_sn_u, *d_sn_v;
} siggan_storage;

/* Optimiser / loss hyper-parameters of one step (vanilla_gan_model.py:60-72,
 * train_vanilla_gan_signatures.py:63-79). */
typedef struct siggan_hyper {
    double lr, beta1, beta2, eps; /* Adam, as the Python doubles torch.optim.Adam holds (eps: torch default
                                   * 1e-8); 1 - beta and the bias corrections are formed in double, as torch does */
    float label_smoothing;        /* real-label value in the D step (0.9) */
    float clip_max_norm;          /* <= 0: gradient clipping disabled */
    float grad_scale;             /* multiplies the gradients before clip/Adam (1/world for DP sums; 1 otherwise) */
} siggan_hyper;

/* metrics written by the step calls (fp32 each) */
enum {
    SIGGAN_M_D_LOSS = 0, SIGGAN_M_D_LOSS_REAL, SIGGAN_M_D_LOSS_FAKE, SIGGAN_M_D_REAL_MEAN,
    SIGGAN_M_D_FAKE_MEAN, SIGGAN_M_D_REAL_ACC, SIGGAN_M_D_FAKE_ACC, SIGGAN_M_D_GRAD_NORM,
    SIGGAN_M_G_LOSS, SIGGAN_M_G_FAKE_MEAN, SIGGAN_M_G_GRAD_NORM,
    SIGGAN_M_D_SKIPPED, SIGGAN_M_G_SKIPPED,    /* F16 only: 1 when the update was skipped for a non-finite gradient, else 0 */
    SIGGAN_M_COUNT = 16
};

/* ---- lifecycle ------------------------------------------------------------------------- */
int siggan_abi_version(void);
const char *siggan_last_error(void);
int siggan_create(const siggan_config *cfg, siggan_ctx **out);
int siggan_destroy(siggan_ctx *ctx);

/* which: 0 = generator, 1 = discriminator */
int64_t siggan_param_count(const siggan_ctx *ctx, int which);          /* scalars in the flat arena */
int32_t siggan_param_tensors(const siggan_ctx *ctx, int which);        /* number of parameter tensors */
/* offset (in floats) and element count of parameter tensor `idx` inside the flat arena */
int siggan_param_span(const siggan_ctx *ctx, int which, int32_t idx, int64_t *offset, int64_t *numel);
int64_t siggan_bn_count(const siggan_ctx *ctx);                        /* floats in g_bn_running_* */
int64_t siggan_sn_count(const siggan_ctx *ctx, int which);             /* floats in d_sn_u (which = 0) / d_sn_v (1) */
int32_t siggan_bn_layers(const siggan_ctx *ctx);
int64_t siggan_workspace_bytes(const siggan_ctx *ctx);

int siggan_bind(siggan_ctx *ctx, const siggan_storage *st);
/* tell the library the caller changed parameters / BN buffers / Adam step tensors behind its back
 * (load_state_dict, optimizer.load_state_dict, manual edits): packed weight copies are rebuilt on next
 * use, and the Adam step counts (which the library otherwise tracks on the host, so that an update is ONE
 * launch) are read back from *_adam_steps before the next *_apply */
int siggan_params_changed(siggan_ctx *ctx);
int siggan_seed(siggan_ctx *ctx, uint64_t seed, uint64_t offset);
/* reads the RNG position back (seed, call counter): a caller that re-creates a context (larger max_batch) or resumes a
 * run hands them to siggan_seed so the z / dropout stream continues instead of restarting; the reference's equivalent is
 * torch's global generator, which simply keeps running (train_vanilla_gan_signatures.py:313,356).  Synchronises. */
int siggan_rng_state(siggan_ctx *ctx, uint64_t *seed, uint64_t *offset);
/* execution mode of the step phases (default: SIGGAN_MODE_OVERLAP): SIGGAN_MODE_GRAPH replays each phase as a
 * hipGraph captured once per distinct (batch, flags, hyper-parameters); SIGGAN_MODE_OVERLAP runs
 * the weight-gradient / reduction kernels on side streams beside the input-gradient chain. */
#define SIGGAN_MODE_GRAPH 1
#define SIGGAN_MODE_OVERLAP 2
int siggan_set_mode(siggan_ctx *ctx, int32_t mode);

/* Which of the reference's two G+D iterations the step calls implement (default SIGGAN_STEP_TRAINER):
 *   TRAINER   GANTrainer._train_discriminator / _train_generator == VanillaGAN.train_*_step (train_vanilla_gan_signatures.py:
 *             281-376): D step with G in eval mode on its own z; G step with D in eval mode (no dropout), its own z, target 1.
 *   ABLATION  AblationGANTrainer.train_epoch (ablation_vanilla_gan_signatures.py:397-467): both nets stay in train mode; ONE
 *             Generator forward per iteration (BatchNorm batch statistics) whose detached output feeds the D update and
 *             through which the G update back-propagates; the G update's D pass draws fresh dropout masks and its target
 *             is the smoothed real label (hp->label_smoothing of siggan_g_grads).  Call order per iteration:
 *             siggan_d_grads(real, z, masks) -> siggan_d_apply -> siggan_g_grads(batch, z = NULL) -> siggan_g_apply; explicit
 *             masks are three sets (real pass, fake pass of the D update, fake pass of the G update). */
enum { SIGGAN_STEP_TRAINER = 0, SIGGAN_STEP_ABLATION = 1 };
int siggan_set_step_variant(siggan_ctx *ctx, int32_t variant);

/* ---- forward passes ---------------------------------------------------------------------- */
/* z_dev (B,latent) -> images_dev (B,1,S,S) in [-1,1].  training!=0: BatchNorm batch statistics,
 * running stats and num_batches_tracked updated (nn.Module.train()); 0: running stats (eval).
 * A training-mode Generator forward of ONE sample is refused with SIGGAN_E_INVALID and torch's text ("Expected more than 1
 * value per channel when training ...": the fc block's BatchNorm1d, generator_vanilla_gan.py:112) -- here, in siggan_g_grads
 * and in the ablation variant's siggan_d_grads; the trainer variant's D step accepts one sample (G.eval(), no BatchNorm in D),
 * and siggan_step_begin then does not start the Generator forward ahead of the siggan_g_grads that refuses it. */
int siggan_g_forward(siggan_ctx *ctx, const float *z_dev, int32_t batch, int32_t training,
                     float *images_dev, void *stream);

/* x_dev (B,1,S,S) -> probs_dev (B) probabilities.  features_dev (B,512*4*4, reference
 * flatten order c,h,w) optional.  training!=0 enables Dropout2d: masks_dev, if given, holds the
 * keep masks (1 keep / 0 drop) of all blocks concatenated [(B,C_1),(B,C_2),...]; NULL draws
 * them from the library RNG. */
int siggan_d_forward(siggan_ctx *ctx, const float *x_dev, int32_t batch, int32_t training,
                     const float *masks_dev, float *probs_dev, float *features_dev, void *stream);

/* ---- training steps ---------------------------------------------------------------------- */
/* D step: D(real) vs label_smoothing, G_eval(z) under no-grad, D(fake) vs 0, backward into D,
 * optional clip, Adam(D).  z_dev NULL: z ~ N(0,1) from the library RNG.  masks_dev NULL: library
 * RNG; else keep masks for the real pass then the fake pass, each [(B,C_1)...(B,C_n)].
 * metrics_dev (SIGGAN_M_COUNT floats, optional) receives the metrics on the device;
 * metrics_host (optional) additionally copies them to the host and synchronises the stream. */
int siggan_d_step(siggan_ctx *ctx, const float *real_dev, int32_t batch, const float *z_dev,
                  const float *masks_dev, const siggan_hyper *hp, float *metrics_dev,
                  float *metrics_host, void *stream);
/* G step: G_train(z) (BN batch stats + running-stat update), D_eval(fake), BCE vs 1.0, backward
 * through D into G (D weight gradients are not formed), optional clip, Adam(G). */
int siggan_g_step(siggan_ctx *ctx, int32_t batch, const float *z_dev, const siggan_hyper *hp,
                  float *metrics_dev, float *metrics_host, void *stream);

/* ---- data parallelism: one process per GPU, per-replica BatchNorm, ONE sum all-reduce of the flat
 * gradient bucket per network per step over RCCL (xGMI), issued INSIDE siggan_d_apply / siggan_g_apply -- hence inside
 * siggan_d_step / siggan_g_step -- once a communicator exists.  The reference is single-process (no torch.distributed
 * anywhere); this is the definition SURVEY 8(e) fixes: every rank runs the reference's step (train_vanilla_gan_signatures.py:
 * 281-376) on its contiguous shard of the global batch, gradients are averaged (the optimiser multiplies the summed bucket by
 * grad_scale / world), clipping acts on the averaged gradient, metrics stay per rank.
 *   siggan_comm_unique_id  rank 0 draws the 128-byte id (ncclGetUniqueId) and hands it to the other ranks by any channel the
 *                          launcher has (a file, MPI, a TCP store; the Python shim uses torch.distributed's store once)
 *   siggan_comm_init       every rank, same id: ncclCommInitRank on the context's device.  world == 1 is allowed (the
 *                          all-reduce then runs and changes nothing) -- used by the single-GPU tests
 *   siggan_comm_broadcast  root's bytes to every rank (initial parameters / optimiser state), in place
 * RCCL is resolved at run time from the copy the process already holds (torch's), see siggan.hip; the library owns the
 * communicator and destroys it with the context.  Not available under SIGGAN_MODE_GRAPH. */
#define SIGGAN_COMM_ID_BYTES 128
int siggan_comm_unique_id(void *id_out);
int siggan_comm_init(siggan_ctx *ctx, int32_t rank, int32_t world, const void *id);
int siggan_comm_destroy(siggan_ctx *ctx);
int32_t siggan_comm_world(const siggan_ctx *ctx);   /* 1 without a communicator */
int siggan_comm_broadcast(siggan_ctx *ctx, void *buf_dev, int64_t bytes, int32_t root, void *stream);

/* step halves: *_grads leaves the local gradient in the bound *_grads arena (and the forward metrics in metrics_dev);
 * *_apply (all-reduces the arena when a communicator exists, then) scales by hp->grad_scale (/ world), clips and runs
 * Adam.  Without a communicator a caller may reduce the arena itself between the halves (the gloo tests do). */
int siggan_d_grads(siggan_ctx *ctx, const float *real_dev, int32_t batch, const float *z_dev,
                   const float *masks_dev, const siggan_hyper *hp, float *metrics_dev, void *stream);
/* siggan_d_grads that ALSO enqueues the following G step's training forward (z: zg_dev, or the library
 * RNG when NULL) on its own lane beside the D step's backward -- that forward depends on nothing the D
 * step changes (train_vanilla_gan_signatures.py:349-357).  Must be followed by siggan_d_apply and then
 * siggan_g_grads(batch, z_dev = NULL), which picks the forward up.  Results are bit-identical to the
 * un-pipelined calls; without SIGGAN_MODE_OVERLAP it degrades to siggan_d_grads.
 * With a communicator this call (like siggan_d_step, and unlike siggan_d_grads, which never starts a collective) already
 * issues the all-reduce of the last Discriminator block's weight gradient -- three quarters of the D bucket, complete first
 * -- on a lane of its own under the rest of the backward pass; siggan_d_apply reduces the remainder and waits for it.  EVERY
 * rank must therefore follow it with siggan_d_apply. */
int siggan_step_begin(siggan_ctx *ctx, const float *real_dev, int32_t batch, const float *z_dev,
                      const float *masks_dev, const float *zg_dev, const siggan_hyper *hp,
                      float *metrics_dev, void *stream);
/* Software-pipelining across steps: hands the library the real batch of the NEXT D step (BORROWED, not
 * copied: real_dev must stay valid and unmodified until the siggan_d_grads / siggan_step_begin that consumes
 * it has returned -- that call copies it into the workspace; the loop of GANTrainer.train_epoch,
 * train_vanilla_gan_signatures.py:378-405, knows the batch one iteration early and its loader keeps it alive).  Call it after siggan_d_apply and before siggan_g_grads: that siggan_g_grads then
 * runs the staged batch's D(real) forward (discriminator weights are final by then) on its own lane beside
 * the Generator backward.  The next siggan_d_grads / siggan_step_begin consumes the staged batch when
 * called with real_dev = NULL (a non-NULL real_dev discards it).  Results are bit-identical to the
 * un-staged sequence as long as exactly one siggan_g_apply lies in between; without SIGGAN_MODE_OVERLAP
 * (or under graph replay) the staged batch is simply used as the real batch. */
int siggan_stage_real(siggan_ctx *ctx, const float *real_dev, int32_t batch, void *stream);
int siggan_d_apply(siggan_ctx *ctx, const siggan_hyper *hp, float *metrics_dev, float *metrics_host,
                   void *stream);
int siggan_g_grads(siggan_ctx *ctx, int32_t batch, const float *z_dev, const siggan_hyper *hp,
                   float *metrics_dev, void *stream);
int siggan_g_apply(siggan_ctx *ctx, const siggan_hyper *hp, float *metrics_dev, float *metrics_host,
                   void *stream);

/* ---- operator-level entry points (tests, profiling) --------------------------------------- */
/* Activation tensors NHWC on the device in the CONTEXT's element type (fp32, or bf16 / f16 for a narrow context);
 * weights and weight gradients always fp32 in the reference's layouts.  4x4 stride-2 pad-1 convolution family on MFMA:
 *   form 0 "down": out[n,oh,ow,co] = sum_{kh,kw,ci} in[n,2oh-1+kh,2ow-1+kw,ci] * w[co,ci,kh,kw]
 *                  (Conv2d forward; ConvTranspose2d input-gradient)            w is (Cout,Cin,4,4)
 *   form 1 "up"  : out[n,oh,ow,co] = sum_{ci,kh,kw: oh=2ih-1+kh} in[n,ih,iw,ci] * w[ci,co,kh,kw]
 *                  (ConvTranspose2d forward; Conv2d input-gradient)            w is (Cin,Cout,4,4)
 * w_dev is in the reference's (torch) layout; the library packs it. */
int siggan_op_conv4x4s2(siggan_ctx *ctx, int32_t form, const void *in_dev, const float *w_dev,
                        void *out_dev, int32_t batch, int32_t h_in, int32_t c_in, int32_t c_out,
                        void *stream);
/* weight gradient of the same family: dw[cs,cl,kh,kw] = sum_{n,p,q} small[n,p,q,cs] *
 * large[n,2p-1+kh,2q-1+kw,cl]; dw_dev in torch layout (Cs,Cl,4,4). */
int siggan_op_conv4x4s2_wgrad(siggan_ctx *ctx, const void *small_dev, const void *large_dev,
                              float *dw_dev, int32_t batch, int32_t h_small, int32_t c_small,
                              int32_t c_large, void *stream);
/* fused Adam over a flat arena (torch.optim.Adam single step, step = count AFTER increment) */
int siggan_op_adam(siggan_ctx *ctx, float *p_dev, float *g_dev, float *m_dev, float *v_dev,
                   int64_t n, int32_t step, const siggan_hyper *hp, void *stream);
/* library RNG: n standard normals / n Bernoulli(keep) keep-masks */
int siggan_op_randn(siggan_ctx *ctx, float *out_dev, int64_t n, void *stream);
/* Input pipeline (replaces SignatureDataset.__getitem__ + the transform chain of
 * get_train_transforms / get_val_transforms, data_loader_signatures.py:107-138,153-243, for a whole batch):
 * out[b] = Normalize(ToTensor(hflip?(RandomAffine_scale(RandomRotation(cache[index[b]])))))  as (batch,1,size,size)
 * fp32 in HBM, from an (N,size,size) uint8 cache of decoded + resized images resident in HBM.  Needs no context.
 *   params_dev [batch][8] int32: {mode, a0, a1, a2, a3, a4, a5, flags} -- rotation stage: mode 0 = copy, 1 = Pillow's
 *     16.16 fixed-point affine (output (x,y) samples input ((a2 + y*a1 + x*a0) >> 16, (a5 + y*a4 + x*a3) >> 16)),
 *     2 = per-axis tables (rows 0,1 of tables_dev); flags bit 0: horizontal flip, bit 1: scale stage present
 *   tables_dev [batch][4][size] int16: source column / row per output column / row (-1 = outside -> fill); rows 2,3 are
 *     the scale stage (Pillow's ImagingScaleAffine positions, tabulated by the host in double precision)
 *   lut_dev [256] fp32: value of each byte after ToTensor + Normalize;  augment = 0: out[b] = lut[cache[index[b]]].
 * The host side (signature-gan_amd/data_loader_signatures.py) draws indices, angles and scales with the reference
 * DataLoader's own RNG protocol and fills params / tables. */
int siggan_augment_batch(int32_t device, const uint8_t *cache_dev, int64_t n_images, const int32_t *index_dev,
                         const int32_t *params_dev, const int16_t *tables_dev, const float *lut_dev,
                         float *out_dev, int32_t batch, int32_t size, int32_t augment, int32_t fill,
                         void *stream);
/* what the roofline peaks are derived from (bench.py): compute units x shader clock (kHz) of the device, and its HBM size */
int siggan_device_info(int32_t device, int32_t *compute_units, int32_t *clock_khz, int64_t *hbm_bytes);
/* measurement hook (bench.py roofline leg): while enabled, every MFMA implicit-GEMM launch is
 * bracketed by HIP events on the stream it is launched on.  siggan_prof_read synchronises the
 * device and returns, for kernel slot idx (0..siggan_prof_slots()-1): its name, launch count,
 * summed device time (ms), summed algorithmic FLOPs and summed algorithmic HBM bytes (operands read once + result
 * written once, in the context's element type). */
int siggan_prof_enable(siggan_ctx *ctx, int32_t on);
int32_t siggan_prof_slots(void);
int siggan_prof_read(siggan_ctx *ctx, int32_t idx, char *name, int32_t name_cap, int64_t *launches,
                     double *ms, double *flops, double *bytes);

/* test hook: copy the first n elements of a library-owned workspace tensor (NHWC), converted to fp32, into out_dev:
 * "g_y"/"g_a"/"g_da" (layer 0..Lg), "d_a"/"d_dv" (block 1..Ld), "img", "dpre", "logits",
 * "probs", "dlogit".  Used by tests that localise a parity failure. */
int siggan_debug_tensor(siggan_ctx *ctx, const char *name, int32_t index, float *out_dev, int64_t n, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SIGGAN_H */
