/*
 * siggan_mlp.h -- C ABI of the fully-connected ("MLP") vanilla-GAN variant of the MI355X engine.
 *
 * BUILD-DEFINED EXTENSION, PARITY UNPINNED.  BASELINE.json's configs[0] ("z=100 -> 28x28x1 MLP G (100-256-512-784) / MLP D,
 * bs=32") and the wording of configs[1] name a fully-connected generator / discriminator; the reference contains none
 * (its "vanilla" model is the conv G/D of generator_vanilla_gan.py:124-163 / discriminator_vanilla_gan.py:131-207, sizes
 * other than 64 / 128 are rejected at :106-107 / :121-122; the only trace of an MLP is the unread `hidden_layers`
 * field of model_versions.yaml:15).  There is therefore no reference interface this header replaces and no reference
 * output to compare with: the checker is the build's own CPU restatement (oracle/mlp_oracle.py), and every result of this
 * path is labelled "parity unpinned".  The model follows the conv path's conventions where they apply:
 *
 *   G: z (latent) -> [Linear(h_i) -> BatchNorm1d -> ReLU] for each hidden width -> Linear(S*S) -> Tanh -> (B,1,S,S)
 *   D: (B,1,S,S) -> [Linear(h_i) -> LeakyReLU(slope)] for the hidden widths reversed -> Linear(1) -> Sigmoid
 *   BCELoss on probabilities (log clamp -100), label smoothing on the real labels of the D step, Adam(lr, betas), the
 *   trainer's step order (D step with G in eval mode under no-grad; G step with BatchNorm batch statistics), n_critic = 1.
 *
 * Every dense product (forward, input-gradient, weight-gradient) runs on v_mfma_f32_32x32x2_f32; BatchNorm / bias /
 * activation / BCE / Adam reuse the conv engine's kernels.  Same conventions as siggan.h: plain pointers and sizes,
 * caller-owned flat fp32 arenas in parameters() order (borrowed by mlpgan_bind), every call enqueues on `stream`,
 * 0 = OK / negative = SIGGAN_E_* with the message in siggan_last_error().
 */
#ifndef SIGGAN_MLP_H
#define SIGGAN_MLP_H

#include "siggan.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MLPGAN_MAX_HIDDEN 4

typedef struct mlpgan_ctx mlpgan_ctx;

typedef struct mlpgan_config {
    int32_t device;
    int32_t latent_dim;                 /* 100 */
    int32_t image_size;                 /* S: images are (B,1,S,S); 28 (configs[0]) or 64 */
    int32_t n_hidden;                   /* 1..MLPGAN_MAX_HIDDEN */
    int32_t hidden[MLPGAN_MAX_HIDDEN];  /* Generator widths in order (256, 512); the Discriminator uses them reversed */
    int32_t max_batch;
    float   leaky_slope;                /* Discriminator LeakyReLU slope (0.2) */
    uint64_t seed;
} mlpgan_config;

/* parameters() order.  G: per hidden layer i: weight (h_i, in), bias (h_i), bn.weight (h_i), bn.bias (h_i); then the output
 * layer weight (S*S, h_last), bias (S*S).  D: per layer weight (out, in), bias (out), the last being (1, h_0).  BatchNorm
 * running statistics: all hidden layers concatenated; one int64 counter per BatchNorm layer. */
typedef struct mlpgan_storage {
    float *g_params, *g_grads, *g_exp_avg, *g_exp_avg_sq, *g_adam_steps;
    float *g_bn_running_mean, *g_bn_running_var;
    int64_t *g_bn_batches;
    float *d_params, *d_grads, *d_exp_avg, *d_exp_avg_sq, *d_adam_steps;
} mlpgan_storage;

int mlpgan_create(const mlpgan_config *cfg, mlpgan_ctx **out);
int mlpgan_destroy(mlpgan_ctx *ctx);
int64_t mlpgan_param_count(const mlpgan_ctx *ctx, int which);     /* 0 = generator, 1 = discriminator */
int32_t mlpgan_param_tensors(const mlpgan_ctx *ctx, int which);
int64_t mlpgan_bn_count(const mlpgan_ctx *ctx);
int mlpgan_bind(mlpgan_ctx *ctx, const mlpgan_storage *st);
int mlpgan_seed(mlpgan_ctx *ctx, uint64_t seed, uint64_t offset);

/* z_dev (B,latent) -> images_dev (B,1,S,S); training != 0: BatchNorm batch statistics (+ running update) */
int mlpgan_g_forward(mlpgan_ctx *ctx, const float *z_dev, int32_t batch, int32_t training, float *images_dev, void *stream);
/* x_dev (B,1,S,S) -> probs_dev (B) */
int mlpgan_d_forward(mlpgan_ctx *ctx, const float *x_dev, int32_t batch, float *probs_dev, void *stream);
/* the two training steps (metrics: the SIGGAN_M_* slots of siggan.h); z_dev NULL: drawn by the library RNG */
int mlpgan_d_step(mlpgan_ctx *ctx, const float *real_dev, int32_t batch, const float *z_dev, const siggan_hyper *hp,
                  float *metrics_dev, void *stream);
int mlpgan_g_step(mlpgan_ctx *ctx, int32_t batch, const float *z_dev, const siggan_hyper *hp, float *metrics_dev, void *stream);

/* operator entry (tests): C[M][N] = A . op(B) on the fp32 matrix cores.  layout 0 "NT": A (M,K), B (N,K);  1 "NN": A (M,K),
 * B (K,N);  2 "TN": A (K,M), B (K,N).  All row-major fp32. */
int mlpgan_op_gemm(int32_t device, int32_t layout, const float *a_dev, const float *b_dev, float *c_dev, int32_t m, int32_t n,
                   int32_t k, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SIGGAN_MLP_H */
