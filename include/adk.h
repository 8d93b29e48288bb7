/*
 * adk.h — C ABI of libadk.so: the MI355X (gfx950) kernels behind the audio-diffusion sampling
 * hot path  sample(model, x, steps, eta)  ->  DiffusionAttnUnet1D.forward(x, t).
 *
 * The reference has NO FFI/plugin layer for this path (it is 100 % Python on ATen ops), so each
 * entry point below cites the reference Python call it replaces (paths relative to the reference
 * checkout).  Conventions for every entry point:
 *   - plain C: raw DEVICE pointers, ints, floats, a hipStream_t passed as void*; no torch types;
 *   - returns 0 on success, a negative ADK_E* code otherwise; adk_last_error() gives the text;
 *   - no ownership transfer, no allocation, no host sync: callers own every buffer (graph-capturable);
 *   - all launches go to the given stream; arguments are validated on the host BEFORE any launch.
 *
 * Activation layout ("chunk-plane", CPL) used by every kernel except the NCL boundary ops:
 *   element (b, c, l) of a [B, C, L] tensor with halo P lives at
 *       (((c / CB) * M) + b * Lp + P + l) * CB + (c % CB)        (in elements)
 *   with Lp = L + 2P, M = B * Lp, CB = 64 bytes / sizeof(elem) (16 for f32, 32 for bf16 / f16);
 *   C is padded up to a multiple of CB (pad channels hold zeros), halo rows hold zeros
 *   (every producer kernel writes them), so k-tap convolutions read their zero padding from memory.
 */
#ifndef ADK_H
#define ADK_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADK_OK 0
#define ADK_EINVAL (-1)   /* bad argument / unsupported shape (nothing was launched)   */
#define ADK_ELAUNCH (-2)  /* HIP launch error                                           */

#define ADK_F32 0
#define ADK_BF16 1
#define ADK_F16 2              /* IEEE half: same layout (CB = 32), kernels and MFMA rate as bf16; the dtype of the
                                  reference's own autocast region (diffusion/sampling.py:30) */

/* op kinds for adk_op / adk_run_plan */
enum {
    ADK_OP_CONV = 1,        /* nn.Conv1d k in {1,3,5,7}: blocks/blocks.py:70,73 (ResConvBlock), :68 (skip),
                               :109-110 (qkv_proj/out_proj), :95,99 (ResModConvBlock), the folded learned
                               resamplers (:214-255) and, with ADK_FLAG_CAUSAL / ELU / i0 = dilation, the
                               SoundStream convolutions (autoencoders/soundstream.py:11-115)           */
    ADK_OP_GN_APPLY = 2,    /* nn.GroupNorm(1,C) [+ nn.GELU] [+ residual add]: blocks/blocks.py:71-75,
                               :14-15 (ResidualBlock add), :107,115 (attention norm)                 */
    ADK_OP_GN_STATS = 3,    /* per-sample (sum, sumsq) partials of a CPL tensor (GroupNorm statistics) */
    ADK_OP_FIR_DOWN = 4,    /* Downsample1d.forward: blocks/blocks.py:191-196                        */
    ADK_OP_FIR_UP = 5,      /* Upsample1d.forward:   blocks/blocks.py:207-212                        */
    ADK_OP_ATTN = 6,        /* SelfAttention1d core (softmax(QK^T)V): blocks/blocks.py:116-121       */
    ADK_OP_PACK_INPUT = 7,  /* FourierFeatures + expand_to_planes + cat: blocks/blocks.py:161-167,
                               decoders/diffusion_decoder.py:144-171                                 */
    ADK_OP_UNPACK_OUTPUT = 8, /* CPL f32 -> [B,C,L] f32 (network output boundary)                    */
    ADK_OP_PQMF_ANALYSIS = 9, /* CachedPQMF.forward: diffusion/pqmf.py:271-283                       */
    ADK_OP_PQMF_SYNTHESIS = 10, /* CachedPQMF.inverse: diffusion/pqmf.py:285-302                     */
    ADK_OP_DDIM_UPDATE = 11,  /* per-step v-DDIM update: diffusion/sampling.py:34-52                 */
    ADK_OP_STEP_ADVANCE = 12, /* device-side step counter ++ (lets one hipGraph replay every step)   */
    ADK_OP_AUDIO_TO_INT16 = 13, /* demo output stage: train_uncond.py:182-187                         */
    ADK_OP_SPACE_TO_DEPTH = 14, /* phase fold in front of Downsample1d_2's strided conv: blocks/blocks.py:214-226 */
    ADK_OP_DEPTH_TO_SPACE = 15, /* phase unfold behind Upsample1d_2's transposed conv:   blocks/blocks.py:229-255 */
    ADK_OP_LINEAR = 16,       /* nn.Linear on [B, F] rows: MappingNet, mapping_cond, the AdaGN mappers
                                 (blocks/blocks.py:59,62,292-301; decoders/diffusion_decoder.py:277-278,301-304) */
    ADK_OP_FOURIER = 17       /* FourierFeatures(1, F) of c_noise = log(sigma)/4: decoders/diffusion_decoder.py:297-300 */
};

/* flags */
#define ADK_FLAG_GELU 1u        /* GN_APPLY: apply exact (erf) GELU after the affine            */
#define ADK_FLAG_OUT_F32 2u     /* CONV: output (and residual) tensors are f32 CPL even if dtype is 16-bit */
#define ADK_FLAG_NOISE 4u       /* DDIM_UPDATE: add c * noise (eta > 0)                           */
#define ADK_FLAG_ADA 8u         /* GN_APPLY: AdaGN (blocks/blocks.py:61-64): gamma/beta are per-SAMPLE rows
                                   (row stride i0 floats) and the scale is gamma + 1                */
#define ADK_FLAG_LOG 16u        /* FOURIER: embed log(sigma)/4 instead of sigma                    */
#define ADK_FLAG_NO_EMBED 32u   /* PACK_INPUT: no timestep-embedding channels ([x | cond] only)    */
#define ADK_FLAG_WT 128u         /* LINEAR: w is given transposed, [Cin][Cout] (wide layers: coalesced, read once) */
#define ADK_FLAG_CAUSAL 256u     /* CONV: CausalConv1d (autoencoders/soundstream.py:11-17): tap k reads x[t - (taps-1-k)*dil],
                                   i.e. left padding dil*(taps-1); needs halo P >= dil*(taps-1).  op.i0 = dilation
                                   (1, and 3 / 9 for taps = 7) */
#define ADK_FLAG_ELU 512u        /* CONV: nn.ELU() on the result (after bias and residual): soundstream.py:46,94-110 */
#define ADK_FLAG_GROUP_STATS 64u /* CONV: partials_out holds one (sum,sumsq) list per 32-channel group,
                                   [B][ceil(Cout/32)][adk_partials_count / groups][2], for a following
                                   AdaGN / GroupNorm(C/32, C) (blocks/blocks.py:63, 93, 97)          */

#define ADK_FLAG_FUSE_GN 1024u   /* CONV: y = [GELU](GroupNorm(1, Cout)(conv(x) + bias)) [+ residual] in ONE launch
                                   (blocks/blocks.py:70-75 and the residual add of :14-15): gamma / beta f32 [CoutPad]
                                   (zero beyond Cout), f0 = eps, ADK_FLAG_GELU selects the activation, `residual` is added
                                   AFTER norm + activation, partials_out (optional, residual form only) = statistics of y.
                                   aux1 = workspace of B * tiles-per-sample * 16 bytes, any contents, may be shared by the
                                   fused convs of a stream; aux0 = uint32 words {epoch base, error, counters[8][512]}, 16 KiB + 8,
                                   zero-initialised ONCE: adk_step_advance on the epoch base once per forward (before the
                                   first fused conv), every fused conv of the forward with its own i1 = slot in [0, 511),
                                   every one of them run exactly once per forward (tiles are handed out, one list per XCD,
                                   by counters[xcd][slot], which grow by a fixed amount per forward).  The error word becomes non-zero if a tile
                                   waited in vain for its sample's other tiles or the counters are out of step.
                                   Only where adk_conv_gn_fusable(). */

#define ADK_FLAG_SKIP_CONV 2048u /* CONV with ADK_FLAG_FUSE_GN: ResidualBlock's 1x1 skip conv (blocks/blocks.py:68, added at :14-15)
                                   folded into the launch: `residual` is then the BLOCK INPUT, CPL [B, i2, L] (i2 = its channel
                                   count, a multiple of CB), w2 = the skip's packed weights [i2/CB][1][CoutPad][CB] (no bias), and
                                   y = [GELU](GroupNorm(conv(x))) + conv1x1(residual, w2), summed in the fp32 accumulators.
                                   Pipelined kernel only (L > 128), 3 or 5 taps. */

#define ADK_FLAG_PRE_NORM 4096u  /* CONV, taps = 1: nn.GroupNorm(1, Cin) of the INPUT folded into the conv (SelfAttention1d's
                                   norm -> qkv_proj, blocks/blocks.py:107,115): w = W diag(gamma_gn) (packed), bias = b + W beta_gn,
                                   gamma = row sums of the packed w as f32 [CoutPad], partials_in / n_partials_in = (sum, sumsq)
                                   partials of x, i2 = logical channel count of x, f0 = eps:
                                   y = rstd_b * conv(x, w) + bias - mean_b * rstd_b * gamma.  Not with ADK_FLAG_FUSE_GN. */

#define ADK_FLAG_QKV 8192u       /* ATTN, 16-bit dtypes, head dim 32, L <= 256: norm -> qkv_proj -> attention core in one launch
                                   (blocks/blocks.py:113-121 without the out_proj): x is the block's INPUT, CPL [B, Cin, L];
                                   w / bias / gamma / partials_in / n_partials_in / i2 / f0 exactly as ADK_FLAG_PRE_NORM describes
                                   them for the qkv projection (CoutPad = rows of the packed w >= 3 Cin); y CPL [B, Cin, L]. */

/* One flat descriptor for every op kind; unused fields are 0/NULL.  Field use per kind is
 * documented at the matching adk_* entry point below (same names). */
typedef struct adk_op {
    int32_t kind, dtype;
    uint32_t flags;
    int32_t B, L, P;             /* batch, valid length, halo rows of the CPL layout              */
    int32_t Cin, Cout;           /* logical channel counts                                        */
    int32_t CoutPad, taps;       /* CONV: rows of the packed weight (multiple of the co-tile), k  */
    int32_t n_partials_in;       /* GN_APPLY: (sum,sumsq) pairs per sample in partials_in         */
    int32_t i0, i1, i2;          /* kind-specific ints                                            */
    float f0, f1;                /* kind-specific floats (eps, scale)                             */
    const void *x, *w, *bias, *residual;
    void *y;
    const float *gamma, *beta, *partials_in;
    float *partials_out;
    const void *aux0;
    void *aux1;
    const void *w2;              /* CONV + ADK_FLAG_SKIP_CONV: second weight tensor                 */
} adk_op;

int adk_version(void);
const char *adk_last_error(void);
int adk_sizeof_op(void);        /* sizeof(adk_op) of this build: lets a binding check its mirror of the struct */

/* Validate and launch n ops in order on `stream` (one host call per denoiser forward). */
int adk_run_plan(const adk_op *ops, int n, void *stream);

/* Host-only argument validation of a plan (exactly the checks adk_run_plan performs); launches nothing. */
int adk_validate_plan(const adk_op *ops, int n);

/* Number of (sum,sumsq) pairs per sample that an op writes to partials_out (0 if none). */
int adk_partials_count(const adk_op *op);

/* Positions per block (32, 64, 128 or 256) the conv kernel will use for this op (needs B, L, Cout, CoutPad). */
int adk_conv_postile(const adk_op *op);
/* Output channels per block (32, 128 or 256) the conv kernel will use for this op. */
int adk_conv_cotile_used(const adk_op *op);

/* 1 if the conv op (dtype, B, L, Cout, CoutPad, taps, flags filled in) can run with ADK_FLAG_FUSE_GN on the CURRENT device:
 * the pipelined 128-channel tile (L > 128, Cout > 32) or the short-level kernel, taps 3 / 5 / 7, at most 512 tiles per
 * sample - and, for the pipelined kernel, a device that deals the workgroups of a grid evenly to 8 XCDs (its per-XCD ticket
 * lists need that; a partitioned GPU does not).  The device is PROBED on the first call per device: two tiny launches and
 * a synchronising 64-byte copy on the null stream (nothing is allocated), so make the first call outside stream capture -
 * plan-building time.  A launch with ADK_FLAG_FUSE_GN on a device that was never probed probes it then (refused with
 * ADK_EINVAL inside a capture). */
int adk_conv_gn_fusable(const adk_op *op);

/* Co-tile (rows of packed weight must be a multiple of this) the conv kernel will use. */
int adk_conv_cotile(int dtype, int Cout, int L);

/* ---- single-op entry points (thin wrappers that fill an adk_op and call adk_run_plan) ---- */

/* Replaces nn.Conv1d(Cin, Cout, taps, padding=taps//2) (blocks/blocks.py:70,73; 1x1: :68,:109,:110), taps in {1,3,5,7}.
 * flags: ADK_FLAG_OUT_F32, ADK_FLAG_GROUP_STATS, ADK_FLAG_ELU (taps 1,3,7); causal / dilated taps go through adk_run_plan
 * (adk_op.i0 = dilation).
 *   x   CPL [B,Cin,L] halo P (P >= taps/2); w packed [Cin/CB][taps][CoutPad][CB] (dtype);
 *   bias f32 [CoutPad] or NULL; residual CPL [B,Cout,L] or NULL (added to the result);
 *   y CPL [B,Cout,L]; partials_out f32 [B][adk_partials_count][2] or NULL (stats of y). */
int adk_conv1d(int dtype, uint32_t flags, int B, int L, int P, int Cin, int Cout, int CoutPad, int taps,
               const void *x, const void *w, const float *bias, const void *residual, void *y,
               float *partials_out, void *stream);

/* Replaces nn.GroupNorm(1,C)(x) [-> nn.GELU()] [+ residual] (blocks/blocks.py:71-75,14-15,107).
 *   mean/var per sample come from partials_in [B][n_partials_in][2]; y may alias x. */
int adk_gn_apply(int dtype, uint32_t flags, int B, int L, int P, int C, float eps,
                 const void *x, const float *partials_in, int n_partials_in,
                 const float *gamma, const float *beta, const void *residual, void *y,
                 float *partials_out, void *stream);

/* Per-sample (sum,sumsq) partials of a CPL tensor, for inputs whose producer emitted none. */
int adk_gn_stats(int dtype, int B, int L, int P, int C, const void *x, float *partials_out, void *stream);

/* Replaces Downsample1d(kernel).forward (blocks/blocks.py:191-196): reflect pad K/2-1, K-tap FIR,
 * stride 2.  taps f32 [K] (device), K in {4,8,12}; x CPL [B,C,L] -> y CPL [B,C,L/2]. */
int adk_fir_down2(int dtype, int B, int L, int P, int C, int K, const float *taps,
                  const void *x, void *y, void *stream);

/* Replaces Upsample1d(kernel).forward (blocks/blocks.py:207-212): taps = 2*table (the module's
 * buffer), reflect pad, transposed conv stride 2.  x CPL [B,C,L] -> y CPL [B,C,2L]. */
int adk_fir_up2(int dtype, int B, int L, int P, int C, int K, const float *taps,
                const void *x, void *y, void *stream);

/* Replaces the attention core of SelfAttention1d.forward (blocks/blocks.py:116-121):
 *   qkv CPL [B,3C,L] (channel order q-heads,k-heads,v-heads; head dim 32) -> y CPL [B,C,L]. */
int adk_attn1d(int dtype, int B, int L, int P, int C, const void *qkv, void *y, void *stream);

/* The same for SelfAttentionMod1d (blocks/blocks.py:134-144; DBlock/UBlock head_size 64): head_dim 32 or 64. */
int adk_attn1d_hd(int dtype, int B, int L, int P, int C, int head_dim, const void *qkv, void *y, void *stream);

/* Replaces AdaGN.forward (blocks/blocks.py:61-64) [+ nn.GELU, flag ADK_FLAG_GELU]: F.group_norm(x, groups) then
 * bias[b] + xn * (weight[b] + 1).  weight/bias f32, one row of `row_stride` floats per sample (the two halves of the
 * mapper's output, :62).  Groups must be whole channel planes (C/groups a multiple of 16 f32 / 32 bf16, f16) or 1;
 * partials_in = adk_gn_stats of x ([B][adk_partials_count][2]).  y may alias x. */
int adk_adagn_apply(int dtype, uint32_t flags, int B, int L, int P, int C, int groups, float eps,
                    const void *x, const float *partials_in, int n_partials_in,
                    const float *weight, const float *bias, int row_stride, void *y, void *stream);

/* Replaces nn.Linear on per-sample rows (MappingNet blocks/blocks.py:292-301, mapping_cond and the AdaGN mappers
 * blocks/blocks.py:59,62): y[b][o] = act(sum_i w[o][i] x[b][i] + bias[o] + residual[b][o]), all f32, w row-major
 * [Cout][Cin] (or [Cin][Cout] with ADK_FLAG_WT, the fast form for wide layers); bias/residual may be NULL;
 * act = exact GELU with ADK_FLAG_GELU; strides in floats. */
int adk_linear(uint32_t flags, int B, int Cin, int Cout, const float *x, int x_stride, const float *w, const float *bias,
               const float *residual, int res_stride, float *y, int y_stride, void *stream);

/* Replaces timestep_embed(append_dims(c_noise, 2)) of AudioDenoiserModel.forward (decoders/diffusion_decoder.py:297-300):
 * y[b] = [cos(2 pi s w) | sin(2 pi s w)], s = log(sigma[b]) / 4 with ADK_FLAG_LOG else sigma[b]; w f32 [F/2]. */
int adk_fourier_features(uint32_t flags, int B, int F, const float *sigma, const float *w, float *y, int y_stride,
                         void *stream);

/* Input boundary of AudioDenoiserModel.forward (decoders/diffusion_decoder.py:306-308) with the Karras input scaling of
 * Denoiser.forward (:336-337): y CPL [B, roundup(Cx + Cc, 32), L] = [x * x_scale[b] | interpolate(cond)], x_scale and
 * cond optional (NULL). */
int adk_pack_input_plain(int dtype, int B, int L, int P, int Cx, const float *x, const float *x_scale,
                         const float *cond, int Cc, int Lc, void *y, void *stream);

/* Output boundary with the Karras skip of Denoiser.forward (decoders/diffusion_decoder.py:337):
 * y = unpack(x) * c_out[b] + x_in * c_skip[b]   (x_in = the UNSCALED network input, f32 NCL). */
int adk_unpack_output_scaled(int B, int L, int P, int C, const float *x, const float *c_out, const float *c_skip,
                             const float *x_in, float *y, void *stream);

/* Replaces FourierFeatures + expand_to_planes + torch.cat (blocks/blocks.py:161-167,
 * decoders/diffusion_decoder.py:144-171):  x f32 [B,Cx,L] (NCL), t: f32 [B] if t_table==0, else
 * t = ((const float*)t)[*step * stride] broadcast (graph replay); fw f32 [8] (timestep_embed.weight);
 * y CPL [B, roundup(Cx+16,32), L] = [x | cos(2*pi*t*fw) | sin(2*pi*t*fw) | 0...]. */
int adk_pack_input(int dtype, int B, int L, int P, int Cx, const float *x, const float *t,
                   const int32_t *step, int t_stride, const float *fw, void *y, void *stream);

/* The same with the conditioning inputs of DiffusionAttnUnet1D.forward (decoders/diffusion_decoder.py:148-169):
 *   cond f32 [B,Cc,Lc] is resampled to L like F.interpolate(mode='linear', align_corners=False) (:150-151);
 *   aug_level f32 [B] or NULL: with cond_noise_aug a second Fourier embedding of the noise level goes in front
 *   of cond (:162-165) and cond += aug_noise * aug_level[b] (:160), aug_noise f32 [B,Cc,L] = the caller's N(0,1)
 *   draw (or NULL: no noise added).
 *   y CPL [B, roundup(Cx + 16 (+16) + Cc, 32), L] = [x | emb(t) | (emb(aug)) | cond | 0...]. */
int adk_pack_input_cond(int dtype, int B, int L, int P, int Cx, const float *x, const float *t,
                        const int32_t *step, int t_stride, const float *fw,
                        const float *cond, int Cc, int Lc, const float *aug_level, const float *aug_noise,
                        void *y, void *stream);

/* CPL f32 [B,C,L] -> NCL f32 [B,C,L] (output boundary of the network). */
int adk_unpack_output(int B, int L, int P, int C, const float *x, float *y, void *stream);

/* Replaces CachedPQMF.forward (diffusion/pqmf.py:271-283).  x f32 [B,Ch,T]; fwT f32 [K][M] = the
 * forward_conv weight transposed (K odd, zero padding K/2 each side, stride M); y f32 [B,Ch*M,T/M]. */
int adk_pqmf_analysis(int B, int Ch, int T, int M, int K, const float *x, const float *fwT, float *y,
                      void *stream);

/* Replaces CachedPQMF.inverse (diffusion/pqmf.py:285-302).  x f32 [B,Ch*M,N]; iwT f32 [M(c)][Kt][M(m)]
 * = inverse_conv weight permuted (m fastest); y f32 [B,Ch,N*M]. */
int adk_pqmf_synthesis(int B, int Ch, int N, int M, int Kt, const float *x, const float *iwT, float *y,
                       void *stream);

/* Replaces the body of the sampling loop after the model call (diffusion/sampling.py:34-52), literal
 * operation order, for step i = *step:  coef f32 [steps][8] = {t, alpha, sigma, alpha_next, adjusted_sigma,
 * ddim_sigma, is_last, 0}.  pred = x*alpha - v*sigma; eps = x*sigma + v*alpha;
 * if !is_last: x = pred*alpha_next + eps*adjusted (+ noise*ddim_sigma).  x updated in place. */
int adk_ddim_update(uint32_t flags, int64_t n, float *x, const float *v, const float *noise, float *pred,
                    const float *coef, const int32_t *step, void *stream);

int adk_step_advance(int32_t *step, void *stream);

/* Learned resamplers (blocks/blocks.py:214-255).  Downsample1d_2 = Conv1d(k = 2f+1, stride f, padding f) and
 * Upsample1d_2 = ConvTranspose1d(k = 2f, stride f) are each ONE 3-tap stride-1 adk_conv1d on phase-folded data
 * (weights re-indexed on the host: audio_diffusion_amd/unet.py), with these row shuffles around it:
 *   space_to_depth: x CPL [B,C,L]     -> y CPL [B,f*C,L/f],  y[p*C + c][m] = x[c][m*f + p]
 *   depth_to_space: x CPL [B,f*C,L]   -> y CPL [B,C,L*f],    y[c][j*f + s] = x[s*C + c][j]
 * C must be a multiple of the channel chunk (16 f32 / 32 bf16, f16); L is the INPUT length. */
int adk_space_to_depth(int dtype, int B, int L, int P, int C, int factor, const void *x, void *y, void *stream);
int adk_depth_to_space(int dtype, int B, int L, int P, int C, int factor, const void *x, void *y, void *stream);

/* Replaces the output stage every caller of sample() applies (train_uncond.py:182-187, generate.py):
 *   rearrange(fakes, 'b d n -> d (b n)').clamp(-1, 1).mul(32767).to(torch.int16)
 * x f32 [B, C, T] -> y int16 [C, B*T] (truncation toward zero, like torch's float -> int16 cast). */
int adk_audio_to_int16(int B, int C, int T, const float *x, int16_t *y, void *stream);

/* nn.GroupNorm(1, C)(x) [-> nn.GELU() with ADK_FLAG_GELU] [+ residual] in ONE call that allocates nothing and takes no
 * workspace: the variant of adk_gn_stats + adk_gn_apply for callers that cannot call hipMalloc (code under HIP graph
 * capture, tight serving loops, harnesses that forbid allocation).  It has NO counterpart in the reference (which
 * allocates freely through PyTorch).  The (sum, sumsq) partials that adk_gn_apply takes as the workspace argument
 * `partials_in` live in a fixed 4 MiB array inside the library (one copy per GPU, held until the process ends):
 * B * adk_partials_count(GN op) must be <= 524288 pairs, else ADK_EINVAL (e.g. B=16, C=128, L=65536 needs 4096).
 * Output bits equal adk_gn_stats + adk_gn_apply.  Calls that overlap in time on ONE GPU (two streams) share that
 * array and race; calls on one stream, or on different GPUs, do not.  The stream's device must be current. */
int adk_group_norm_noalloc(int dtype, uint32_t flags, int B, int L, int P, int C, float eps,
                           const void *x, const float *gamma, const float *beta, const void *residual, void *y,
                           void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ADK_H */
