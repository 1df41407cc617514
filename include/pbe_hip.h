/*
 * libpbe_hip.so — C-ABI of the MI355X (gfx950) native kernels behind Paint-by-Example's
 * PLMS denoising hot path.
 *
 * The reference (zhanwenchen/pbe) is 100 % Python on PyTorch and has NO native layer or FFI
 * (SURVEY.md "Quick facts"); every entry point below therefore replaces an ATen dispatch made
 * from a reference Python function, cited per entry as file:line under /root/reference.
 * The host side (this repo's `ldm/` package, same import paths / class names / state_dict keys
 * as the reference) binds these symbols with ctypes (pbe_amd/lib.py); INTEGRATION.md shows the
 * stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 on success or a negative PBE_E* code; pbe_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *   - the CALLER allocates every buffer (device memory); the library never allocates, frees or
 *     retains a pointer.  Kernels are enqueued on `stream` (a hipStream_t) and never synchronise,
 *     so every call is capturable in a hipGraph.
 *   - activations are fp16, NHWC ("tokens x channels") unless an entry says otherwise; GEMM and
 *     attention accumulate in fp32 on the matrix cores; norms / softmax reduce in fp32.
 *   - descriptors are plain C structs of pointers and sizes (no torch types).
 */
#ifndef PBE_HIP_H
#define PBE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PBE_ABI_VERSION 8

#define PBE_OK 0
#define PBE_EINVAL (-1)  /* bad shape / alignment / null pointer          */
#define PBE_ELAUNCH (-2) /* hipLaunch failed                              */
#define PBE_ENOTSUP (-3) /* shape outside what the kernels are built for  */

/* epilogue activations */
#define PBE_ACT_NONE 0
#define PBE_ACT_SILU 1
#define PBE_ACT_GELU_ERF 2
#define PBE_ACT_QUICK_GELU 3
#define PBE_ACT_GEGLU 4 /* GEMM only: W rows interleaved (x_j, gate_j); C[m, j] = x_j * gelu_erf(gate_j), width N/2 (attention.py:43-45) */

#define PBE_DTYPE_F16 0
#define PBE_DTYPE_F8E4M3 1 /* OCP e4m3fn (gfx950's fp8; not the fnuz form of gfx942) */

typedef void* pbe_stream_t; /* hipStream_t */

int pbe_abi_version(void);
const char* pbe_last_error(void);
/* sha256 (first 16 hex digits) over the sources and flags the library was built from (pbe_amd/build.py); the ctypes loader
 * recomputes it from the tree and refuses a stale binary instead of silently running old kernels. */
const char* pbe_source_hash(void);
/* sizeof() of the three descriptor structs as THIS binary was compiled: a binding checks its own struct layout against them
 * at load time (pbe_amd/lib.py does; the ctypes stub of INTEGRATION.md section 2 does), so a binding written for an older
 * ABI hands over a short struct and gets an error instead of an out-of-bounds read. */
size_t pbe_sizeof_gemm_desc(void);
size_t pbe_sizeof_conv3x3_desc(void);
size_t pbe_sizeof_attn_desc(void);

/* ---------------------------------------------------------------------------------------------
 * pbe_gemm_f16 — C[m,n] = act(alpha * sum_k A[m,k] * W[n,k] + bias + rowvec[m / group_rows, n]) + R[m,n]
 * Replaces torch.nn.Linear / 1x1 Conv2d / einsum dispatches:
 *   ldm/modules/attention.py:198-205,41,61,270-285 (to_q/k/v, to_out, GEGLU proj, FF out, proj_in/out),
 *   ldm/modules/diffusionmodules/openaimodel.py:218-224,234-241,623-628 (emb_layers, skip 1x1, time_embed),
 *   ldm/modules/diffusionmodules/model.py:152-204 (VAE q/k/v/proj_out and both bmm's),
 *   ldm/modules/encoders/xf.py:30-57, transformers CLIP q/k/v/out_proj/fc1/fc2, latent_diffusion.py:112.
 * A may be split along K into two sources (A | A2 at K1) so torch.cat((h, skip), 1)
 * (openaimodel.py:883) is never materialised.
 * ------------------------------------------------------------------------------------------ */
typedef struct pbe_gemm_desc {
    const void* A;      /* fp16 [M, K1] row-major, leading dim lda                    */
    const void* A2;     /* fp16 [M, K-K1] or NULL                                     */
    const void* W;      /* fp16 [N, K] row-major (torch Linear layout), leading dim ldw */
    void* C;            /* fp16 [M, N], leading dim ldc                               */
    const float* bias;  /* fp32 [N] (or [M] when bias_per_row) or NULL                */
    const void* rowvec; /* fp16 [ceil(M/group_rows), ldv] broadcast over row groups, or NULL */
    const void* resid;  /* fp16 [M, N] leading dim ldr, added AFTER act, or NULL      */
    int32_t M, N, K, K1;
    int64_t lda, lda2, ldw, ldc, ldr;
    int32_t ldv, group_rows;
    int64_t strideA, strideW, strideC, strideR; /* batch strides (elements)           */
    int32_t batch;
    float alpha;
    int32_t act;
    int32_t bias_per_row;
    void* workspace;        /* optional device scratch for split-K partial sums (fp32), or NULL     */
    size_t workspace_bytes; /* any size: the split is clamped to what fits (64 MiB covers the path) */
    int32_t tile_cfg;       /* -1 = built-in heuristic; else (block-tile config 0..18; 10..14 are the halo-resident conv tiles and apply to stride-1 3x3 convs only, 15..18 deep-ring forms of 3 / 4 / 6 / 9) | (split-K factor << 8), factor 0 = library's choice */
    /* fp8 operands (BASELINE configs[4]): operand_dtype = PBE_DTYPE_F8E4M3 -> A [M, K] and W [N, K] hold OCP e4m3 bytes, lda / ldw /
     * strideA / strideW count BYTES (multiples of 16, K % 16 == 0, no A2), and C = act(alpha * a_scale[m] * w_scale[n] * sum_k A W + ...):
     * a_scale fp32 [M] per row of A (e.g. per token, from pbe_layernorm_f8), w_scale fp32 [N] per row of W (per output channel, from the
     * pack); *_scale_stride = elements between the batches' scale vectors (0: shared).  C, bias, rowvec, resid stay as in the fp16 form. */
    const float* a_scale;
    const float* w_scale;
    int64_t a_scale_stride, w_scale_stride;
    int32_t operand_dtype;  /* PBE_DTYPE_F16 (0) or PBE_DTYPE_F8E4M3 (1) */
    /* Extended epilogue (ABI 7; fp16 operands, batch 1, never split-K) - the transformer block's GEMM chain, attention.py:198-252:
     *  alpha_cols > 0: alpha multiplies columns n < alpha_cols only (q of a fused q | k | v projection, pre-scaled for pbe_attention_f16);
     *  LayerNorm FOLDED into this GEMM (attention.py:248-252 norm1 / norm3 -> to_q/k/v, ff.net[0].proj): A holds the raw rows x, W must be
     *    W * gamma, bias must be W beta (+ the layer's bias), ln_colsum[n] = sum_k (W gamma)[n, k] of the fp16 values, and the epilogue forms
     *    rstd[m] * (acc - mean[m] * ln_colsum[n]) from the row statistics: (sum, sum of squares) of row m = sum over p < ln_parts of the
     *    float2 ln_stats[(p * ln_stats_ld + m)] (written by the PRODUCER of x through row_stats_out, or by pbe_row_stats_f16);
     *  row_stats_out: float2 [column tiles][row_stats_ld] partial (sum, sumsq) of THIS launch's stored fp16 output rows, one partial per column tile
     *    (pbe_gemm_plan's out6[5] tells how many), for the LayerNorm that reads the output;
     *  VT: columns n >= vt_col0 go to VT[b * vt_bs + (n - vt_col0) * vt_rs + tok] instead of C (row m = b * vt_tokens + tok): V^T for
     *    pbe_attention_f16 out of the same launch as q | k.  vt_col0 must be a multiple of the tile width the plan picks. */
    int32_t alpha_cols;
    const float* ln_stats;
    int32_t ln_parts;
    int64_t ln_stats_ld;
    const float* ln_colsum;
    float ln_eps;
    float* row_stats_out;
    int64_t row_stats_ld;   /* rows between two column tiles' partials in row_stats_out (>= M; 0 = M) */
    void* VT;
    int32_t vt_col0, vt_tokens;
    int64_t vt_bs, vt_rs;
} pbe_gemm_desc;
int pbe_gemm_f16(const pbe_gemm_desc* d, pbe_stream_t stream);
/* Plan / workspace query for the SAME descriptor (nothing is launched): out6 = {tile config index, split-K factor,
 * tile rows BM, tile columns BN, workgroups, column tiles}; *workspace_needed = bytes of split-K scratch the plan uses (0 when the
 * plan does not split).  A descriptor with a smaller workspace gets a smaller factor, never an error. */
int pbe_gemm_plan(const pbe_gemm_desc* d, int32_t* out6, size_t* workspace_needed);

/* ---------------------------------------------------------------------------------------------
 * pbe_conv3x3_f16 — NHWC 3x3 convolution as an implicit GEMM on the matrix cores.
 *   Y[b,oy,ox,co] = act(sum_{dy,dx,ci} X[b, iy, ix, ci] * Wp[co, k(dy*3+dx, ci)] + bias[co]
 *                       + rowvec[b, co]) + R[b,oy,ox,co]        (k(tap, ci): see kblock below)
 *   iy = oy*stride + dy - pad (zero outside), optional nearest-2x upsample of X fused in the gather,
 *   X optionally the channel concat of two tensors (X | X2).
 * Replaces Conv2d(k=3) dispatches in openaimodel.py:216,229-231 (ResBlock), :109-119 (Upsample:
 * F.interpolate + conv), :150-160 (Downsample s2 p1), :658-662,824-828 (conv in/out);
 * model.py:44-81,92-121 (VAE convs; Downsample pad (0,1,0,1) + s2 p0 == pad=0 here).
 * Requires C1 % 64 == 0 and C2 % 64 == 0; small-Cin convs go through pbe_im2col3x3_f16 + GEMM.
 * Wp is the OIHW weight re-packed by the host to [Cout, 9*Cin] in (channel block, tap, channel) order — see kblock.
 * ------------------------------------------------------------------------------------------ */
typedef struct pbe_conv3x3_desc {
    const void* X;
    const void* X2;
    const void* Wp;
    void* Y;
    const float* bias;
    const void* rowvec; /* fp16 [B, ldv]: per-(sample, channel) add (ResBlock emb, openaimodel.py:273) */
    const void* resid;  /* fp16 [B,Ho,Wo,Cout] */
    int32_t B, H, W, C1, C2, Cout;
    int32_t stride, pad, upsample; /* upsample: 0; 1 = nearest-2x fused in the gather (Wp as usual, 9 taps); 2 = the same operation in PHASE form: Wp holds
                                      four weight sets [4][Cout][4*Cin] (phase (py, px) = output pixel parity; per phase the 3x3 taps that fall on one
                                      source pixel are summed: pbe_amd.ops.pack_conv3x3_up_phases), four 2x2 convs on the source grid, 4/9 of the MACs;
                                      pad 1, stride 1, no X2 / rowvec / resid */
    int32_t ldv;
    int32_t act;
    void* workspace;        /* optional split-K scratch, as in pbe_gemm_desc */
    size_t workspace_bytes;
    int32_t tile_cfg;       /* -1 = heuristic, else tile config | (split-K factor << 8), as in pbe_gemm_desc */
    int32_t kblock;         /* channel block cb of Wp's K order: k = ((ci/cb)*9 + tap)*cb + ci%cb; multiple of 64 that
                               divides C1 and C2 (0 = 64).  A pixel's 9 taps are then re-read within 9*cb/64 k-tiles (L2 hits) */
    /* Y feeds a GroupNorm (ResBlock: conv -> GroupNorm32 -> SiLU, openaimodel.py:213-227; the block's output -> the next normalisation):
     * group_stats_out != NULL asks the conv's copy-out for the per-(sample, row block, group) partial (sum, sumsq) of the STORED fp16
     * values, in pbe_groupnorm_f16's partial layout [B][blocks][group_stats_groups][2] (fp32; size it for blocks <= Ho*Wo / 64).
     * *group_stats_blocks (host int, written before the call returns) = blocks per sample actually produced, or 0 when the planned tile
     * cannot (split-K, a tile that is not whole groups of one sample, multi-pass epilogue): the caller then runs pbe_groupnorm_f16,
     * else pbe_groupnorm_apply_f16 with these partials.  Fixed summation order: bit-reproducible run to run. */
    float* group_stats_out;
    int32_t group_stats_groups;
    int32_t* group_stats_blocks;
} pbe_conv3x3_desc;
int pbe_conv3x3_f16(const pbe_conv3x3_desc* d, pbe_stream_t stream);
int pbe_conv3x3_plan(const pbe_conv3x3_desc* d, int32_t* out6, size_t* workspace_needed); /* as pbe_gemm_plan */

/* im2col for the three small-Cin convs (9->320 U-Net in, 3->128 VAE in, 4->512 VAE decoder in):
 * X fp16 NHWC [B,H,W,Cp] -> out fp16 [B*Ho*Wo, 9*Cp]. */
int pbe_im2col3x3_f16(const void* X, void* out, int32_t B, int32_t H, int32_t W, int32_t Cp,
                      int32_t stride, int32_t pad, pbe_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * pbe_groupnorm_f16 — GroupNorm(groups) [+ SiLU] over NHWC fp16, fp32 statistics.
 * Replaces GroupNorm32 (util.py:214-216, eps 1e-5) + SiLU in openaimodel.py:213-215,225-227,824-826;
 * Normalize (attention.py:77-78 / model.py:40-41, eps 1e-6) + swish (model.py:35-37).
 * Input may be the channel concat X | X2 (C2 = 0 for a single source).  Output [B, HW, C1+C2].
 * workspace: pbe_groupnorm_workspace_bytes(B, HW) bytes of device scratch.
 * ------------------------------------------------------------------------------------------ */
size_t pbe_groupnorm_workspace_bytes(int32_t B, int32_t HW);
int pbe_groupnorm_f16(const void* X, const void* X2, const float* gamma, const float* beta, void* Y,
                      int32_t B, int32_t HW, int32_t C1, int32_t C2, int32_t groups, float eps,
                      int32_t silu, void* workspace, size_t workspace_bytes, pbe_stream_t stream);
/* The normalisation pass alone, statistics given as partials [B][blocks][groups][2] (sum, sumsq) - from pbe_conv3x3_f16's
 * group_stats_out.  Single source (no concat), any map size. */
int pbe_groupnorm_apply_f16(const void* X, const float* partials, int32_t blocks, const float* gamma, const float* beta, void* Y,
                            int32_t B, int32_t HW, int32_t C, int32_t groups, float eps, int32_t silu, pbe_stream_t stream);

/* pbe_layernorm_f16 — LayerNorm over the last dim C (C % 8 == 0, C <= 2048) of fp16 rows.
 * Replaces attention.py:240-242 (norm1/3), xf.py:22-28, HF CLIP layer norms. */
int pbe_layernorm_f16(const void* X, const float* gamma, const float* beta, void* Y, int64_t rows,
                      int32_t C, int64_t ldx, int64_t ldy, float eps, pbe_stream_t stream);

/* pbe_row_stats_f16 — out[r] = float2(sum, sum of squares) of the fp16 row X[r, :C]: the one-partial form of the row statistics a
 * LayerNorm-folding GEMM reads (pbe_gemm_desc.ln_stats, ln_parts = 1) when the producer of X did not emit them (row_stats_out). */
int pbe_row_stats_f16(const void* X, float* out, int64_t rows, int32_t C, int64_t ldx, pbe_stream_t stream);

/* pbe_layernorm_f8 — the same LayerNorm emitting OCP e4m3 bytes and one fp32 scale per row (BASELINE configs[4]):
 * Y[r, :] = e4m3(LN(X[r, :]) / row_scale[r]), row_scale[r] = max|LN(X[r, :])| / 448; ldy in bytes (multiple of 16).  Feeds the fp8
 * operand form of pbe_gemm_f16 (A = Y, a_scale = row_scale). */
int pbe_layernorm_f8(const void* X, const float* gamma, const float* beta, void* Y, float* row_scale, int64_t rows,
                     int32_t C, int64_t ldx, int64_t ldy, float eps, pbe_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * pbe_attention_f16 — fused softmax(Q K^T * scale) V (flash-style, no [N,N] tensor in HBM).
 * Replaces attention.py:214-229 (einsum, softmax, einsum) and HF CLIP eager attention.
 *   Q: fp16, element (b, n, h, d) at Q[b*q_bs + n*q_rs + h*D + d]   (same for K with k_*),
 *   VT: V transposed, element (b, h, d, n) at VT[b*vt_bs + (h*D+d)*vt_rs + n]  (vt_rs % 8 == 0,
 *       the row must be readable up to the next multiple of 8 past Nk),
 *   O: fp16, element (b, n, h, d) at O[b*o_bs + n*o_rs + h*D + d].
 * D % 8 == 0, D <= 160.  The softmax reference maximum is deferred (raised when a tile exceeds it by 2^8); the d = 40 form keeps it
 * as an fp16 number / 64, i.e. |scale * log2(e) * q.k| must stay below 4e6.
 * ------------------------------------------------------------------------------------------ */
typedef struct pbe_attn_desc {
    const void* Q;
    const void* K;
    const void* VT;
    void* O;
    int32_t B, H, Nq, Nk, D;
    int64_t q_bs, q_rs, k_bs, k_rs, vt_bs, vt_rs, o_bs, o_rs;
    float scale;
    int32_t q_prescaled; /* 1: Q already holds scale * log2(e) * q (the projection GEMM applied it in its fp32 epilogue, pbe_gemm_desc.alpha_cols);
                            `scale` is then ignored */
} pbe_attn_desc;
int pbe_attention_f16(const pbe_attn_desc* d, pbe_stream_t stream);

/* pbe_softmax_rows_f16 — Y[r,:] = softmax(scale * X[r,:]) over rows of `cols` fp16 (VAE mid attention,
 * model.py:193-195: one head, d = 512, N = 4096, scores kept in HBM once per image). */
int pbe_softmax_rows_f16(const void* X, void* Y, int64_t rows, int32_t cols, int64_t ldx, int64_t ldy,
                         float scale, pbe_stream_t stream);

/* pbe_geglu_f16 — Y[m, f] = H[m, f] * gelu_erf(H[m, F + f])  (attention.py:43-45). */
int pbe_geglu_f16(const void* H, void* Y, int64_t M, int32_t F, pbe_stream_t stream);

/* pbe_timestep_embedding_f16 — util.py:151-171: out[b, :] = cat(cos(t f_k), sin(t f_k)), fp16 [B, dim]. */
int pbe_timestep_embedding_f16(const int64_t* t, void* out, int32_t B, int32_t dim, float max_period,
                               pbe_stream_t stream);

/* Layout conversion at the API boundary (the reference API is NCHW fp32/fp16):
 * src fp32 NCHW [B,C,HW] -> dst fp16 NHWC [B,HW,Cp] (channels C..Cp-1 zero) and back. */
int pbe_nchw_f32_to_nhwc_f16(const float* src, void* dst, int32_t B, int32_t C, int32_t HW, int32_t Cp,
                             pbe_stream_t stream);
int pbe_nhwc_f16_to_nchw_f32(const void* src, float* dst, int32_t B, int32_t C, int32_t HW, int32_t ld,
                             pbe_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * PLMS sampler element-wise steps (plms.py:177-248), state kept fp32 NCHW [B,4,HW].
 * pbe_plms_pack_input: x9 = cat(x, z_inpaint, mask) (plms.py:225) duplicated `dup` times along the
 *   batch for classifier-free guidance (plms.py:185), as fp16 NHWC [dup*B, HW, 16] (9 real channels).
 * pbe_plms_update: eps_out fp16 NHWC [dup*B, HW, ld] from the U-Net ->
 *   e_t = e_u + scale (e_c - e_u)                       (plms.py:188-189; dup == 1: e_t = eps_out)
 *   e'  = c0 e_t + c1 h1 + c2 h2 + c3 h3                (plms.py:230-244; Adams-Bashforth weights)
 *   pred_x0 = (x - sqrt(1-a_t) e') / sqrt(a_t);  x_prev = sqrt(a_prev) pred_x0 + sqrt(1-a_prev) e'
 *   coef = {c0,c1,c2,c3, sqrt_one_minus_at, 1/sqrt(a_t), sqrt(a_prev), sqrt(1-a_prev)} (host floats).
 * ------------------------------------------------------------------------------------------ */
int pbe_plms_pack_input(const float* x, const float* z_inpaint, const float* mask, void* x9,
                        int32_t B, int32_t HW, int32_t dup, pbe_stream_t stream);
int pbe_plms_update(const void* eps_out, int32_t ld, int32_t dup, float cfg_scale, const float* x,
                    const float* h1, const float* h2, const float* h3, const float* coef8,
                    float* e_t, float* x_prev, float* pred_x0, int32_t B, int32_t HW,
                    pbe_stream_t stream);

/* Stochastic sampler options reachable from the reference CLI (scripts/inference.py:164,342 --ddim_eta; plms.py:150-153 / ddim.py:178-181):
 * pbe_axpy_f32          y += a * x : the sigma_t * noise * temperature term of a DDIM step with eta > 0 (ddim.py:236-238);
 * pbe_qsample_blend_f32 out = (sqrt_ac x0 + sqrt_1m_ac noise) * mask + (1 - mask) * img : img_orig = q_sample(x0, ts) blended under `mask`
 *                       (fp32 NCHW [B,C,HW]; mask [B,1,HW] or [B,C,HW]).  The noise tensors are the caller's (the reference draws them
 *                       from the device RNG: not reproducible across devices, so parity uses injected noise). */
int pbe_axpy_f32(float* y, float a, const float* x, int64_t n, pbe_stream_t stream);
int pbe_qsample_blend_f32(const float* x0, const float* noise, const float* mask, const float* img, float sqrt_ac, float sqrt_1m_ac,
                          float* out, int32_t B, int32_t C, int32_t HW, int32_t mask_channels, pbe_stream_t stream);

/* pbe_posterior_sample — distributions.py:25-37 + latent_diffusion.py:262:
 * moments fp16 NHWC [B,HW,ld] (mean 0..3 | logvar 4..7), eps fp32 NCHW [B,4,HW] ->
 * z fp32 NCHW = scale * (mean + exp(0.5 clamp(logvar,-30,20)) * eps). */
int pbe_posterior_sample(const void* moments, int32_t ld, const float* eps, float* z, int32_t B,
                         int32_t HW, float scale, pbe_stream_t stream);

/* pbe_scale_latent_f16 — decode_first_stage prologue (latent_diffusion.py:454,506): fp32 NCHW
 * [B,C>=4,HW] -> fp16 NHWC [B,HW,8] of (1/scale_factor) * z[:, :4]. */
int pbe_scale_latent_f16(const float* z, void* out, int32_t B, int32_t C, int32_t HW, float inv_scale,
                         pbe_stream_t stream);

/* pbe_clip_patchify_f16 — HF CLIPVisionEmbeddings patch conv (14x14 s14, no bias) as im2col:
 * pixels fp32 NCHW [B,3,S,S] -> fp16 [B*(S/P)^2, Kp], k = c*P*P + ky*P + kx, zero padded to Kp. */
int pbe_clip_patchify_f16(const float* pixels, void* out, int32_t B, int32_t S, int32_t P, int32_t Kp,
                          pbe_stream_t stream);

/* pbe_add_rows_f16 — Y[b, r, :] = X[r, :] + V[:]  for r in [0, rows): class-token row of the CLIP
 * embedding (class_embedding + position_embedding[0]) written to tokens[b, 0, :]. */
int pbe_bcast_row_f16(const void* a, const void* b, void* Y, int32_t B, int32_t C, int64_t y_bs,
                      pbe_stream_t stream);

/* pbe_image_post_f32 — scripts/inference.py:347: clamp((x+1)/2, 0, 1) of fp16 NHWC [B,HW,ld] -> fp32 NCHW [B,3,HW]. */
int pbe_image_post_f32(const void* src, float* dst, int32_t B, int32_t HW, int32_t ld, pbe_stream_t stream);

/* pbe_resize_bilinear_f32 — scripts/inference.py:332 `Resize([h, w])(mask)` on fp32 planes [planes, Hin, Win] -> [planes, Hout, Wout]:
 * bilinear, align_corners = False; antialias = 1 is the triangle filter of torchvision >= 0.17 (F.interpolate(antialias=True)),
 * antialias = 0 the 2-tap form of the reference's pinned torchvision 0.12. */
int pbe_resize_bilinear_f32(const float* src, float* dst, int32_t planes, int32_t Hin, int32_t Win, int32_t Hout,
                            int32_t Wout, int32_t antialias, pbe_stream_t stream);

/* Image I/O on the device (scripts/inference.py:305-322 pre-processing, :346-399 outputs; test_bench_dataset.py:74-99):
 * pbe_u8_to_planes_f32     u8 HWC [B, H*W, C] -> fp32 CHW planes: (v/255 - mean[c]) / std[c]  (ToTensor + Normalize), or the mask
 *                          forms binarize = 1: (1 - v/255) thresholded at 0.5 (inference.py:311-315), 2: 1 - v/255 (test bench);
 * pbe_mul_planes_f32       out[b,c] = x[b,c] * m[b,0]                                  (inpaint_image = image * mask, :319);
 * pbe_planes_to_u8_canvas  one CHW image -> a rectangle of a u8 HWC canvas: trunc(255 * clamp(x * a[c] + b[c], 0, 1)); the result /
 *                          GT / inpaint / ref / mask files and the 4-tile grid (make_grid, pad 2) are such rectangles. */
int pbe_u8_to_planes_f32(const void* src, float* dst, int32_t B, int32_t C, int32_t HW, const float* mean3, const float* std3,
                         int32_t binarize, pbe_stream_t stream);
int pbe_mul_planes_f32(const float* x, const float* m, float* out, int32_t B, int32_t C, int32_t HW, pbe_stream_t stream);
int pbe_planes_to_u8_canvas(const float* src, void* canvas, int32_t H, int32_t W, int32_t Hc, int32_t Wc, int32_t y0, int32_t x0,
                            const float* a3, const float* b3, int32_t bcast, pbe_stream_t stream);

/* pbe_tune — developer knobs for A/B runs in one process (never needed for correctness):
 * key 1: force an implicit-GEMM tile config index (-1 = heuristic); key 2: allow split-K (0/1);
 * key 3: attention queries-per-wave factor (0 = heuristic, 1, 2); key 4: ping-pong main loop of the halo-resident conv tiles (0/1);
 * key 5: per-launch choice of the XCD tile order (m fastest where that fetches fewer bytes into the 8 L2s; 0 = always n fastest);
 * key 6: attention at d = 40 keeps the softmax reference maximum in the head-dim padding (0 = the multiply-add form);
 * key 7: rows per thread of the two-pass GroupNorm kernels (default 16); key 8: extra dynamic LDS bytes per attention workgroup
 *        (fewer resident workgroups per CU: occupancy experiments).
 */
int pbe_tune(int32_t key, int32_t value);

/* ---- per-kernel-class timing with HIP events on the launch stream (bench.py roofline leg) ---- */
int pbe_prof_enable(int32_t on);
int pbe_prof_reset(void);
/* out[5*k + {0..4}] = {launches, total ms, total work (FLOP for the matrix-core classes, bytes otherwise), total algorithmic bytes,
 * roofline ms = sum over launches of max(FLOP / 2.5e15, bytes / 8e12) - the bound that binds each launch} for class k; returns
 * #classes.  Synchronises the recorded events (call outside any timed region). */
int pbe_prof_collect(double* out, int32_t max_classes);
const char* pbe_prof_class_name(int32_t klass);

#ifdef __cplusplus
}
#endif
#endif /* PBE_HIP_H */
