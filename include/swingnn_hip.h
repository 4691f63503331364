/* libswingnn_hip.so -- C-ABI of the MI355X (gfx950) SwinGNN denoiser hot path.
 *
 * Plain pointers and sizes only; no C++ or torch types cross this boundary.
 * All pointers are DEVICE pointers (contiguous, row-major) unless a comment says "host".
 * `stream` is a hipStream_t passed as void*.  No entry point allocates, frees or
 * synchronises: scratch memory comes in through (ws, ws_bytes), sized by
 * swg_workspace_bytes().  Return value: 0 ok; <0 SWG_E_* (bad shape / unsupported
 * config / workspace too small); >0 a hipError_t from a launch.
 *
 * Citations are file:line under the reference tree (DSL-Lab/SwinGNN).
 * Activations between entry points: fp32 residual stream [B, L, C]; bf16 is used only
 * for MFMA operands inside the library.
 */
#ifndef SWINGNN_HIP_H
#define SWINGNN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SWG_E_SHAPE (-1)       /* inconsistent or unsupported shape                       */
#define SWG_E_UNSUPPORTED (-2) /* configuration outside what the kernels implement        */
#define SWG_E_WORKSPACE (-3)   /* ws_bytes smaller than swg_workspace_bytes()             */
#define SWG_E_NULL (-4)        /* a required pointer is NULL                              */

#define SWG_NOISE_EMB 512 /* noise embedding width, model/swin_gnn/swin_gnn.py:634 */

/* nn.Linear as the kernels consume it: weight converted to bf16 and zero-padded along K to
 * Kp (a multiple of 32); bias stays fp32.  A derived cache of the fp32 nn.Parameter. */
typedef struct swg_lin {
  const void* w;  /* bf16 [N, Kp] row-major */
  const float* b; /* [N] or NULL            */
  int32_t N, K, Kp;
} swg_lin;

/* nn.LayerNorm(dim), eps 1e-5 */
typedef struct swg_ln {
  const float* g; /* weight [dim] */
  const float* b; /* bias   [dim] */
} swg_ln;

/* SwinTransformerBlock (model/swin_gnn/swin_gnn.py:158-277).  win/shift are the EFFECTIVE
 * values after the clamp rule of :189-192. */
typedef struct swg_block {
  int32_t C, res, win, shift, heads;
  int32_t cond_off; /* offset of this block's (scale | shift), 2C floats, inside a cond row */
  swg_ln ln1, ln2;
  swg_lin qkv, proj, fc1, fc2;
  const float* rpb; /* relative_position_bias_table [(2win-1)^2, heads] fp32, :84-85 */
  /* optional derived cache for the fused LN2+fc1+GELU+fc2 kernel (C = 96 or 192): fc1/fc2 weights as a
   * per-64-hidden-chunk LDS image in swizzled MFMA fragment order (swingnn_amd/engine.py mlp_image).
   * NULL -> the GEMM-based path is used. */
  const void* mlp_img;
  /* optional derived cache for the fused window-attention kernels (swingnn_amd/engine.py wattn_ring_image): per head
   * the Wv / Wq / Wk rows as 1-KiB MFMA fragments, then -- C = 96: the Wproj columns in the same stream; C = 192 / 384
   * (head_dim 32, windows 2 / 4 / 6 / 8) and C = 60 / 120 / 240 (head_dim 20, windows 2 / 4 / 8; every head padded to a
   * 32-slot group, channels to a multiple of 32) -- the slot-permuted Wproj [C, heads * 32] behind the stream for the
   * projection GEMM / the fused tail.  NULL -> the 4-kernel path (qkv GEMM, k_wattn_core, projection GEMM) is used. */
  const void* attn_img;
  /* optional derived cache for the fused attention-tail + MLP kernel: the projection weight as MFMA fragments (k-step
   * major, rho-ordered out tiles, input columns in the slot order of the attention output; swingnn_amd/engine.py
   * proj_frag_image).  C = 192 / 384: immediately followed by the mlp_stream image (one allocation, mlp_img points at
   * its second half).  C = 60 / 120 (head_dim 20): the fragments alone, rows padded to 64 / 128, one 32-slot column
   * group per head (mlp_img is the k_mlp_fused image).  NULL -> projection GEMM + MLP kernel. */
  const void* proj_mlp_img;
  /* optional derived cache for k_wattn_ring at 6x6 windows: the relative-position bias (:121-124) of every head in
   * ACCUMULATOR layout, [heads][3 key tiles][3 query tiles][64 lanes] x 4 fp32, scaled by log2(e), -3e38 on padded
   * keys (swg_rpb_tiles).  The kernel loads a head's nine tiles straight into the accumulators of S^T = K Q^T one head
   * ahead.  NULL -> the bias is read from an LDS table per score. */
  const float* rpb_tiles;
} swg_block;

/* PatchMerging (:298-335) at input resolution res x res, input width C */
typedef struct swg_merge {
  int32_t C, res;
  swg_ln ln;   /* LayerNorm(4C) */
  swg_lin red; /* Linear(4C -> 2C), no bias */
  /* optional weight image of the fused kernel (C = 96 / 192; swingnn_amd/engine.py merge_image): gather, LayerNorm and
   * the reduction then run as one pass over the fp32 rows.  NULL -> LayerNorm pass + GEMM. */
  const void* img;
} swg_merge;

/* PatchBreakup (:347-403) at input resolution res x res, input width Cin = 4C' */
typedef struct swg_breakup {
  int32_t Cin, res;
  swg_lin pre;    /* Linear(4C' -> 4C'), no bias */
  swg_ln ln;      /* LayerNorm(4C') */
  swg_ln post_ln; /* LayerNorm(C')  */
  swg_lin post;   /* Linear(C' -> C'), no bias */
  /* optional weight images of the one-pass kernel (Cin = 384; swingnn_amd/engine.py merge_image(pre, 32) and
   * expand_post_image(post)): concat, pre_linear, both LayerNorms, the 2x2 scatter and post_linear then run as one pass
   * over x and skip (taken only when x_out aliases neither input).  Either NULL -> concat pass, GEMM, breakup tail
   * kernel(s). */
  const void* img_pre;
  const void* img_post;
} swg_breakup;

/* BasicLayer (:415-483): [upsample] -> blocks -> [downsample] */
typedef struct swg_layer {
  int32_t n_blocks;
  int32_t is_decoder;
  const swg_block* blocks; /* host pointer to n_blocks entries */
  const swg_merge* down;   /* host pointer or NULL */
  const swg_breakup* up;   /* host pointer or NULL */
} swg_layer;

/* SwinGNN (:587-779) */
typedef struct swg_net {
  int32_t N;         /* img_size = max_node_num                         */
  int32_t p;         /* patch size                                      */
  int32_t in_ch;     /* conv input channels (2 with self-conditioning)  */
  int32_t C0;        /* embed_dim                                       */
  int32_t n_layers;  /* 2 * len(depths): encoder layers then decoder    */
  int32_t self_cond; /* SwinGNN.self_condition                          */
  int32_t cond_total;/* sum over patch_embed + all blocks of 2C         */
  int32_t pe_cond_off;
  /* noise embedding: PositionalEmbedding freqs (:508-510) and the two mapping layers (:705-706) */
  const float* freqs; /* [C0/2] */
  const float *map0_w, *map0_b; /* [512, C0], [512] */
  const float *map1_w, *map1_b; /* [512, 512], [512] */
  /* every `affine` layer stacked: rows [cond_off, cond_off+2C) belong to one block (:195, :554) */
  const float *aff_w, *aff_b; /* [cond_total, 512], [cond_total] */
  /* PatchEmbed (:516-577) */
  const float *pe_w, *pe_b; /* conv weight [C0, in_ch*p*p], bias [C0] */
  swg_ln pe_ln;
  const swg_layer* layers; /* host pointer to n_layers entries */
  swg_ln norm;             /* final LayerNorm (:708) */
  /* read_out (:697-701) collapsed to one affine map per token (no non-linearity between the
   * three convolutions): out[Y*p+i, X*p+j] = sum_c tok[c] * ro_w[c, i*p+j] + ro_b */
  const float* ro_w; /* [C0, p*p] */
  float ro_b;
} swg_net;

/* ---- library ------------------------------------------------------------------------- */
int swg_version(void);
/* Bytes of scratch that swg_precond_forward / the op entry points need for batch B. */
size_t swg_workspace_bytes(const swg_net* net, int32_t B);
/* Number of ops of one pass over B graphs that take a batch-gated one-pass kernel (streamed MLP, projection + MLP
 * tail, one-pass PatchMerging / PatchBreakup) instead of the GEMM route.  Host-side planning only (no launch): the
 * caller that issues a pass as part-batches (swingnn_amd/engine.py network_pass; the reference's per-rank batch,
 * runner/sampler/sampler_utils.py:254-258) compares the count of the part-batch with the count of the whole batch and
 * does not split when the part-batch would fall below a gate.  <0: SWG_E_*. */
int32_t swg_fused_op_count(const swg_net* net, int32_t B);

/* Relative-position bias of one block as accumulator-layout tiles (see swg_block.rpb_tiles): rpb [(2win-1)^2, heads]
 * -> tiles [heads][NTM][NTM][64][4] fp32 with NTM = ceil(win*win / 16).  Derived cache, rebuilt when rpb changes. */
int swg_rpb_tiles(const float* rpb, int32_t heads, int32_t win, float* tiles, void* stream);

/* ---- whole network pass -------------------------------------------------------------- *
 * mode 1: Precond.forward core, model/precond/precond.py:47-51 (edm branch of
 *         runner/objectives/edm.py:122-126): D = mask(c_skip*x + c_out*F(c_in*x, ln(sigma)/4, self_cond)).
 * mode 0: SwinGNN.forward, model/swin_gnn/swin_gnn.py:755-779: `labels` are the noise labels,
 *         x is used as given, output = sym(mask(F)).
 * labels: [B], or one value when labels_uniform != 0 (the sampler's case, mcmc_sampler/edm.py:180).
 * self_cond may be NULL (zeros channel, :770).  sym_err (nullable): device float that
 * accumulates sum|D - D^T| (the quantity asserted in utils/graph_utils.py:89-94). */
int swg_precond_forward(const swg_net* net, int32_t B, int32_t mode, const float* x, const float* node_flags,
                        const float* labels, int32_t labels_uniform, const float* self_cond, float* out,
                        float* sym_err, void* ws, size_t ws_bytes, void* stream);
/* swg_precond_forward with the conditioning rows handed in.  cond_rows: device [Bc, cond_total] fp32 (Bc = 1 when
 * labels_uniform, else B), row b = the stacked (scale | shift) vectors of every block for label b, as swg_cond_table
 * (or swg_noise_embed) produces them; NULL = compute them here (then identical to swg_precond_forward).  In sampling
 * sigma is one value per Heun step for the whole batch (runner/mcmc_sampler/edm.py:180) and the 511 model calls of a
 * sample use only num_steps distinct values, so the sampler builds the table once per sample() and every pass skips
 * the noise-embedding chain (PositionalEmbedding + map_layer0/1, model/swin_gnn/swin_gnn.py:757-760, and the stacked
 * affine layers :238-239 / :574-575): four small dependent launches at the head of each pass.  Same arithmetic, same
 * bits.  `labels` is still read (preconditioning coefficients). */
int swg_precond_forward_cond(const swg_net* net, int32_t B, int32_t mode, const float* x, const float* node_flags,
                             const float* labels, int32_t labels_uniform, const float* cond_rows,
                             const float* self_cond, float* out, float* sym_err, void* ws, size_t ws_bytes, void* stream);
/* Conditioning rows for n labels at once: cond_rows[i] = affine_all(silu(map1(silu(map0(PosEmb(c_noise_i)))))) with
 * c_noise = ln(sigma)/4 (mode 1, runner/objectives/edm.py:126) or the label itself (mode 0).  ws: scratch of at least
 * 4 * n * (5 + C0 + 2 * SWG_NOISE_EMB) bytes + 1 KiB. */
int swg_cond_table(const swg_net* net, int32_t n, int32_t mode, const float* labels, float* cond_rows, void* ws,
                   size_t ws_bytes, void* stream);

/* ---- op-level entry points (each replaces one reference module's forward) ------------ */
/* a8: PositionalEmbedding + map_layer0/1 (:507-513, :757-760) and every affine layer:
 * c_noise [Bc] -> cond [Bc, cond_total]. */
int swg_noise_embed(const swg_net* net, int32_t Bc, const float* c_noise, float* cond, void* ws, size_t ws_bytes,
                    void* stream);
/* a9 (+ self-cond concat :769-771, + c_in scaling precond.py:47): x [B,N,N], self_cond [B,N,N] or NULL,
 * c_in [B] or NULL (=1), cond rows of stride cond_stride (0 = one row for the whole batch) -> tok [B, L, C0] */
int swg_patch_embed_fwd(const swg_net* net, int32_t B, const float* x, const float* self_cond, const float* c_in,
                        int32_t c_in_stride, const float* cond, int32_t cond_stride, float* tok, void* stream);
/* a1 + a2 + first half of a3 (:238-272): x <- silu(shift + x(1+scale)); x <- x + Attn(LN1(x)) with the
 * window partition / cyclic shift folded into addressing.  x_in, x_out [B, res*res, C] fp32. */
int swg_wattn_fwd(const swg_block* blk, int32_t B, const float* x_in, const float* cond, int32_t cond_stride,
                  float* x_out, void* ws, size_t ws_bytes, void* stream);
/* One SwinTransformerBlock.forward (model/swin_gnn/swin_gnn.py:232-277) = swg_wattn_fwd then swg_ln_mlp_fwd; where
 * the block carries proj_mlp_img (C = 192 / 384; 60 / 120 at head_dim 20) the projection, the shortcut and the MLP half run as one kernel behind the
 * attention core.  x_in == x_out allowed.  This is what the network pass (swg_precond_forward) runs per block. */
int swg_block_fwd(const swg_block* blk, int32_t B, const float* x_in, const float* cond, int32_t cond_stride,
                  float* x_out, void* ws, size_t ws_bytes, void* stream);
/* a4 + LN2 + residual (:275): x_out = x_in + fc2(GELU(fc1(LN2(x_in)))) */
int swg_ln_mlp_fwd(const swg_block* blk, int32_t B, const float* x_in, float* x_out, void* ws, size_t ws_bytes,
                   void* stream);
/* a5: [B, res*res, C] -> [B, res*res/4, 2C] */
int swg_patch_merge_fwd(const swg_merge* m, int32_t B, const float* x_in, float* x_out, void* ws, size_t ws_bytes,
                        void* stream);
/* a6 incl. the concat of :743: x [B, res*res, Cin/2], skip [B, res*res, Cin/2] -> [B, 4*res*res, Cin/4] */
int swg_patch_expand_fwd(const swg_breakup* u, int32_t B, const float* x, const float* skip, float* x_out, void* ws,
                         size_t ws_bytes, void* stream);
/* a10 tail + a11 + a13 epilogue: final LN, collapsed read-out, mask, (F+F^T)/2, then
 * out = mask(c_skip*x + c_out*F); coef [B,4] = (c_skip, c_out, c_in, c_noise) or NULL (c_skip 0, c_out 1). */
int swg_readout_sym_precond_fwd(const swg_net* net, int32_t B, const float* tok, const float* x,
                                const float* node_flags, const float* coef, float* out, float* sym_err, void* ws,
                                size_t ws_bytes, void* stream);

/* nn.Linear forward on bf16 operands (the building block behind qkv/proj/fc1/fc2/reduction/pre/post_linear):
 * out[M,N] = act(a[M,Kp] @ w[N,Kp]^T + bias) + residual; a, w bf16 (K zero-padded to Kp, a multiple of 32);
 * exactly one of out_f32 / out_bf16; residual (fp32, stride ldo) only with out_f32; act 0 none, 1 exact GELU. */
int swg_linear_bf16(int32_t M, int32_t N, int32_t Kp, const void* a, int32_t lda, const void* w, const float* bias,
                    int32_t act, const float* residual, float* out_f32, void* out_bf16, int32_t ldo, void* stream);

/* ---- EDM Heun sampler updates (runner/mcmc_sampler/edm.py:168-217, shipped defaults) -- */
/* :174-176  x_hat = mask(x + noise_coef * sym(raw)),  sym(raw) = triu(raw,1) + triu(raw,1)^T
 * (utils/graph_utils.py:113-119).  All tensors [B,N,N] fp32. */
int swg_heun_churn(int32_t B, int32_t N, const float* x, const float* raw_noise, const float* node_flags,
                   float noise_coef, float* x_hat, void* stream);
/* :188-193  d = mask((x_hat - mask(D))/t_hat); x_prime = x_hat + h*d (not masked); if last step
 * (:197) callers use x_prime masked -> `mask_out` != 0.  d_cur is written when non-NULL. */
int swg_heun_euler(int32_t B, int32_t N, const float* x_hat, const float* denoised, const float* node_flags,
                   float t_hat, float h, int32_t mask_out, float* d_cur, float* x_prime, void* stream);
/* :211-215  d' = (x_prime - D2)/t_prime; x_next = mask(x_hat + h*(0.5 d_cur + 0.5 d')) */
int swg_heun_correct(int32_t B, int32_t N, const float* x_hat, const float* x_prime, const float* d_cur,
                     const float* denoised2, const float* node_flags, float t_prime, float h, float* x_next,
                     void* stream);
/* utils/graph_utils.py:5-38 (in place) and :89-94 (acc += sum|a - a^T|, no host sync) */
int swg_mask_adjs(int32_t B, int32_t N, float* a, const float* node_flags, void* stream);
int swg_sym_check(int32_t B, int32_t N, const float* a, float* acc, void* stream);

/* ---- molecule path: node features + multi-channel adjacency ----------------------- *
 * NodeAdjSwinGNN (model/swin_gnn/swin_gnn.py:782-1021) shares the SwinGNN trunk; `net` describes it with
 * in_ch = (self_cond ? 2 : 1) * (Ca + 2 Cn) convolution inputs, and ro_w / ro_b unused.  Derived read-out operands:
 *   ro       ConvTranspose2d(C0,C0,p,p) -> Conv1x1 -> Conv1x1 (:900-904, no non-linearity) collapsed to one map
 *            [C0] -> [p*p][Cp] per token (Cp = ceil32(C0), zero rows on the padding): bf16 [p*p*Cp, Cp], bias fp32
 *   adj_fc1 / node_fc1   readout_{adj,node}_mlp.fc1 (:913-914) as bf16 GEMM operands
 *   *_fc2    the narrow output layers (C0 -> Ca / Cn) in fp32 */
typedef struct swg_nodeadj {
  int32_t Ca, Cn;
  swg_lin ro;
  swg_lin adj_fc1, node_fc1;
  const float *adj_fc2_w, *adj_fc2_b;   /* [Ca, C0], [Ca] */
  const float *node_fc2_w, *node_fc2_b; /* [Cn, C0], [Cn] */
} swg_nodeadj;
size_t swg_nodeadj_workspace_bytes(const swg_net* net, const swg_nodeadj* na, int32_t B);
/* mode 1: NodeAdjPrecond.forward core, model/precond/precond.py:100-105:
 *           (D_adj, D_node) = mask(c_skip * (adj, node) + c_out * F(c_in * adj, c_in * node, ln(sigma)/4, self-cond pair))
 * mode 0: NodeAdjSwinGNN.forward (:960-1021), `labels` are the noise labels.
 * adj, sc_adj, out_adj: [B, Ca, N, N]; node, sc_node, out_node: [B, N, Cn]; sc_* may be NULL (zero channels, :985-988).
 * sym_err accumulates sum|D_adj - D_adj^T| (precond.py:106-107). */
int swg_nodeadj_forward(const swg_net* net, const swg_nodeadj* na, int32_t B, int32_t mode, const float* adj,
                        const float* node, const float* node_flags, const float* labels, int32_t labels_uniform,
                        const float* sc_adj, const float* sc_node, float* out_adj, float* out_node, float* sym_err,
                        void* ws, size_t ws_bytes, void* stream);
/* NodeAdjEDMSampler node-feature updates (runner/mcmc_sampler/edm.py:367-426; the adjacency planes go through
 * swg_heun_* with B*Ca planes).  x [B, N, Cn] fp32, flags [B, N]:
 *   churn  : x_hat = mask(x + noise_coef * raw)                                  (:367-370)
 *   euler  : d = mask((x_hat - mask(D)) / t_hat); x_prime = x_hat + h d  (masked when mask_out, :399/:426)
 *   correct: d' = (x_prime - mask(D2)) / t_prime; x_next = mask(x_hat + h (d/2 + d'/2))      (:418-426) */
int swg_heun_nodes_churn(int32_t B, int32_t N, int32_t Cn, const float* x, const float* raw, const float* node_flags,
                         float noise_coef, float* x_hat, void* stream);
int swg_heun_nodes_euler(int32_t B, int32_t N, int32_t Cn, const float* x_hat, const float* denoised,
                         const float* node_flags, float t_hat, float h, int32_t mask_out, float* d_cur, float* x_prime,
                         void* stream);
int swg_heun_nodes_correct(int32_t B, int32_t N, int32_t Cn, const float* x_hat, const float* x_prime,
                           const float* d_cur, const float* denoised2, const float* node_flags, float t_prime, float h,
                           float* x_next, void* stream);

/* ---- EDM training objective and loss, forward only ---- */
/* runner/objectives/edm.py:160-212 (get_input_output with sigma_dist = precond = 'edm'), randn draws factored out:
 *   sigma = exp(1.2 r - 1.2), w = (sigma^2 + 1/4) / (sigma/2)^2 (:174-176), noisy = mask(clean + sigma sym(raw)),
 *   noise_added = mask(sigma sym(raw)) (:181-186 -> utils/graph_utils.py:122-152), and the preconditioning scalars of
 *   :122-126.  noise_added and the four c_* outputs may be NULL.  rnd_normal, sigmas, weights, c_*: [B]. */
int swg_edm_objective(int32_t B, int32_t N, const float* clean, const float* node_flags, const float* rnd_normal,
                      const float* raw_noise, float* noisy, float* noise_added, float* sigmas, float* weights,
                      float* c_skip, float* c_out, float* c_in, float* c_noise, void* stream);
/* loss/rainbow_loss.py:84-111, objective 'edm', no reweighting:  S_b = w_b sum_ij mask (pred - target)^2,
 *   reduction 'none' (reduction_mean = 0): loss_b = S_b / (sum flags_b)^2 * regression_loss_weight
 *   reduction 'mean' (:94-96, = 1):        loss_b = (sum_b S_b) / (sum flags_b)^2 * regression_loss_weight
 * weights may be NULL (ones).  scratch2B: 2*B floats of device scratch.  Deterministic summation order. */
int swg_edm_loss(int32_t B, int32_t N, const float* pred, const float* target, const float* node_flags,
                 const float* weights, float regression_loss_weight, int32_t reduction_mean, float* loss,
                 float* scratch2B, void* stream);


/* swg_edm_loss without the scratch2B argument.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  Same
 * kernels and results as swg_edm_loss; the 2*B partial sums live in a 512-KiB file-scope device array of the library
 * (one copy per GPU, held until the process ends).  B <= 65536, else SWG_E_WORKSPACE.  Calls no hipMalloc / hipFree.
 * The current device must be the stream's device.  All calls on one GPU share the array: calls that overlap on one
 * GPU (different streams / threads) race; stream-ordered calls are safe (nothing needs to start at zero). */
int swg_edm_loss_noalloc(int32_t B, int32_t N, const float* pred, const float* target, const float* node_flags,
                         const float* weights, float regression_loss_weight, int32_t reduction_mean, float* loss,
                         void* stream);
/* Set-up half of swg_edm_loss_noalloc (no counterpart in the reference either): resolves the address of the array on
 * the CURRENT device (hipGetSymbolAddress, once per device; may load the code object).  Launches nothing.  Callers that
 * will use swg_edm_loss_noalloc under stream capture call this once per device beforehand; otherwise the entry point's
 * first call on a device does it. */
int swg_edm_loss_noalloc_warmup(void);

/* ---- training step: backward passes of the op-level entry points ------------------------ *
 * The reference has no explicit backward code: `loss.backward()` (runner/trainer/trainer.py:135) differentiates the
 * module forwards cited above with autograd.  Each *_bwd entry point below is the derivative of the *_fwd entry point
 * of the same name: it RECOMPUTES the op's intermediates from the saved op input (nothing but op inputs is kept
 * between forward and backward), runs the Linear-layer gradients on the bf16 MFMA GEMM (dX = dY W, dW = dY^T X with a
 * split reduction) and everything else in fp32 row kernels.  Parameter gradients are ACCUMULATED (+=) into the
 * caller's fp32 buffers (the layout of the nn.Parameter); activation gradients are written.  Scratch comes from
 * (ws, ws_bytes) >= swg_train_workspace_bytes().  Gradients carry bf16 operand rounding (relative 2^-9 per product
 * term, fp32 accumulation), the usual mixed-precision training contract. */
typedef struct swg_lin_train {
  const void* wt; /* bf16 [K, Np]: the weight TRANSPOSED, N zero-padded to Np = ceil32(N) (derived cache)   */
  float* dw;      /* fp32 [N, K] gradient accumulator                                                      */
  float* db;      /* fp32 [N] or NULL                                                                      */
} swg_lin_train;
typedef struct swg_ln_train {
  float* dg; /* [dim] */
  float* db; /* [dim] */
} swg_ln_train;
typedef struct swg_block_train {
  swg_ln_train ln1, ln2;
  swg_lin_train qkv, proj, fc1, fc2;
  float* drpb; /* [(2win-1)^2, heads] */
  /* optional derived cache for the fused front half of the MLP backward (C = 96, 192): per 64-wide hidden chunk the
   * fc1 rows and the fc2^T rows in the LDS row layout of mlp_img (swingnn_amd/engine.py mlp_bwd_front_image).
   * NULL -> LayerNorm, fc1 GEMM, cast, fc2^T GEMM and the GELU pass run as separate kernels. */
  const void* mlp_bwd_img;
} swg_block_train;
typedef struct swg_merge_train {
  swg_ln_train ln;
  swg_lin_train red;
} swg_merge_train;
typedef struct swg_breakup_train {
  swg_lin_train pre;
  swg_ln_train ln, post_ln;
  swg_lin_train post;
} swg_breakup_train;
typedef struct swg_ends_train {
  const void* pe_wb; /* bf16 [C0, ceil32(in_ch p p)]: conv weight as a padded GEMM operand                  */
  float* dpe_w;      /* [C0, in_ch p p]                                                                     */
  float* dpe_b;      /* [C0]                                                                                */
  swg_ln_train pe_ln, norm;
  const void* ro_wb; /* bf16 [C0, ceil32(p p)]: the collapsed read-out map as a padded GEMM operand         */
  float* dro_w;      /* [C0, p p] gradient of the collapsed map (the caller chains it to read_out.{0,1,2})  */
  float* dro_b;      /* [32]: column sums of the pixel gradients; sum of the first p p = d(collapsed bias)  */
} swg_ends_train;

size_t swg_train_workspace_bytes(const swg_net* net, int32_t B);

/* Weight gradient of a token-wise nn.Linear on bf16 operands (autograd of qkv / proj / fc1 / fc2 / reduction /
 * pre_linear / post_linear, model/swin_gnn/swin_gnn.py:19-25, :113, :137, :334, :385-403; the op behind every dW of
 * the train step, runner/trainer/trainer.py:130):  dw[N,K] (dense fp32) += dy[rows, :N]^T x[rows, :K].
 * dy bf16 [rows, ldy], x bf16 [rows, ldx] TOKEN-MAJOR as the forward wrote them (ldy, ldx multiples of 8, >= N, K;
 * K a multiple of 4).  Workspace >= swg_linear_wgrad_workspace_bytes(N, K) (fp32 partials of the token slices). */
size_t swg_linear_wgrad_workspace_bytes(int32_t N, int32_t K);
int swg_linear_wgrad_bf16(int32_t rows, int32_t N, int32_t K, const void* dy, int32_t ldy, const void* x, int32_t ldx,
                          float* dw, void* ws, size_t ws_bytes, void* stream);
/* (c_skip, c_out, c_in, c_noise) rows of runner/objectives/edm.py:122-126 for `labels` = sigmas (mode 1) or
 * (0, 1, 1, label) (mode 0): coef [B,4], c_noise [B]. */
int swg_edm_coef(int32_t B, const float* labels, int32_t labels_uniform, int32_t mode, float* coef, float* c_noise,
                 void* stream);
/* d swg_wattn_fwd: dy [B,L,C] -> dx [B,L,C]; dcond rows (stride dcond_stride) get the block's (dscale | dshift). */
int swg_wattn_bwd(const swg_block* blk, const swg_block_train* tr, int32_t B, const float* x_in, const float* cond,
                  int32_t cond_stride, const float* dy, float* dx, float* dcond, int32_t dcond_stride, void* ws,
                  size_t ws_bytes, void* stream);
/* d swg_ln_mlp_fwd */
int swg_ln_mlp_bwd(const swg_block* blk, const swg_block_train* tr, int32_t B, const float* x_in, const float* dy,
                   float* dx, void* ws, size_t ws_bytes, void* stream);
/* d swg_patch_merge_fwd: dy [B, res*res/4, 2C] -> dx [B, res*res, C] */
int swg_patch_merge_bwd(const swg_merge* m, const swg_merge_train* tr, int32_t B, const float* x_in, const float* dy,
                        float* dx, void* ws, size_t ws_bytes, void* stream);
/* d swg_patch_expand_fwd: dy [B, 4*res*res, Cin/4] -> dx, dskip [B, res*res, Cin/2] */
int swg_patch_expand_bwd(const swg_breakup* u, const swg_breakup_train* tr, int32_t B, const float* x, const float* skip,
                         const float* dy, float* dx, float* dskip, void* ws, size_t ws_bytes, void* stream);
/* d swg_patch_embed_fwd w.r.t. its parameters and the conditioning row (the adjacency input is data) */
int swg_patch_embed_bwd(const swg_net* net, const swg_ends_train* tr, int32_t B, const float* x, const float* self_cond,
                        const float* c_in, int32_t c_in_stride, const float* cond, int32_t cond_stride, const float* dtok,
                        float* dcond, int32_t dcond_stride, void* ws, size_t ws_bytes, void* stream);
/* d swg_readout_sym_precond_fwd: dD [B,N,N] -> dtok [B,L,C0] (+ final LayerNorm and collapsed read-out gradients) */
int swg_readout_sym_precond_bwd(const swg_net* net, const swg_ends_train* tr, int32_t B, const float* tok,
                                const float* node_flags, const float* coef, const float* dD, float* dtok, void* ws,
                                size_t ws_bytes, void* stream);
/* d swg_edm_loss (reduction 'none', the trainer's call :120-129): grad_loss [B] -> grad_pred [B,N,N] */
int swg_edm_loss_bwd(int32_t B, int32_t N, const float* pred, const float* target, const float* node_flags,
                     const float* weights, const float* grad_loss, float regression_loss_weight, float* grad_pred,
                     void* stream);

/* ---- training step of the molecule path (runner/trainer/trainer_node_adj.py:95-131) ----------------------- *
 * Op-level forward halves of swg_nodeadj_forward (the trunk between them goes through swg_wattn_fwd, ... above), their
 * backward passes, and the node-feature halves of NodeAdjEDMObjectiveGenerator / NodeAdjRainbowLoss (the adjacency
 * halves are swg_edm_objective / swg_edm_loss over the B*Ca planes). */
typedef struct swg_nodeadj_train {
  const void* ro_wt;            /* bf16 [C0, p*p*Cp]: the collapsed read-out map transposed (dxn = drep Wro)        */
  float* dro_w;                 /* fp32 [p*p*Cp, C0] gradient of the collapsed map (caller chains it to read_out.*)  */
  float* dro_b;                 /* fp32 [p*p*Cp]                                                                      */
  swg_lin_train adj_fc1, node_fc1;
  float *dadj_fc2_w, *dadj_fc2_b;   /* [Ca, C0], [Ca] */
  float *dnode_fc2_w, *dnode_fc2_b; /* [Cn, C0], [Cn] */
  swg_ln_train norm;
} swg_nodeadj_train;
size_t swg_nodeadj_train_workspace_bytes(const swg_net* net, const swg_nodeadj* na, int32_t B);
/* input planes (:985-996, c_in of precond.py:100) + PatchEmbed (:562-577) -> tok [B, L, C0] */
int swg_nodeadj_embed_fwd(const swg_net* net, const swg_nodeadj* na, int32_t B, const float* adj, const float* node,
                          const float* sc_adj, const float* sc_node, const float* node_flags, const float* c_in,
                          int32_t c_in_stride, const float* cond, int32_t cond_stride, float* tok, void* stream);
/* final norm + read_out (:953-956) + MLP heads + pooling + masks / symmetrisation (:1000-1018) + precond epilogue
 * (precond.py:102-105; coef [B,4] or NULL = NodeAdjSwinGNN.forward) */
int swg_nodeadj_readout_fwd(const swg_net* net, const swg_nodeadj* na, int32_t B, const float* tok, const float* adj,
                            const float* node, const float* node_flags, const float* coef, float* out_adj,
                            float* out_node, float* sym_err, void* ws, size_t ws_bytes, void* stream);
/* d swg_nodeadj_embed_fwd w.r.t. its parameters and the conditioning row (inputs are data) */
int swg_nodeadj_embed_bwd(const swg_net* net, const swg_nodeadj* na, const swg_ends_train* tr, int32_t B,
                          const float* adj, const float* node, const float* sc_adj, const float* sc_node,
                          const float* node_flags, const float* c_in, int32_t c_in_stride, const float* cond,
                          int32_t cond_stride, const float* dtok, float* dcond, int32_t dcond_stride, void* ws,
                          size_t ws_bytes, void* stream);
/* d swg_nodeadj_readout_fwd: (d_adj [B,Ca,N,N], d_node [B,N,Cn]) -> dtok [B,L,C0] + parameter gradients */
int swg_nodeadj_readout_bwd(const swg_net* net, const swg_nodeadj* na, const swg_nodeadj_train* tr, int32_t B,
                            const float* tok, const float* node_flags, const float* coef, const float* d_adj,
                            const float* d_node, float* dtok, void* ws, size_t ws_bytes, void* stream);
/* runner/objectives/edm.py:243-247: noisy = clean + mask(sigma_b * raw), x [B, N, Cn] */
int swg_nodes_objective(int32_t B, int32_t N, int32_t Cn, const float* clean, const float* raw_noise,
                        const float* node_flags, const float* sigmas, float* noisy, void* stream);
/* loss/rainbow_loss.py:211-248 (reduction 'none'): loss_b = w_b sum flag (pred - target)^2 / sum(flags_b) / Cn * weight */
int swg_nodes_loss(int32_t B, int32_t N, int32_t Cn, const float* pred, const float* target, const float* node_flags,
                   const float* weights, float node_loss_weight, float* loss, void* stream);
int swg_nodes_loss_bwd(int32_t B, int32_t N, int32_t Cn, const float* pred, const float* target, const float* node_flags,
                       const float* weights, const float* grad_loss, float node_loss_weight, float* grad_pred,
                       void* stream);

/* ---- optimizer tail of the train step (runner/trainer/trainer.py:136-140) ---------------- *
 * Multi-tensor kernels over all parameters.  p_ptrs / g_ptrs: DEVICE arrays of n_tensors fp32 pointers (the
 * nn.Parameters and their .grad); numel [n_tensors] int32; flat_off [n_tensors] int64 = offset of the tensor inside
 * the flat moment / EMA buffers; (chunk_tensor, chunk_off) [n_chunks] int32: chunk c covers elements
 * [chunk_off[c], chunk_off[c] + swg_optim_chunk()) of tensor chunk_tensor[c].  All tables are device memory.
 * swg_grad_sqnorm: *sqnorm = sum of squares of every gradient element (the call zeroes it first) --
 *   torch.nn.utils.clip_grad_norm_(norm_type=2) before the scaling.
 * swg_adam_ema_step: g <- g * min(1, max_norm / (sqrt(*sqnorm) + 1e-6)) when sqnorm != NULL (clip_grad_norm_,
 *   :136); torch.optim.Adam update with bias correction at step `step` >= 1 (utils/learning_utils.py:245-248;
 *   weight_decay is Adam's L2 term); then for k < n_ema (<= 8): ema_k <- lerp(ema_k, p, 1 - ema_decay[k])
 *   (ema_pytorch's update_moving_average as used at :268; ema_decay is a HOST array, decay 0 = copy).
 *   m, v [total]; ema [n_ema, total] fp32. */
int swg_optim_chunk(void);
int swg_grad_sqnorm(const void* g_ptrs, const int32_t* numel, const int32_t* chunk_tensor, const int32_t* chunk_off,
                    int32_t n_chunks, float* sqnorm, void* stream);
int swg_adam_ema_step(const void* p_ptrs, const void* g_ptrs, const int32_t* numel, const int64_t* flat_off,
                      const int32_t* chunk_tensor, const int32_t* chunk_off, int32_t n_chunks, float* m, float* v,
                      float* ema, int64_t total, int32_t n_ema, const float* ema_decay, const float* sqnorm,
                      float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                      void* stream);

/* ---- post-sampling: attribute codecs of the molecule path -------------------------------- *
 * utils/attribute_code.py:14-329.  Node attributes [B,N] (int / ddpm) or [B,N,C] (one-hot / bits); adjacency
 * attributes [B,N,N] or [B,C,N,N]; node_flags [B,N] fp32 0/1 (rows AND columns are masked for adjacency entries).
 * encoding: 0 ddpm (:124-177 / :241-255), 1 one-hot (:213-238 / :284-305), 2 bits (:180-210 / :258-281).  Integer
 * attributes travel as fp32 (exact below 2^24), as the reference's float tensors do.
 * swg_attr_decode: `in` -> integer attribute.  ddpm: interval bounds lo / hi [num_types] (DEVICE arrays: the fp32 casts
 *   of the reference's doubles, :133-150); C must be 1.  ddpm_range: the one-hot / bits channels are -1/+1 (else 0/1).
 *   binarise != 0 first applies the sampler's quantisation (runner/sampler/sampler_node_adj.py:179-181, :191-193):
 *   clamp to [-1,1], then channel > 0 -> +1 else -1.  clamp_int: bits only, clamp to [0, num_types-1] (:203-204).
 * swg_attr_encode: integer attribute -> `out` in the given encoding (C channels; C = ceil(log2(num_types)) for bits). */
int swg_attr_decode(int32_t encoding, int32_t B, int32_t N, int32_t C, int32_t num_types, int32_t is_adj,
                    int32_t ddpm_range, int32_t binarise, int32_t clamp_int, const float* in, const float* node_flags,
                    const float* lo, const float* hi, float* out_int, void* stream);
int swg_attr_encode(int32_t encoding, int32_t B, int32_t N, int32_t C, int32_t num_types, int32_t is_adj,
                    int32_t ddpm_range, const float* in_int, const float* node_flags, float* out, void* stream);

/* ---- post-sampling: binarised-graph statistics and MMD ---------------------------------- *
 * evaluation/stats.py:26-175 (degree_stats, clustering_stats, spectral_stats) and evaluation/mmd.py:82-163
 * (gaussian_tv, compute_mmd with is_hist=True) for graphs given as adjacency matrices adj [B,N,N] (any non-zero entry
 * is an edge; symmetrised, self loops and isolated nodes dropped as adjs_to_graphs does, stats.py:291-305; N <= 1024).
 * swg_graph_hists: deg_hist [B,N] (nx.degree_histogram, zero padded), clu_hist [B,cluster_bins] (np.histogram of
 *   nx.clustering over [0,1]), and, when lap != NULL, the normalised Laplacian [B,N,N] whose removed nodes carry
 *   iso_diag on the diagonal (choose it outside the histogram range: their eigenvalues are then ignored).  All fp64.
 * swg_hist_pmf: np.histogram(vals[b], bins, range=(first,last)) / count for each row of vals [B,n] (the 200-bin
 *   spectral pmf of stats.py:133-135 from the Laplacian's eigenvalues).  snap = 0: exactly numpy (a value outside the
 *   range is dropped); snap > 0: values up to `snap` outside are counted in the end bins (eigenvalues of a normalised
 *   Laplacian lie in [0,2]; whether a solver returns 2 or 2 + 4e-16 should not decide if a node is counted).
 * swg_mmd_gaussian_tv: *out = MMD^2 between the histogram sets X [n1,L] and Y [n2,L] (rows zero padded to a common
 *   L; each row is normalised to sum 1 unless it sums to 0).  scratch: 3*(n1+n2) doubles.  Deterministic order. */
size_t swg_graph_stats_workspace_bytes(int32_t B, int32_t N);
int swg_graph_hists(int32_t B, int32_t N, int32_t cluster_bins, const float* adj, double* deg_hist, double* clu_hist,
                    double* lap, double iso_diag, void* ws, size_t ws_bytes, void* stream);
int swg_hist_pmf(int32_t B, int32_t n, int32_t bins, const double* vals, double first, double last, double snap,
                 double* out, void* stream);
int swg_mmd_gaussian_tv(int32_t n1, int32_t n2, int32_t L, const double* X, const double* Y, double sigma, double* out,
                        double* scratch, void* stream);

/* ---- derived-operand refresh (no counterpart in the reference, which reads the fp32 parameters directly) ---- *
 * After an optimizer step every bf16 operand derived from the fp32 nn.Parameters (K-padded copies, transposed copies
 * for the backward GEMMs, fragment-order images of the fused kernels) is rewritten IN PLACE by one launch, so the
 * descriptors above stay valid.  items / chunk tables are DEVICE arrays.  Chunk c works on item chunk_item[c]:
 *   mode 0  gather : dst[i] = idx[i] == 0 ? 0 : concat(src0[0..n0), src1)[idx[i] - 1], i in [16384 c0, 16384 (c0+1)) < count
 *   mode 1  pad    : dst [N, ld] = src0 [N, K] with zero columns K..ld,             same linear chunking over N*ld
 *   mode 2  padT   : dst [K, ld] = src0 [N, K] transposed with zero columns N..ld,   64 x 64 tile (c0: k block, c1: n block) */
typedef struct swg_refresh_item {
  const float* src0;
  const float* src1;
  void* dst;          /* bf16 */
  const int32_t* idx; /* mode 0 */
  int32_t mode, n0, N, K, ld, pad_;
  int64_t count;
} swg_refresh_item;
int swg_refresh_derived(const swg_refresh_item* items, const int32_t* chunk_item, const int32_t* chunk_c0,
                        const int32_t* chunk_c1, int32_t n_chunks, void* stream);

/* ---- measurement aid (no counterpart in the reference) ---------------------------------- *
 * When enabled, every internal kernel launch is bracketed by HIP events on its stream and its
 * algorithmic FLOPs / bytes are tallied per kernel.  swg_prof_read synchronises the recorded events.
 * Keep it off under HIP graph capture and inside timed throughput regions. */
int swg_prof_enable(int32_t on);
int swg_prof_reset(void);
int swg_prof_kernels(void);
const char* swg_prof_name(int32_t kid);
int swg_prof_read(int32_t kid, double* ms, double* flops, double* bytes, int64_t* launches);

#ifdef __cplusplus
}
#endif
#endif /* SWINGNN_HIP_H */
