// extern "C" backward entry points (include/swingnn_hip.h, "training step") and their host-side sequencing.
// Every op recomputes its forward intermediates from the saved op input with the unfused kernels (row kernels +
// MFMA GEMM + attention core), then walks the chain backwards.  Nothing allocates or synchronises; all buffers
// are carved from the caller's workspace.  A Carver in `measure` mode only adds up sizes, so
// swg_train_workspace_bytes() and the ops can never disagree about a buffer.
#include "common.h"

namespace {

struct Carver {
  char* p;
  size_t left, used;
  bool ok, measure;
  Carver(void* ws, size_t bytes) : p((char*)ws), left(bytes), used(0), ok(ws != nullptr), measure(false) {}
  Carver() : p(nullptr), left(0), used(0), ok(true), measure(true) {}
  template <typename T>
  T* take(size_t n) {
    const size_t b = align256(n * sizeof(T));
    used += b;
    if (measure) return (T*)(uintptr_t)256;
    if (!ok || b > left) { ok = false; return nullptr; }
    T* r = (T*)p;
    p += b;
    left -= b;
    return r;
  }
};

inline int kp(int k) { return round_up(k, 32); }
inline int ldq_of(int C) { return round_up(3 * C, 8); }

struct WgradScratch {
  bf16 *tA, *tB;
  float* partial;
  size_t partial_floats;
};
WgradScratch take_wgrad(Carver& cv, int Nmax, int Kmax, int rows) {
  WgradScratch w;
  const int Tp = round_up(rows, 32);
  w.tA = cv.take<bf16>((size_t)Nmax * Tp);
  w.tB = cv.take<bf16>((size_t)Kmax * Tp);
  w.partial_floats = gemm_splitk_ws_floats(Nmax, Kmax);
  w.partial = cv.take<float>(w.partial_floats);
  return w;
}

// dW[N,K] += dY^T X.  dY comes as fp32 [rows, N] (dy_f32) or bf16 [rows, ldy]; X bf16 [rows, ldx].
int wgrad(const float* dy_f32, const bf16* dy_b, int ldy, int N, const bf16* xb, int ldx, int K, int rows, float* dw,
          const WgradScratch& w, hipStream_t s) {
  const int Tp = round_up(rows, 32);
  int r;
  if (dy_f32) r = launch_transpose_f32(dy_f32, rows, N, ldy, w.tA, Tp, s);
  else r = launch_transpose_bf16(dy_b, rows, N, ldy, w.tA, Tp, s);
  if (r) return r;
  if ((r = launch_transpose_bf16(xb, rows, K, ldx, w.tB, Tp, s))) return r;
  return launch_gemm_splitk(w.tA, Tp, w.tB, Tp, N, K, Tp, dw, w.partial, w.partial_floats, s);
}

// dW[N,K] += dY^T X on the token-major operands themselves (k_gemm_tn): dY bf16 [rows, ldy], X bf16 [rows, ldx]
// db (optional): the bias gradient db[n] += sum_rows dy[., n] as a by-product of the same pass over dy
int wgrad_tn(const bf16* dy_b, int ldy, int N, const bf16* xb, int ldx, int K, int rows, float* dw, const WgradScratch& w,
             hipStream_t s, float* db = nullptr) {
  return launch_gemm_tn_splitk(dy_b, ldy, xb, ldx, rows, N, K, dw, w.partial, w.partial_floats, s, db);
}

int check_lin_train(const swg_lin_train& t, bool bias) {
  if (!t.wt || !t.dw || (bias && !t.db)) return SWG_E_NULL;
  return 0;
}

int wattn_bwd_impl(const swg_block* blk, const swg_block_train* tr, int B, const float* x_in, const float* cond,
                   int cond_stride, const float* dy, float* dx, float* dcond, int dcond_stride, Carver& cv, hipStream_t s) {
  const int C = blk->C, L = blk->res * blk->res, T = B * L, Cp = kp(C), ldq = ldq_of(C), Q3p = kp(3 * C);
  const bool fused_tail = ln_cond_bwd_supported(C);
  float* x1 = fused_tail ? nullptr : cv.take<float>((size_t)T * C);
  bf16* xn = cv.take<bf16>((size_t)T * Cp);
  bf16* qkv = cv.take<bf16>((size_t)T * ldq);
  bf16* o = cv.take<bf16>((size_t)T * Cp);
  bf16* dyb = cv.take<bf16>((size_t)T * Cp);
  bf16* dO = cv.take<bf16>((size_t)T * Cp);
  bf16* dqkv = cv.take<bf16>((size_t)T * Q3p);
  float* dxn = cv.take<float>((size_t)T * C);
  float* dxc = fused_tail ? nullptr : cv.take<float>((size_t)T * C);
  WgradScratch w = take_wgrad(cv, 3 * C, C, T);
  if (cv.measure) return 0;
  if (!cv.ok) return SWG_E_WORKSPACE;
  int r;
  if ((r = check_lin_train(tr->qkv, true)) || (r = check_lin_train(tr->proj, true))) return r;
  if (!tr->ln1.dg || !tr->ln1.db || !tr->drpb) return SWG_E_NULL;
  // ---- recompute (:238-262) ----
  if ((r = launch_cond_ln(x_in, T, C, L, cond, cond_stride, blk->cond_off, blk->ln1.g, blk->ln1.b, x1, xn, Cp, s))) return r;
  GemmEpi e = {};
  e.bias = blk->qkv.b; e.out_bf16 = qkv; e.ldo = ldq; e.n_store = 3 * C;
  if ((r = launch_gemm(xn, Cp, (const bf16*)blk->qkv.w, T, 3 * C, Cp, e, s))) return r;
  if ((r = launch_wattn_core(qkv, ldq, B, blk->res, blk->win, blk->shift, blk->heads, C, blk->rpb, o, Cp, s))) return r;
  // ---- proj (:137): dO = dy Wproj, dWproj += dy^T o, dbproj += colsum(dy) ----
  if ((r = launch_cast_pad_colsum(dy, T, C, C, dyb, Cp, tr->proj.db, s))) return r;
  GemmEpi e1 = {};
  e1.out_bf16 = dO; e1.ldo = Cp; e1.n_store = Cp;
  if ((r = launch_gemm(dyb, Cp, (const bf16*)tr->proj.wt, T, C, Cp, e1, s))) return r;
  if ((r = wgrad_tn(dyb, Cp, C, o, Cp, C, T, tr->proj.dw, w, s))) return r;
  // ---- attention core ----
  if ((r = launch_wattn_bwd(qkv, ldq, dO, Cp, B, blk->res, blk->win, blk->shift, blk->heads, C, blk->rpb, dqkv, Q3p,
                            tr->drpb, s)))
    return r;
  // ---- qkv (:113) ----
  GemmEpi e2 = {};
  e2.out_f32 = dxn; e2.ldo = C;
  if ((r = launch_gemm(dqkv, Q3p, (const bf16*)tr->qkv.wt, T, C, Q3p, e2, s))) return r;
  if ((r = wgrad_tn(dqkv, Q3p, 3 * C, xn, Cp, C, T, tr->qkv.dw, w, s, tr->qkv.db))) return r;  // + qkv bias gradient
  // ---- LN1 + shortcut (:241-244, :272) and conditioning (:238-240) in one pass; the shortcut x1 is recomputed ----
  if (fused_tail)
    return launch_ln_cond_bwd(x_in, dxn, dy, blk->ln1.g, cond, cond_stride, blk->cond_off, B, L, C, dx, tr->ln1.dg,
                              tr->ln1.db, dcond, dcond_stride, s);
  // ---- LN1 + shortcut: dxc = dy + d LN1 ----
  if ((r = launch_ln_bwd(x1, dxn, blk->ln1.g, dy, T, C, dxc, tr->ln1.dg, tr->ln1.db, s))) return r;
  // ---- conditioning (:238-240) ----
  return launch_cond_bwd(x_in, dxc, cond, cond_stride, blk->cond_off, B, L, C, dx, dcond, dcond_stride, s);
}

int mlp_bwd_impl(const swg_block* blk, const swg_block_train* tr, int B, const float* x_in, const float* dy, float* dx,
                 Carver& cv, hipStream_t s) {
  const int C = blk->C, L = blk->res * blk->res, T = B * L, Cp = kp(C), H = 4 * C, Hp = kp(H);
  bf16* xn = cv.take<bf16>((size_t)T * Cp);
  bf16* h = cv.take<bf16>((size_t)T * Hp);
  bf16* a = cv.take<bf16>((size_t)T * Hp);
  bf16* dyb = cv.take<bf16>((size_t)T * Cp);
  bf16* da = cv.take<bf16>((size_t)T * Hp);
  float* dxn = cv.take<float>((size_t)T * C);
  // C = 384: one 192-token workgroup per CU streams the whole 2.4-MB weight image, which only pays with enough tokens
  // to fill the CUs (the unfused chain is faster below: grid B = 8 step 10.5 ms against 11.2 ms)
  const bool front = tr && tr->mlp_bwd_img && mlp_bwd_front_supported(C) && Cp == C && Hp == H && (C < 384 || T >= 16384);
  float* part = cv.take<float>(mlp_bwd_front_supported(C) ? mlp_bwd_front_part_floats(C, T) : 0);
  WgradScratch w = take_wgrad(cv, H, H, T);
  if (cv.measure) return 0;
  if (!cv.ok) return SWG_E_WORKSPACE;
  int r;
  if ((r = check_lin_train(tr->fc1, true)) || (r = check_lin_train(tr->fc2, true))) return r;
  if (!tr->ln2.dg || !tr->ln2.db) return SWG_E_NULL;
  if (front) {
    // ---- narrow stages: LayerNorm2, fc1, GELU, dy W2 and GELU' in one pass; h and da stay in registers; the bias
    // gradients come out as per-workgroup column sums, folded by two small column-sum launches ----
    if ((r = launch_mlp_bwd_front(C, T, x_in, dy, (const bf16*)tr->mlp_bwd_img, blk->fc1.b, blk->ln2.g, blk->ln2.b, xn, dyb,
                                  Cp, a, da, Hp, part, s)))
      return r;
    const int nwg = cdiv(T, mlp_bwd_front_tokens_per_wg(C));
    if ((r = launch_colsum_f32(part, nwg, H, 5 * C, tr->fc1.db, s))) return r;
    if ((r = launch_colsum_f32(part + H, nwg, C, 5 * C, tr->fc2.db, s))) return r;
  } else {
    // ---- recompute (:275, :19-25) ----
    if ((r = launch_cond_ln(x_in, T, C, L, nullptr, 0, 0, blk->ln2.g, blk->ln2.b, nullptr, xn, Cp, s))) return r;
    GemmEpi e = {};
    e.bias = blk->fc1.b; e.out_bf16 = h; e.ldo = Hp; e.n_store = Hp;
    if ((r = launch_gemm(xn, Cp, (const bf16*)blk->fc1.w, T, H, Cp, e, s))) return r;
    // ---- fc2: da = dy W2 ----
    if ((r = launch_cast_pad_colsum(dy, T, C, C, dyb, Cp, tr->fc2.db, s))) return r;
    GemmEpi e1 = {};
    e1.out_bf16 = da; e1.ldo = Hp; e1.n_store = Hp;
    if ((r = launch_gemm(dyb, Cp, (const bf16*)tr->fc2.wt, T, H, Cp, e1, s))) return r;
    // ---- GELU: one pass over h gives the activation a = gelu(h) (operand of dW2) and dh = da gelu'(h) (+ db1) ----
    if ((r = launch_gelu_bwd_colsum(h, da, da, a, T, H, Hp, tr->fc1.db, s))) return r;
  }
  if ((r = wgrad_tn(dyb, Cp, C, a, Hp, H, T, tr->fc2.dw, w, s))) return r;
  // ---- fc1 ----
  GemmEpi e2 = {};
  e2.out_f32 = dxn; e2.ldo = C;
  if ((r = launch_gemm(da, Hp, (const bf16*)tr->fc1.wt, T, C, Hp, e2, s))) return r;
  if ((r = wgrad_tn(da, Hp, H, xn, Cp, C, T, tr->fc1.dw, w, s))) return r;
  // ---- LN2 + residual ----
  return launch_ln_bwd(x_in, dxn, blk->ln2.g, dy, T, C, dx, tr->ln2.dg, tr->ln2.db, s);
}

int merge_bwd_impl(const swg_merge* m, const swg_merge_train* tr, int B, const float* x_in, const float* dy, float* dx,
                   Carver& cv, hipStream_t s) {
  const int C = m->C, K4 = 4 * C, N2 = 2 * C, T4 = B * (m->res / 2) * (m->res / 2), K4p = kp(K4), N2p = kp(N2);
  bf16* xn = cv.take<bf16>((size_t)T4 * K4p);
  bf16* dyb = cv.take<bf16>((size_t)T4 * N2p);
  float* dxn = cv.take<float>((size_t)T4 * K4);
  WgradScratch w = take_wgrad(cv, N2, K4, T4);
  if (cv.measure) return 0;
  if (!cv.ok) return SWG_E_WORKSPACE;
  int r;
  if ((r = check_lin_train(tr->red, false))) return r;
  if (!tr->ln.dg || !tr->ln.db) return SWG_E_NULL;
  // recompute: 2x2 gather + LayerNorm(4C) in one pass (the forward kernel); the gathered fp32 rows are never stored
  if ((r = launch_merge_ln(x_in, B, m->res, C, m->ln.g, m->ln.b, xn, K4p, s))) return r;
  if ((r = launch_cast_pad(dy, T4, N2, N2, dyb, N2p, s))) return r;
  GemmEpi e = {};
  e.out_f32 = dxn; e.ldo = K4;
  if ((r = launch_gemm(dyb, N2p, (const bf16*)tr->red.wt, T4, K4, N2p, e, s))) return r;
  if ((r = wgrad_tn(dyb, N2p, N2, xn, K4p, K4, T4, tr->red.dw, w, s))) return r;
  // LayerNorm backward with the gather (x) and the scatter (dx) folded into its addressing
  return launch_ln_bwd(x_in, dxn, m->ln.g, nullptr, T4, K4, dx, tr->ln.dg, tr->ln.db, s, m->res, 1 | 4);
}

int expand_bwd_impl(const swg_breakup* u, const swg_breakup_train* tr, int B, const float* x, const float* skip,
                    const float* dy, float* dx, float* dskip, Carver& cv, hipStream_t s) {
  const int Cin = u->Cin, Co = Cin / 4, T = B * u->res * u->res, T4 = 4 * T, Kp = kp(Cin), Cop = kp(Co);
  bf16* cat = cv.take<bf16>((size_t)T * Kp);
  float* y = cv.take<float>((size_t)T * Cin);
  float* z = cv.take<float>((size_t)T4 * Co);
  bf16* zn = cv.take<bf16>((size_t)T4 * Cop);
  bf16* dyb = cv.take<bf16>((size_t)T4 * Cop);
  float* dzn = cv.take<float>((size_t)T4 * Co);
  float* dyn = cv.take<float>((size_t)T * Cin);
  bf16* dyb2 = cv.take<bf16>((size_t)T * Kp);
  WgradScratch w = take_wgrad(cv, Cin, Cin, T4);  // Cin x round32(4T) covers both (Co, 4T) and (Cin, T)
  if (cv.measure) return 0;
  if (!cv.ok) return SWG_E_WORKSPACE;
  int r;
  if ((r = check_lin_train(tr->pre, false)) || (r = check_lin_train(tr->post, false))) return r;
  if (!tr->ln.dg || !tr->ln.db || !tr->post_ln.dg || !tr->post_ln.db) return SWG_E_NULL;
  // ---- recompute (:743, :374-403) ----
  if ((r = launch_concat_cast(x, skip, T, Cin / 2, cat, Kp, s))) return r;
  GemmEpi e = {};
  e.out_f32 = y; e.ldo = Cin;
  if ((r = launch_gemm(cat, Kp, (const bf16*)u->pre.w, T, Cin, Kp, e, s))) return r;
  // norm (Cin) + 2x2 scatter + post_norm (Co) in one pass over y (the forward kernel), which also leaves z
  if ((r = launch_breakup_ln2(y, B, u->res, Cin, u->ln.g, u->ln.b, u->post_ln.g, u->post_ln.b, zn, Cop, s, z))) return r;
  // ---- post_linear, post_norm ----
  if ((r = launch_cast_pad(dy, T4, Co, Co, dyb, Cop, s))) return r;
  GemmEpi e1 = {};
  e1.out_f32 = dzn; e1.ldo = Co;
  if ((r = launch_gemm(dyb, Cop, (const bf16*)tr->post.wt, T4, Co, Cop, e1, s))) return r;
  if ((r = wgrad_tn(dyb, Cop, Co, zn, Cop, Co, T4, tr->post.dw, w, s))) return r;
  if ((r = launch_ln_bwd(z, dzn, u->post_ln.g, nullptr, T4, Co, dzn, tr->post_ln.dg, tr->post_ln.db, s))) return r;
  // ---- scatter, norm, pre_linear ----
  // norm backward reads its upstream gradient through the 2x2 gather (dzn is in token layout)
  if (Kp == Cin) {  // ... and writes it as the bf16 GEMM operand directly (no fp32 copy, no cast pass)
    if ((r = launch_ln_bwd(y, dzn, u->ln.g, nullptr, T, Cin, nullptr, tr->ln.dg, tr->ln.db, s, 2 * u->res, 2, dyb2))) return r;
  } else {
    if ((r = launch_ln_bwd(y, dzn, u->ln.g, nullptr, T, Cin, dyn, tr->ln.dg, tr->ln.db, s, 2 * u->res, 2))) return r;
    if ((r = launch_cast_pad(dyn, T, Cin, Cin, dyb2, Kp, s))) return r;
  }
  const bf16* wt = (const bf16*)tr->pre.wt;  // [Cin (input feature), Kp]: rows [0, Cin/2) belong to x, the rest to skip
  GemmEpi e2 = {};
  e2.out_f32 = dx; e2.ldo = Cin / 2;
  if ((r = launch_gemm(dyb2, Kp, wt, T, Cin / 2, Kp, e2, s))) return r;
  e2.out_f32 = dskip;
  if ((r = launch_gemm(dyb2, Kp, wt + (size_t)(Cin / 2) * Kp, T, Cin / 2, Kp, e2, s))) return r;
  return wgrad_tn(dyb2, Kp, Cin, cat, Kp, Cin, T, tr->pre.dw, w, s);
}

int embed_bwd_impl(const swg_net* net, const swg_ends_train* tr, int B, const float* x, const float* sc, const float* c_in,
                   int c_in_stride, const float* cond, int cond_stride, const float* dtok, float* dcond, int dcond_stride,
                   Carver& cv, hipStream_t s) {
  const int hp = net->N / net->p, L = hp * hp, T = B * L, C0 = net->C0, kin = net->in_ch * net->p * net->p, Kp = kp(kin);
  bf16* P = cv.take<bf16>((size_t)T * Kp);
  float* u = cv.take<float>((size_t)T * C0);
  float* un = cv.take<float>((size_t)T * C0);
  float* dun = cv.take<float>((size_t)T * C0);
  bf16* dunb = cv.take<bf16>((size_t)T * C0);
  float* tmp = cv.take<float>((size_t)C0 * Kp);
  WgradScratch w = take_wgrad(cv, C0, Kp, T);
  if (cv.measure) return 0;
  if (!cv.ok) return SWG_E_WORKSPACE;
  if (!tr->pe_wb || !tr->dpe_w || !tr->dpe_b || !tr->pe_ln.dg || !tr->pe_ln.db) return SWG_E_NULL;
  int r;
  // ---- recompute (:566-574): conv as a GEMM over the p x p taps, LayerNorm ----
  if ((r = launch_embed_im2col(net, B, x, sc, c_in, c_in_stride, P, Kp, s))) return r;
  GemmEpi e = {};
  e.bias = net->pe_b; e.out_f32 = u; e.ldo = C0;
  if ((r = launch_gemm(P, Kp, (const bf16*)tr->pe_wb, T, C0, Kp, e, s))) return r;
  if ((r = launch_ln_f32(u, T, C0, net->pe_ln.g, net->pe_ln.b, un, s))) return r;
  // ---- conditioning, LayerNorm, conv weight / bias ----
  if ((r = launch_cond_bwd(un, dtok, cond, cond_stride, net->pe_cond_off, B, L, C0, dun, dcond, dcond_stride, s))) return r;
  // the LayerNorm backward writes its result as the bf16 token-major operand of the weight-gradient GEMM, which also
  // sums it over the tokens (conv bias gradient): no transposed copies, no 64-workgroup split-K GEMM, no column-sum pass
  // (row stride C0 must be a multiple of 8 for that GEMM: dim-60 nets keep the transposed-copy route)
  hipError_t he = hipMemsetAsync(tmp, 0, (size_t)C0 * Kp * sizeof(float), s);
  if (he != hipSuccess) return (int)he;
  if (C0 % 8 == 0) {
    if ((r = launch_ln_bwd(u, dun, net->pe_ln.g, nullptr, T, C0, nullptr, tr->pe_ln.dg, tr->pe_ln.db, s, 0, 0, dunb))) return r;
    if ((r = wgrad_tn(dunb, C0, C0, P, Kp, Kp, T, tmp, w, s, tr->dpe_b))) return r;
    return launch_add_slice(tmp, C0, kin, Kp, tr->dpe_w, s);
  }
  if ((r = launch_ln_bwd(u, dun, net->pe_ln.g, nullptr, T, C0, dun, tr->pe_ln.dg, tr->pe_ln.db, s))) return r;
  if ((r = wgrad(dun, nullptr, C0, C0, P, Kp, Kp, T, tmp, w, s))) return r;
  if ((r = launch_add_slice(tmp, C0, kin, Kp, tr->dpe_w, s))) return r;
  return launch_colsum_f32(dun, T, C0, C0, tr->dpe_b, s);
}

int readout_bwd_impl(const swg_net* net, const swg_ends_train* tr, int B, const float* tok, const float* flags,
                     const float* coef, const float* dD, float* dtok, Carver& cv, hipStream_t s) {
  const int hp = net->N / net->p, L = hp * hp, T = B * L, C0 = net->C0, pp = net->p * net->p, Kp = kp(pp), Cp = kp(C0);
  bf16* dpix = cv.take<bf16>((size_t)T * Kp);
  bf16* tn = cv.take<bf16>((size_t)T * Cp);
  float* dtn = cv.take<float>((size_t)T * C0);
  float* tmp = cv.take<float>((size_t)C0 * Kp);
  WgradScratch w = take_wgrad(cv, C0, Kp, T);
  if (cv.measure) return 0;
  if (!cv.ok) return SWG_E_WORKSPACE;
  if (!tr->ro_wb || !tr->dro_w || !tr->dro_b || !tr->norm.dg || !tr->norm.db) return SWG_E_NULL;
  if (Kp != 32) return SWG_E_UNSUPPORTED;
  int r;
  if ((r = launch_readout_dpix(dD, flags, coef, B, net->N, net->p, dpix, Kp, s))) return r;
  if ((r = launch_cond_ln(tok, T, C0, 1, nullptr, 0, 0, net->norm.g, net->norm.b, nullptr, tn, Cp, s))) return r;
  GemmEpi e = {};
  e.out_f32 = dtn; e.ldo = C0;
  if ((r = launch_gemm(dpix, Kp, (const bf16*)tr->ro_wb, T, C0, Kp, e, s))) return r;
  hipError_t he = hipMemsetAsync(tmp, 0, (size_t)C0 * Kp * sizeof(float), s);
  if (he != hipSuccess) return (int)he;
  if ((r = wgrad_tn(tn, Cp, C0, dpix, Kp, Kp, T, tmp, w, s))) return r;
  if ((r = launch_add_slice(tmp, C0, pp, Kp, tr->dro_w, s))) return r;
  if ((r = launch_colsum_bf16(dpix, T, Kp, Kp, tr->dro_b, s))) return r;
  return launch_ln_bwd(tok, dtn, net->norm.g, nullptr, T, C0, dtok, tr->norm.dg, tr->norm.db, s);
}

// ---- molecule path (NodeAdjSwinGNN head / tail, model/swin_gnn/swin_gnn.py:953-1018) -------------------------
int na_embed_bwd_impl(const swg_net* net, const swg_nodeadj* na, const swg_ends_train* tr, int B, const float* adj,
                      const float* node, const float* sc_adj, const float* sc_node, const float* flags, const float* c_in,
                      int c_in_stride, const float* cond, int cond_stride, const float* dtok, float* dcond,
                      int dcond_stride, Carver& cv, hipStream_t s) {
  const int hp = net->N / net->p, L = hp * hp, T = B * L, C0 = net->C0, kin = net->in_ch * net->p * net->p, Kp = kp(kin);
  bf16* P = cv.take<bf16>((size_t)T * Kp);
  float* u = cv.take<float>((size_t)T * C0);
  float* un = cv.take<float>((size_t)T * C0);
  float* dun = cv.take<float>((size_t)T * C0);
  bf16* dunb = cv.take<bf16>((size_t)T * C0);
  float* tmp = cv.take<float>((size_t)C0 * Kp);
  WgradScratch w = take_wgrad(cv, C0, Kp, T);
  if (cv.measure) return 0;
  if (!cv.ok) return SWG_E_WORKSPACE;
  if (!tr->pe_wb || !tr->dpe_w || !tr->dpe_b || !tr->pe_ln.dg || !tr->pe_ln.db) return SWG_E_NULL;
  int r;
  if ((r = launch_na_im2col(net, na, B, adj, node, sc_adj, sc_node, flags, c_in, c_in_stride, P, Kp, s))) return r;
  GemmEpi e = {};
  e.bias = net->pe_b; e.out_f32 = u; e.ldo = C0;
  if ((r = launch_gemm(P, Kp, (const bf16*)tr->pe_wb, T, C0, Kp, e, s))) return r;
  if ((r = launch_ln_f32(u, T, C0, net->pe_ln.g, net->pe_ln.b, un, s))) return r;
  if ((r = launch_cond_bwd(un, dtok, cond, cond_stride, net->pe_cond_off, B, L, C0, dun, dcond, dcond_stride, s))) return r;
  // the LayerNorm backward writes its result as the bf16 token-major operand of the weight-gradient GEMM, which also
  // sums it over the tokens (conv bias gradient): no transposed copies, no 64-workgroup split-K GEMM, no column-sum pass
  // (row stride C0 must be a multiple of 8 for that GEMM: dim-60 nets keep the transposed-copy route)
  hipError_t he = hipMemsetAsync(tmp, 0, (size_t)C0 * Kp * sizeof(float), s);
  if (he != hipSuccess) return (int)he;
  if (C0 % 8 == 0) {
    if ((r = launch_ln_bwd(u, dun, net->pe_ln.g, nullptr, T, C0, nullptr, tr->pe_ln.dg, tr->pe_ln.db, s, 0, 0, dunb))) return r;
    if ((r = wgrad_tn(dunb, C0, C0, P, Kp, Kp, T, tmp, w, s, tr->dpe_b))) return r;
    return launch_add_slice(tmp, C0, kin, Kp, tr->dpe_w, s);
  }
  if ((r = launch_ln_bwd(u, dun, net->pe_ln.g, nullptr, T, C0, dun, tr->pe_ln.dg, tr->pe_ln.db, s))) return r;
  if ((r = wgrad(dun, nullptr, C0, C0, P, Kp, Kp, T, tmp, w, s))) return r;
  if ((r = launch_add_slice(tmp, C0, kin, Kp, tr->dpe_w, s))) return r;
  return launch_colsum_f32(dun, T, C0, C0, tr->dpe_b, s);
}

// one read-out MLP head backwards: rows of `in` (bf16 [rows, Cp]) -> fc1 -> GELU -> fc2 (Co outputs, fp32 weights).
// dy fp32 [rows, Co]; writes din fp32 [rows, C0] (+= when `accumulate`), accumulates the four parameter gradients.
int na_head_bwd(const swg_lin& fc1, const swg_lin_train& t1, const float* w2, float* dw2, float* db2, int Co,
                const bf16* in, const float* dy, int rows, int C0, bf16* pre, bf16* act, float* din, bool accumulate,
                const WgradScratch& w, hipStream_t s) {
  const int Cp = kp(C0);
  int r;
  GemmEpi e = {};
  e.bias = fc1.b; e.out_bf16 = pre; e.ldo = Cp; e.n_store = Cp;
  if ((r = launch_gemm(in, Cp, (const bf16*)fc1.w, rows, C0, Cp, e, s))) return r;
  if ((r = launch_gelu_fwd(pre, act, (size_t)rows * Cp, s))) return r;
  // fc2: dW2 += dy^T act, db2 += colsum(dy), dact = dy W2
  if ((r = wgrad(dy, nullptr, Co, Co, act, Cp, C0, rows, dw2, w, s))) return r;
  if ((r = launch_colsum_f32(dy, rows, Co, Co, db2, s))) return r;
  if ((r = launch_na_head_bwd(dy, w2, rows, C0, Co, act, Cp, s))) return r;   // act now holds d(act)
  if ((r = launch_gelu_bwd_colsum(pre, act, act, nullptr, rows, C0, Cp, t1.db, s))) return r;   // ... and now d(pre); db1 too
  GemmEpi e2 = {};
  e2.out_f32 = din; e2.ldo = C0;
  if (accumulate) { e2.residual = din; e2.ldr = C0; }
  if ((r = launch_gemm(act, Cp, (const bf16*)t1.wt, rows, C0, Cp, e2, s))) return r;
  return wgrad_tn(act, Cp, C0, in, Cp, C0, rows, t1.dw, w, s);
}

int na_readout_bwd_impl(const swg_net* net, const swg_nodeadj* na, const swg_nodeadj_train* tr, int B, const float* tok,
                        const float* flags, const float* coef, const float* dDa, const float* dDn, float* dtok,
                        Carver& cv, hipStream_t s) {
  const int N = net->N, p = net->p, pp = p * p, hp = N / p, L = hp * hp, T = B * L, P = T * pp, C0 = net->C0, Cp = kp(C0);
  const int Ca = na ? na->Ca : 8, Cn = na ? na->Cn : 16, BN = B * N;
  bf16* xn = cv.take<bf16>((size_t)T * Cp);
  bf16* rep = cv.take<bf16>((size_t)P * Cp);
  bf16* pre = cv.take<bf16>((size_t)P * Cp);
  bf16* act = cv.take<bf16>((size_t)P * Cp);
  bf16* pooled = cv.take<bf16>((size_t)BN * Cp);
  float* dya = cv.take<float>((size_t)P * Ca);
  float* dyn = cv.take<float>((size_t)BN * Cn);
  float* dpooled = cv.take<float>((size_t)BN * C0);
  float* drep = cv.take<float>((size_t)P * C0);
  bf16* drepb = cv.take<bf16>((size_t)P * Cp);
  float* dxn = cv.take<float>((size_t)T * C0);
  // weight-gradient scratch shared by the heads (Co | C0 rows x P) and the read-out map (pp Cp rows x T)
  WgradScratch w;
  {
    const size_t cols = (size_t)round_up(P, 32) + (size_t)32 * pp;  // >= pp * round_up(T, 32)
    w.tA = cv.take<bf16>((size_t)Cp * cols);
    w.tB = cv.take<bf16>((size_t)Cp * cols);
    w.partial_floats = gemm_splitk_ws_floats(pp * Cp, Cp);
    w.partial = cv.take<float>(w.partial_floats);
  }
  if (cv.measure) return 0;
  if (!cv.ok) return SWG_E_WORKSPACE;
  int r;
  if ((r = check_lin_train(tr->adj_fc1, true)) || (r = check_lin_train(tr->node_fc1, true))) return r;
  if (!tr->ro_wt || !tr->dro_w || !tr->dro_b || !tr->dadj_fc2_w || !tr->dadj_fc2_b || !tr->dnode_fc2_w ||
      !tr->dnode_fc2_b || !tr->norm.dg || !tr->norm.db)
    return SWG_E_NULL;
  // ---- recompute (:953-956, :1006-1007) ----
  if ((r = launch_cond_ln(tok, T, C0, L, nullptr, 0, 0, net->norm.g, net->norm.b, nullptr, xn, Cp, s))) return r;
  GemmEpi e = {};
  e.bias = na->ro.b; e.out_bf16 = rep; e.ldo = pp * Cp; e.n_store = pp * Cp;
  if ((r = launch_gemm(xn, Cp, (const bf16*)na->ro.w, T, pp * Cp, Cp, e, s))) return r;
  if ((r = launch_na_pool(rep, Cp, B, N, p, flags, pooled, s))) return r;
  // ---- node head (:1009-1015, precond.py:103-105), then the pooling ----
  if ((r = launch_na_dnode(dDn, flags, coef, B, N, Cn, dyn, s))) return r;
  if ((r = na_head_bwd(na->node_fc1, tr->node_fc1, na->node_fc2_w, tr->dnode_fc2_w, tr->dnode_fc2_b, Cn, pooled, dyn, BN,
                       C0, pre, act, dpooled, false, w, s)))
    return r;
  if ((r = launch_na_pool_bwd(dpooled, flags, B, N, p, C0, drep, s))) return r;
  // ---- adjacency head (:1000-1003, :1016-1018, precond.py:102-104): accumulates into drep ----
  if ((r = launch_na_dadj(dDa, flags, coef, B, N, p, Ca, dya, s))) return r;
  if ((r = na_head_bwd(na->adj_fc1, tr->adj_fc1, na->adj_fc2_w, tr->dadj_fc2_w, tr->dadj_fc2_b, Ca, rep, dya, P, C0, pre,
                       act, drep, true, w, s)))
    return r;
  // ---- collapsed read-out: rep[t, (ij, c')] = xn[t] . Wro[(ij, c'), :] + b ----
  if ((r = launch_cast_pad(drep, P, C0, C0, drepb, Cp, s))) return r;
  GemmEpi e2 = {};
  e2.out_f32 = dxn; e2.ldo = C0;
  if ((r = launch_gemm(drepb, pp * Cp, (const bf16*)tr->ro_wt, T, C0, pp * Cp, e2, s))) return r;
  if ((r = wgrad_tn(drepb, pp * Cp, pp * Cp, xn, Cp, C0, T, tr->dro_w, w, s))) return r;
  if ((r = launch_colsum_bf16(drepb, T, pp * Cp, pp * Cp, tr->dro_b, s))) return r;
  return launch_ln_bwd(tok, dxn, net->norm.g, nullptr, T, C0, dtok, tr->norm.dg, tr->norm.db, s);
}

}  // namespace

extern "C" {

size_t swg_train_workspace_bytes(const swg_net* net, int32_t B) {
  if (!net || !net->layers || B <= 0 || net->p <= 0 || net->N % net->p) return 0;
  size_t need = 0;
  {
    Carver cv;
    embed_bwd_impl(net, nullptr, B, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr, nullptr, 0, cv, nullptr);
    need = smax(need, cv.used);
  }
  {
    Carver cv;
    readout_bwd_impl(net, nullptr, B, nullptr, nullptr, nullptr, nullptr, nullptr, cv, nullptr);
    need = smax(need, cv.used);
  }
  for (int li = 0; li < net->n_layers; ++li) {
    const swg_layer& ly = net->layers[li];
    if (ly.up) {
      Carver cv;
      expand_bwd_impl(ly.up, nullptr, B, nullptr, nullptr, nullptr, nullptr, nullptr, cv, nullptr);
      need = smax(need, cv.used);
    }
    for (int j = 0; j < ly.n_blocks; ++j) {
      Carver c1, c2;
      wattn_bwd_impl(&ly.blocks[j], nullptr, B, nullptr, nullptr, 0, nullptr, nullptr, nullptr, 0, c1, nullptr);
      mlp_bwd_impl(&ly.blocks[j], nullptr, B, nullptr, nullptr, nullptr, c2, nullptr);
      need = smax(need, smax(c1.used, c2.used));
    }
    if (ly.down) {
      Carver cv;
      merge_bwd_impl(ly.down, nullptr, B, nullptr, nullptr, nullptr, cv, nullptr);
      need = smax(need, cv.used);
    }
  }
  return need + 4096;
}

int swg_edm_coef(int32_t B, const float* labels, int32_t labels_uniform, int32_t mode, float* coef, float* c_noise,
                 void* stream) {
  if (!labels || !coef || !c_noise) return SWG_E_NULL;
  if (B <= 0 || (mode != 0 && mode != 1)) return SWG_E_SHAPE;
  return launch_edm_coef(labels, B, labels_uniform, mode, coef, c_noise, (hipStream_t)stream);
}

int swg_wattn_bwd(const swg_block* blk, const swg_block_train* tr, int32_t B, const float* x_in, const float* cond,
                  int32_t cond_stride, const float* dy, float* dx, float* dcond, int32_t dcond_stride, void* ws,
                  size_t ws_bytes, void* stream) {
  if (!blk || !tr || !x_in || !cond || !dy || !dx || !dcond) return SWG_E_NULL;
  if (B <= 0 || blk->C % 4 || blk->heads <= 0 || blk->C % blk->heads || blk->win <= 0 || blk->res % blk->win)
    return SWG_E_SHAPE;
  if (cond_stride <= 0) return SWG_E_UNSUPPORTED;  // training conditions every sample on its own sigma
  Carver cv(ws, ws_bytes);
  return wattn_bwd_impl(blk, tr, B, x_in, cond, cond_stride, dy, dx, dcond, dcond_stride, cv, (hipStream_t)stream);
}

int swg_ln_mlp_bwd(const swg_block* blk, const swg_block_train* tr, int32_t B, const float* x_in, const float* dy,
                   float* dx, void* ws, size_t ws_bytes, void* stream) {
  if (!blk || !tr || !x_in || !dy || !dx) return SWG_E_NULL;
  if (B <= 0 || blk->C % 4) return SWG_E_SHAPE;
  Carver cv(ws, ws_bytes);
  return mlp_bwd_impl(blk, tr, B, x_in, dy, dx, cv, (hipStream_t)stream);
}

int swg_patch_merge_bwd(const swg_merge* m, const swg_merge_train* tr, int32_t B, const float* x_in, const float* dy,
                        float* dx, void* ws, size_t ws_bytes, void* stream) {
  if (!m || !tr || !x_in || !dy || !dx) return SWG_E_NULL;
  if (B <= 0 || m->res % 2 || m->C % 4) return SWG_E_SHAPE;
  Carver cv(ws, ws_bytes);
  return merge_bwd_impl(m, tr, B, x_in, dy, dx, cv, (hipStream_t)stream);
}

int swg_patch_expand_bwd(const swg_breakup* u, const swg_breakup_train* tr, int32_t B, const float* x, const float* skip,
                         const float* dy, float* dx, float* dskip, void* ws, size_t ws_bytes, void* stream) {
  if (!u || !tr || !x || !skip || !dy || !dx || !dskip) return SWG_E_NULL;
  if (B <= 0 || u->Cin % 16) return SWG_E_SHAPE;
  Carver cv(ws, ws_bytes);
  return expand_bwd_impl(u, tr, B, x, skip, dy, dx, dskip, cv, (hipStream_t)stream);
}

int swg_patch_embed_bwd(const swg_net* net, const swg_ends_train* tr, int32_t B, const float* x, const float* self_cond,
                        const float* c_in, int32_t c_in_stride, const float* cond, int32_t cond_stride, const float* dtok,
                        float* dcond, int32_t dcond_stride, void* ws, size_t ws_bytes, void* stream) {
  if (!net || !tr || !x || !cond || !dtok || !dcond) return SWG_E_NULL;
  if (B <= 0 || net->C0 % 4 || net->in_ch * net->p * net->p > 32) return SWG_E_SHAPE;
  if (cond_stride <= 0) return SWG_E_UNSUPPORTED;
  Carver cv(ws, ws_bytes);
  return embed_bwd_impl(net, tr, B, x, self_cond, c_in, c_in_stride, cond, cond_stride, dtok, dcond, dcond_stride, cv,
                        (hipStream_t)stream);
}

int swg_readout_sym_precond_bwd(const swg_net* net, const swg_ends_train* tr, int32_t B, const float* tok,
                                const float* node_flags, const float* coef, const float* dD, float* dtok, void* ws,
                                size_t ws_bytes, void* stream) {
  if (!net || !tr || !tok || !node_flags || !dD || !dtok) return SWG_E_NULL;
  if (B <= 0 || net->C0 % 4) return SWG_E_SHAPE;
  Carver cv(ws, ws_bytes);
  return readout_bwd_impl(net, tr, B, tok, node_flags, coef, dD, dtok, cv, (hipStream_t)stream);
}

size_t swg_nodeadj_train_workspace_bytes(const swg_net* net, const swg_nodeadj* na, int32_t B) {
  size_t need = swg_train_workspace_bytes(net, B);
  if (!need || !na) return 0;
  {
    Carver cv;
    na_embed_bwd_impl(net, na, nullptr, B, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 0, nullptr,
                      nullptr, 0, cv, nullptr);
    need = smax(need, cv.used + 4096);
  }
  {
    Carver cv;
    na_readout_bwd_impl(net, na, nullptr, B, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, cv, nullptr);
    need = smax(need, cv.used + 4096);
  }
  return need;
}

int swg_nodeadj_embed_bwd(const swg_net* net, const swg_nodeadj* na, const swg_ends_train* tr, int32_t B,
                          const float* adj, const float* node, const float* sc_adj, const float* sc_node,
                          const float* node_flags, const float* c_in, int32_t c_in_stride, const float* cond,
                          int32_t cond_stride, const float* dtok, float* dcond, int32_t dcond_stride, void* ws,
                          size_t ws_bytes, void* stream) {
  if (!net || !na || !tr || !adj || !node || !node_flags || !cond || !dtok || !dcond) return SWG_E_NULL;
  if (B <= 0 || net->C0 % 4) return SWG_E_SHAPE;
  if (cond_stride <= 0) return SWG_E_UNSUPPORTED;
  Carver cv(ws, ws_bytes);
  return na_embed_bwd_impl(net, na, tr, B, adj, node, sc_adj, sc_node, node_flags, c_in, c_in_stride, cond, cond_stride,
                           dtok, dcond, dcond_stride, cv, (hipStream_t)stream);
}

int swg_nodeadj_readout_bwd(const swg_net* net, const swg_nodeadj* na, const swg_nodeadj_train* tr, int32_t B,
                            const float* tok, const float* node_flags, const float* coef, const float* d_adj,
                            const float* d_node, float* dtok, void* ws, size_t ws_bytes, void* stream) {
  if (!net || !na || !tr || !tok || !node_flags || !d_adj || !d_node || !dtok) return SWG_E_NULL;
  if (B <= 0 || net->C0 % 4 || net->C0 > 128 || na->Ca <= 0 || na->Ca > 8 || na->Cn <= 0 || na->Cn > 16) return SWG_E_SHAPE;
  Carver cv(ws, ws_bytes);
  return na_readout_bwd_impl(net, na, tr, B, tok, node_flags, coef, d_adj, d_node, dtok, cv, (hipStream_t)stream);
}

size_t swg_linear_wgrad_workspace_bytes(int32_t N, int32_t K) {
  if (N <= 0 || K <= 0) return 0;
  return gemm_splitk_ws_floats(N, K) * sizeof(float) + 256;
}
int swg_linear_wgrad_bf16(int32_t rows, int32_t N, int32_t K, const void* dy, int32_t ldy, const void* x, int32_t ldx,
                          float* dw, void* ws, size_t ws_bytes, void* stream) {
  if (!dy || !x || !dw) return SWG_E_NULL;
  if (rows <= 0 || N <= 0 || K <= 0 || ldy < N || ldx < K) return SWG_E_SHAPE;
  Carver cv(ws, ws_bytes);
  const size_t pf = gemm_splitk_ws_floats(N, K);
  float* partial = cv.take<float>(pf);
  if (!cv.ok) return SWG_E_WORKSPACE;
  return launch_gemm_tn_splitk((const bf16*)dy, ldy, (const bf16*)x, ldx, rows, N, K, dw, partial, pf, (hipStream_t)stream);
}

}  // extern "C"
