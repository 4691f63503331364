// extern "C" entry points of libswingnn_hip.so (declared in include/swingnn_hip.h) and the host-side
// sequencing of one SwinGNN pass (model/swin_gnn/swin_gnn.py:729-779).  Nothing here allocates or
// synchronises; every buffer is carved out of the caller's workspace.
#include "common.h"

namespace {

struct Carver {
  char* p;
  size_t left;
  bool ok;
  Carver(void* ws, size_t bytes) : p((char*)ws), left(bytes), ok(ws != nullptr) {}
  template <typename T>
  T* take(size_t n) {
    const size_t b = align256(n * sizeof(T));
    if (!ok || b > left) { ok = false; return nullptr; }
    T* r = (T*)p;
    p += b;
    left -= b;
    return r;
  }
};

inline int kp(int k) { return round_up(k, 32); }
inline int ldq_of(int C) { return round_up(3 * C, 8); }

// ---- scratch needs of each op (bytes) ----
size_t ws_wattn(int T, int C) {
  return align256((size_t)T * C * 4) + 2 * align256((size_t)T * kp(C) * 2) + align256((size_t)T * ldq_of(C) * 2);
}
size_t ws_mlp(int T, int C) { return align256((size_t)T * kp(C) * 2) + align256((size_t)T * kp(4 * C) * 2); }
size_t ws_merge(int T, int C) { return align256((size_t)(T / 4) * kp(4 * C) * 2); }
size_t ws_expand(int T, int Cin) {
  return align256((size_t)T * kp(Cin) * 2) + align256((size_t)T * Cin * 4) + align256((size_t)4 * T * kp(Cin / 4) * 2);
}
size_t ws_embed(int Bc, int C0) { return align256((size_t)Bc * C0 * 4) + 2 * align256((size_t)Bc * SWG_NOISE_EMB * 4); }
size_t ws_readout(int B, int N) { return align256((size_t)B * N * N * 4); }

int check_lin(const swg_lin& l, int N, int K) {
  if (!l.w) return SWG_E_NULL;
  if (l.N != N || l.K != K || l.Kp != kp(K)) return SWG_E_SHAPE;
  return 0;
}

int check_block(const swg_block* b) {
  if (!b) return SWG_E_NULL;
  if (b->C <= 0 || b->C % 4 || b->heads <= 0 || b->C % b->heads || b->win <= 0 || b->res % b->win) return SWG_E_SHAPE;
  if (b->shift < 0 || b->shift >= b->win) return SWG_E_SHAPE;
  if (!b->ln1.g || !b->ln1.b || !b->ln2.g || !b->ln2.b || !b->rpb) return SWG_E_NULL;
  int r;
  if ((r = check_lin(b->qkv, 3 * b->C, b->C))) return r;
  if ((r = check_lin(b->proj, b->C, b->C))) return r;
  if ((r = check_lin(b->fc1, 4 * b->C, b->C))) return r;
  if ((r = check_lin(b->fc2, b->C, 4 * b->C))) return r;
  return 0;
}

int wattn_impl(const swg_block* blk, int B, const float* x_in, const float* cond, int cond_stride, float* x_out,
               void* ws, size_t ws_bytes, hipStream_t s) {
  int r = check_block(blk);
  if (r) return r;
  if (!x_in || !x_out || !cond) return SWG_E_NULL;
  const int C = blk->C, L = blk->res * blk->res, T = B * L, Cp = kp(C), ldq = ldq_of(C);
  if (blk->attn_img && wattn_res_supported(C, blk->heads, blk->win, blk->res))
    return launch_wattn_res(blk, B, x_in, cond, cond_stride, x_out, s);
  if (blk->attn_img && wattn_ring_supported(C, blk->heads, blk->win, blk->res)) {
    Carver cv(ws, ws_bytes);
    bf16* xn = nullptr;
    if (wattn_ring_wants_xn(C)) {  // xn = LN1(silu(shift + x(1+scale))) in bf16 (:238-244)
      xn = cv.take<bf16>((size_t)T * Cp);
      if (!cv.ok) return SWG_E_WORKSPACE;
      if ((r = launch_cond_ln(x_in, T, C, L, cond, cond_stride, blk->cond_off, blk->ln1.g, blk->ln1.b, nullptr, xn, Cp, s)))
        return r;
    }
    if (wattn_ring_fuses_proj(C))
      return launch_wattn_ring(blk, B, x_in, xn, Cp, cond, cond_stride, x_out, nullptr, 0, s);
    // two kernels: attention core -> o (bf16, head-dim slots in accumulator order), then the projection GEMM on the
    // slot-permuted Wproj (stored behind the fragment stream) with the shortcut recomputed in its epilogue.
    // head_dim 20: o has a 32-slot group per head (CO = heads * 32 columns, slots 20..31 zero) and Wproj matching
    // zero columns, behind a stream of 3 * CO rows of Cp (= C rounded up to 32) channels.
    const int CO = blk->heads * 32;
    bf16* o = cv.take<bf16>((size_t)T * CO);
    if (!cv.ok) return SWG_E_WORKSPACE;
    if ((r = launch_wattn_ring(blk, B, x_in, xn, Cp, cond, cond_stride, nullptr, o, CO, s))) return r;
    const bf16* wp = (const bf16*)blk->attn_img + (size_t)3 * CO * Cp;
    GemmEpi e2 = {};
    e2.bias = blk->proj.b; e2.residual = x_in; e2.ldr = C; e2.out_f32 = x_out; e2.ldo = C;
    e2.cond = cond; e2.cond_stride = cond_stride; e2.cond_off = blk->cond_off; e2.cond_L = L;
    return launch_gemm(o, CO, wp, T, C, CO, e2, s);
  }
  Carver cv(ws, ws_bytes);
  float* x1 = cv.take<float>((size_t)T * C);
  bf16* xn = cv.take<bf16>((size_t)T * Cp);
  bf16* o = cv.take<bf16>((size_t)T * Cp);
  bf16* qkv = cv.take<bf16>((size_t)T * ldq);
  if (!cv.ok) return SWG_E_WORKSPACE;
  // x1 = silu(shift + x(1+scale)) (the shortcut, :240-242); xn = LN1(x1) in bf16
  if ((r = launch_cond_ln(x_in, T, C, L, cond, cond_stride, blk->cond_off, blk->ln1.g, blk->ln1.b, x1, xn, Cp, s))) return r;
  GemmEpi e = {};
  e.bias = blk->qkv.b; e.out_bf16 = qkv; e.ldo = ldq; e.n_store = 3 * C;
  if ((r = launch_gemm(xn, Cp, (const bf16*)blk->qkv.w, T, 3 * C, Cp, e, s))) return r;
  if ((r = launch_wattn_core(qkv, ldq, B, blk->res, blk->win, blk->shift, blk->heads, C, blk->rpb, o, Cp, s))) return r;
  GemmEpi e2 = {};
  e2.bias = blk->proj.b; e2.residual = x1; e2.ldr = C; e2.out_f32 = x_out; e2.ldo = C;
  return launch_gemm(o, Cp, (const bf16*)blk->proj.w, T, C, Cp, e2, s);
}

// One SwinTransformerBlock (:232-277).  Where the attention kernel leaves the projection to a second kernel (C = 384)
// and the block carries the combined image, the projection, the shortcut and the whole MLP half run as ONE kernel
// behind the attention core; otherwise the two halves run one after the other.  x_in == x_out is allowed.
int mlp_impl(const swg_block* blk, int B, const float* x_in, float* x_out, void* ws, size_t ws_bytes, hipStream_t s);
// Token counts from which the one-pass kernels (one workgroup per CU streaming a whole weight) beat the GEMM route.
// Measured on the grid geometry (profiles/r03_batch_sweep*.jsonl): at 28 graphs (16128 tokens at C = 384) the one-pass
// kernels win by 3.5 %, at 14 graphs (8064 tokens) the GEMM route wins by 6 %: the gate sits between, at 12288.
// SWG_FUSED_MIN_TOKENS_PCT rescales all of them (tuning knob of scripts/, read once; 100 = the measured gates).
constexpr size_t kFusedGate = 12288;
static size_t fused_gate(size_t tokens) {
  static const long pct = [] {
    const char* e = getenv("SWG_FUSED_MIN_TOKENS_PCT");
    const long v = e ? atol(e) : 100;
    return v < 0 ? 100 : v;
  }();
  return tokens * (size_t)pct / 100;
}
bool fused_tail_ok(const swg_block* blk, int B) {
  const int C = blk->C, L = blk->res * blk->res;
  return blk->proj_mlp_img && blk->attn_img && mlp_stream_proj_supported(C) &&
         (size_t)B * L >= fused_gate(kFusedGate) && wattn_ring_supported(C, blk->heads, blk->win, blk->res) &&
         !wattn_ring_fuses_proj(C);
}
// head_dim-20 blocks (C = 60 / 120): attention core (k_wattn_ring<hd20>) + k_mlp_fused PROJ form; no token gate (both
// kernels take 64 / 128 tokens per workgroup)
static bool fused_tail_hd20_ok(const swg_block* blk) {
  const int C = blk->C;
  return blk->proj_mlp_img && blk->attn_img && blk->mlp_img && mlp_fused_proj_supported(C) && blk->heads * 20 == C &&
         wattn_ring_supported(C, blk->heads, blk->win, blk->res);
}
static bool mlp_stream_ok(const swg_block* blk, int B) {
  const int C = blk->C;
  const size_t T = (size_t)B * blk->res * blk->res;
  return blk->mlp_img && !mlp_fused_supported(C) && mlp_stream_supported(C) && T >= fused_gate(C == 384 ? kFusedGate : 8192);
}
static bool merge_fused_ok(const swg_merge* m, int B) {
  const size_t T4 = (size_t)B * (m->res / 2) * (m->res / 2);
  return m->img && merge_fused_supported(m->C, m->res) && T4 >= fused_gate(kFusedGate);
}
static bool expand_fused_ok(const swg_breakup* u, int B) {
  const size_t T = (size_t)B * u->res * u->res;
  return u->img_pre && u->img_post && expand_fused_supported(u->Cin) && T >= fused_gate(kFusedGate);
}
int block_impl(const swg_block* blk, int B, const float* x_in, const float* cond, int cond_stride, float* x_out, void* ws,
               size_t ws_bytes, hipStream_t s) {
  int r = check_block(blk);
  if (r) return r;
  if (!x_in || !x_out || !cond) return SWG_E_NULL;
  const int C = blk->C, L = blk->res * blk->res, T = B * L, Cp = kp(C);
  if (fused_tail_ok(blk, B)) {
    Carver cv(ws, ws_bytes);
    bf16* xn = cv.take<bf16>((size_t)T * Cp);
    bf16* o = cv.take<bf16>((size_t)T * Cp);
    if (!cv.ok) return SWG_E_WORKSPACE;
    if (!wattn_ring_wants_xn(C)) xn = nullptr;  // conditioning + LayerNorm1 run in the attention kernel's prologue
    else if ((r = launch_cond_ln(x_in, T, C, L, cond, cond_stride, blk->cond_off, blk->ln1.g, blk->ln1.b, nullptr, xn, Cp, s)))
      return r;
    if ((r = launch_wattn_ring(blk, B, x_in, xn, Cp, cond, cond_stride, nullptr, o, Cp, s))) return r;
    return launch_mlp_stream_proj(C, T, L, x_in, o, Cp, cond, cond_stride, blk->cond_off, blk->proj.b, x_out,
                                  (const bf16*)blk->proj_mlp_img, blk->fc1.b, blk->fc2.b, blk->ln2.g, blk->ln2.b, s);
  }
  if (fused_tail_hd20_ok(blk)) {
    const int CO = blk->heads * 32;
    Carver cv(ws, ws_bytes);
    bf16* xn = cv.take<bf16>((size_t)T * Cp);
    bf16* o = cv.take<bf16>((size_t)T * CO);
    if (!cv.ok) return SWG_E_WORKSPACE;
    if ((r = launch_cond_ln(x_in, T, C, L, cond, cond_stride, blk->cond_off, blk->ln1.g, blk->ln1.b, nullptr, xn, Cp, s)))
      return r;
    if ((r = launch_wattn_ring(blk, B, x_in, xn, Cp, cond, cond_stride, nullptr, o, CO, s))) return r;
    return launch_mlp_fused_proj(C, T, L, x_in, o, CO, cond, cond_stride, blk->cond_off, blk->proj.b, x_out,
                                 (const bf16*)blk->proj_mlp_img, (const bf16*)blk->mlp_img, blk->fc1.b, blk->fc2.b,
                                 blk->ln2.g, blk->ln2.b, s);
  }
  if ((r = wattn_impl(blk, B, x_in, cond, cond_stride, x_out, ws, ws_bytes, s))) return r;
  return mlp_impl(blk, B, x_out, x_out, ws, ws_bytes, s);
}

int mlp_impl(const swg_block* blk, int B, const float* x_in, float* x_out, void* ws, size_t ws_bytes, hipStream_t s) {
  int r = check_block(blk);
  if (r) return r;
  if (!x_in || !x_out) return SWG_E_NULL;
  const int C = blk->C, L = blk->res * blk->res, T = B * L, Cp = kp(C), Hp = kp(4 * C);
  if (blk->mlp_img && mlp_fused_supported(C))
    return launch_mlp_fused(C, T, x_in, x_out, (const bf16*)blk->mlp_img, blk->fc1.b, blk->fc2.b, blk->ln2.g,
                            blk->ln2.b, s);
  // the streamed kernel pays a fixed weight-stream latency per workgroup (~90 us at C = 384, ~35 us at C = 192): below
  // these token counts the three-kernel GEMM path is faster (measured: community-small, 4 / 16 tokens per graph)
  if (mlp_stream_ok(blk, B))
    return launch_mlp_stream(C, T, x_in, x_out, (const bf16*)blk->mlp_img, blk->fc1.b, blk->fc2.b, blk->ln2.g,
                             blk->ln2.b, s);
  Carver cv(ws, ws_bytes);
  bf16* xn = cv.take<bf16>((size_t)T * Cp);
  bf16* hbuf = cv.take<bf16>((size_t)T * Hp);
  if (!cv.ok) return SWG_E_WORKSPACE;
  if ((r = launch_cond_ln(x_in, T, C, L, nullptr, 0, 0, blk->ln2.g, blk->ln2.b, nullptr, xn, Cp, s))) return r;
  GemmEpi e = {};
  e.bias = blk->fc1.b; e.act = 1; e.out_bf16 = hbuf; e.ldo = Hp; e.n_store = Hp;
  if ((r = launch_gemm(xn, Cp, (const bf16*)blk->fc1.w, T, 4 * C, Cp, e, s))) return r;
  GemmEpi e2 = {};
  e2.bias = blk->fc2.b; e2.residual = x_in; e2.ldr = C; e2.out_f32 = x_out; e2.ldo = C;
  return launch_gemm(hbuf, Hp, (const bf16*)blk->fc2.w, T, C, Hp, e2, s);
}

int merge_impl(const swg_merge* m, int B, const float* x_in, float* x_out, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!m || !x_in || !x_out || !m->ln.g || !m->ln.b) return SWG_E_NULL;
  if (m->res % 2 || m->C % 4) return SWG_E_SHAPE;
  int r;
  if ((r = check_lin(m->red, 2 * m->C, 4 * m->C))) return r;
  const int T4 = B * (m->res / 2) * (m->res / 2), Kp = kp(4 * m->C);
  // one pass for the narrow stages; a workgroup takes 128 merged tokens and streams the whole weight, so small token
  // counts stay on the GEMM
  // (not in place: a workgroup's output rows [T4, 2C] overlap other workgroups' input rows)
  if (merge_fused_ok(m, B) && (const float*)x_out != x_in)
    return launch_merge_fused(m->C, B, m->res, x_in, (const bf16*)m->img, m->ln.g, m->ln.b, x_out, s);
  Carver cv(ws, ws_bytes);
  bf16* xn = cv.take<bf16>((size_t)T4 * Kp);
  if (!cv.ok) return SWG_E_WORKSPACE;
  if ((r = launch_merge_ln(x_in, B, m->res, m->C, m->ln.g, m->ln.b, xn, Kp, s))) return r;
  GemmEpi e = {};
  e.out_f32 = x_out; e.ldo = 2 * m->C;
  return launch_gemm(xn, Kp, (const bf16*)m->red.w, T4, 2 * m->C, Kp, e, s);
}

int expand_impl(const swg_breakup* u, int B, const float* x, const float* skip, float* x_out, void* ws, size_t ws_bytes,
                hipStream_t s) {
  if (!u || !x || !skip || !x_out || !u->ln.g || !u->ln.b || !u->post_ln.g || !u->post_ln.b) return SWG_E_NULL;
  if (u->Cin % 16) return SWG_E_SHAPE;
  const int Cin = u->Cin, Co = Cin / 4, T = B * u->res * u->res, Kp = kp(Cin), Cop = kp(Co);
  int r;
  if ((r = check_lin(u->pre, Cin, Cin))) return r;
  if ((r = check_lin(u->post, Co, Co))) return r;
  // one pass over x and skip for the last decoder level; a workgroup takes 128 tokens and streams the whole
  // pre_linear weight, so small token counts stay on the GEMM
  // (not in place: a workgroup's output rows are other workgroups' input rows)
  if (expand_fused_ok(u, B) && x_out != x && x_out != skip)
    return launch_expand_fused(Cin, B, u->res, x, skip, (const bf16*)u->img_pre, (const bf16*)u->img_post, u->ln.g,
                               u->ln.b, u->post_ln.g, u->post_ln.b, x_out, s);
  Carver cv(ws, ws_bytes);
  bf16* cat = cv.take<bf16>((size_t)T * Kp);
  float* y = cv.take<float>((size_t)T * Cin);
  bf16* sc = cv.take<bf16>((size_t)4 * T * Cop);
  if (!cv.ok) return SWG_E_WORKSPACE;
  if ((r = launch_concat_cast(x, skip, T, Cin / 2, cat, Kp, s))) return r;  // torch.cat([x, skip], -1), :743
  GemmEpi e = {};
  e.out_f32 = y; e.ldo = Cin;
  if ((r = launch_gemm(cat, Kp, (const bf16*)u->pre.w, T, Cin, Kp, e, s))) return r;
  if (breakup_post_supported(Co))  // ... and post_linear too, when its weight fits the register file
    return launch_breakup_post(y, B, u->res, Co, u->ln.g, u->ln.b, u->post_ln.g, u->post_ln.b, (const bf16*)u->post.w,
                               u->post.Kp, x_out, s);
  // norm (over Cin) + 2x2 scatter + post_norm (over Co) in one pass over y
  if ((r = launch_breakup_ln2(y, B, u->res, Cin, u->ln.g, u->ln.b, u->post_ln.g, u->post_ln.b, sc, Cop, s))) return r;
  GemmEpi e2 = {};
  e2.out_f32 = x_out; e2.ldo = Co;
  return launch_gemm(sc, Cop, (const bf16*)u->post.w, 4 * T, Co, Cop, e2, s);
}

int embed_impl(const swg_net* net, int Bc, const float* c_noise, float* cond, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!net || !c_noise || !cond || !net->freqs || !net->map0_w || !net->map1_w || !net->aff_w) return SWG_E_NULL;
  if (net->C0 % 4) return SWG_E_UNSUPPORTED;
  Carver cv(ws, ws_bytes);
  float* e0 = cv.take<float>((size_t)Bc * net->C0);
  float* e1 = cv.take<float>((size_t)Bc * SWG_NOISE_EMB);
  float* e2 = cv.take<float>((size_t)Bc * SWG_NOISE_EMB);
  if (!cv.ok) return SWG_E_WORKSPACE;
  int r;
  if ((r = launch_pos_embed(c_noise, Bc, net->freqs, net->C0 / 2, e0, s))) return r;
  if ((r = launch_gemv_rows(net->map0_w, net->map0_b, e0, Bc, SWG_NOISE_EMB, net->C0, 1, e1, s))) return r;
  if ((r = launch_gemv_rows(net->map1_w, net->map1_b, e1, Bc, SWG_NOISE_EMB, SWG_NOISE_EMB, 1, e2, s))) return r;
  return launch_gemv_rows(net->aff_w, net->aff_b, e2, Bc, net->cond_total, SWG_NOISE_EMB, 0, cond, s);
}

int readout_impl(const swg_net* net, int B, const float* tok, const float* x, const float* flags, const float* coef,
                 float* out, float* sym_err, void* ws, size_t ws_bytes, hipStream_t s) {
  if (!net || !tok || !flags || !out || !net->ro_w || !net->norm.g || !net->norm.b) return SWG_E_NULL;
  Carver cv(ws, ws_bytes);
  float* F = cv.take<float>((size_t)B * net->N * net->N);
  if (!cv.ok) return SWG_E_WORKSPACE;
  int r;
  if ((r = launch_final_readout(net, B, tok, F, s))) return r;
  return launch_sym_precond(B, net->N, F, x, flags, coef, out, sym_err, s);
}

// maximum scratch over all ops of one pass + persistent buffers
struct NetPlan {
  size_t persistent, scratch;
  size_t maxTC;  // floats of the widest residual-stream tensor
};

int plan_net(const swg_net* net, int B, NetPlan* pl) {
  if (!net || !net->layers) return SWG_E_NULL;
  if (net->N <= 0 || net->p <= 0 || net->N % net->p || net->n_layers <= 0 || net->n_layers % 2) return SWG_E_SHAPE;
  const int hp = net->N / net->p;
  size_t scratch = ws_embed(B, net->C0);
  scratch = smax(scratch, ws_readout(B, net->N));
  size_t maxTC = (size_t)B * hp * hp * net->C0;
  size_t skips = 0;
  const int S = net->n_layers / 2;
  for (int li = 0; li < net->n_layers; ++li) {
    const swg_layer& ly = net->layers[li];
    if (ly.up) {
      const int T = B * ly.up->res * ly.up->res;
      scratch = smax(scratch, ws_expand(T, ly.up->Cin));
      maxTC = smax(maxTC, (size_t)4 * T * (ly.up->Cin / 4));
    }
    for (int j = 0; j < ly.n_blocks; ++j) {
      const swg_block& b = ly.blocks[j];
      const int T = B * b.res * b.res;
      scratch = smax(scratch, smax(ws_wattn(T, b.C), ws_mlp(T, b.C)));
      maxTC = smax(maxTC, (size_t)T * b.C);
    }
    if (ly.down) {
      const int T = B * ly.down->res * ly.down->res;
      scratch = smax(scratch, ws_merge(T, ly.down->C));
      if (li < S) skips += align256((size_t)(T / 4) * 2 * ly.down->C * 4);
    }
  }
  pl->maxTC = maxTC;
  pl->scratch = scratch;
  pl->persistent = align256((size_t)B * 4 * 4) + align256((size_t)B * 4) + align256((size_t)B * net->cond_total * 4) +
                   2 * align256(maxTC * 4) + skips;
  return 0;
}

// persistent buffers of one pass, carved from the caller's workspace; (sw, sb) is the scratch left for the ops
struct TrunkBufs {
  float *coef, *cnoise, *cond, *A;
  float* A2;  // second token tensor: PatchBreakup's one-pass kernel cannot run in place
  float* skips[8];
  int Bc, cstride, S;
  void* sw;
  size_t sb;
};

int carve_trunk(const swg_net* net, int B, int labels_uniform, void* ws, size_t ws_bytes, size_t extra_scratch,
                TrunkBufs* tb) {
  NetPlan pl;
  int r = plan_net(net, B, &pl);
  if (r) return r;
  if (ws_bytes < pl.persistent + smax(pl.scratch, extra_scratch)) return SWG_E_WORKSPACE;
  const int S = net->n_layers / 2;
  if (S > 8) return SWG_E_UNSUPPORTED;
  Carver cv(ws, ws_bytes);
  tb->coef = cv.take<float>((size_t)B * 4);
  tb->cnoise = cv.take<float>((size_t)B);
  tb->Bc = labels_uniform ? 1 : B;
  tb->cstride = labels_uniform ? 0 : net->cond_total;
  tb->cond = cv.take<float>((size_t)B * net->cond_total);
  tb->A = cv.take<float>(pl.maxTC);
  tb->A2 = cv.take<float>(pl.maxTC);
  tb->S = S;
  for (int li = 0; li < 8; ++li) tb->skips[li] = nullptr;
  for (int li = 0; li < S; ++li) {
    const swg_layer& ly = net->layers[li];
    if (ly.down) {
      const int T = B * ly.down->res * ly.down->res;
      tb->skips[li] = cv.take<float>((size_t)(T / 4) * 2 * ly.down->C);
    }
  }
  if (!cv.ok) return SWG_E_WORKSPACE;
  tb->sw = cv.p;
  tb->sb = cv.left;
  return 0;
}

// encoder / decoder layers over the token tensor in tb.A (model/swin_gnn/swin_gnn.py:735-746); *out = final tokens
int run_trunk(const swg_net* net, int B, const TrunkBufs& tb, const float** out, hipStream_t s) {
  int r;
  float* A = tb.A;
  const int S = tb.S;
  const float* cur = A;
  for (int li = 0; li < net->n_layers; ++li) {
    const swg_layer& ly = net->layers[li];
    if (ly.up) {
      const int si = net->n_layers - 1 - li;  // decoder layer i pops skips[S-1-i], :743
      if (si < 0 || si >= S || !tb.skips[si]) return SWG_E_SHAPE;
      float* dst = (cur == A) ? tb.A2 : A;
      if ((r = expand_impl(ly.up, B, cur, tb.skips[si], dst, tb.sw, tb.sb, s))) return r;
      cur = dst;
    }
    for (int j = 0; j < ly.n_blocks; ++j) {
      const swg_block* blk = &ly.blocks[j];
      if ((r = block_impl(blk, B, cur, tb.cond, tb.cstride, A, tb.sw, tb.sb, s))) return r;
      cur = A;
    }
    if (ly.down) {
      if (li >= S) return SWG_E_SHAPE;
      if ((r = merge_impl(ly.down, B, cur, tb.skips[li], tb.sw, tb.sb, s))) return r;
      cur = tb.skips[li];
    }
  }
  *out = cur;
  return 0;
}

// scratch of the NodeAdj read-out (swg_nodeadj_forward)
size_t ws_na_tail(const swg_net* net, const swg_nodeadj* na, int B) {
  const int Cp = kp(net->C0), hp = net->N / net->p;
  const size_t T = (size_t)B * hp * hp, P = (size_t)B * net->N * net->N;
  return align256(T * Cp * 2) + 2 * align256(P * Cp * 2) + align256(P * na->Ca * 4) +
         2 * align256((size_t)B * net->N * Cp * 2);
}

// NodeAdjSwinGNN read-out (:953-956), the two MLP heads (:1000-1012), masks / symmetrisation (:1015-1018) and the
// preconditioning epilogue (precond.py:102-105) on the final tokens `cur`
int na_readout_impl(const swg_net* net, const swg_nodeadj* na, int B, int mode, const float* cur, const float* adj,
                    const float* node, const float* node_flags, const float* coef, float* out_adj, float* out_node,
                    float* sym_err, void* ws, size_t ws_bytes, hipStream_t s) {
  const int C0 = net->C0, Cp = kp(C0), p = net->p, pp = p * p, N = net->N;
  int r;
  const int hp = N / p, T = B * hp * hp, P = T * pp;  // P = B N N pixels, rows in token-patch order
  Carver cv(ws, ws_bytes);
  bf16* xn = cv.take<bf16>((size_t)T * Cp);
  bf16* rep = cv.take<bf16>((size_t)P * Cp);
  bf16* hid = cv.take<bf16>((size_t)P * Cp);
  float* F = cv.take<float>((size_t)B * na->Ca * N * N);
  bf16* pooled = cv.take<bf16>((size_t)B * N * Cp);
  bf16* hidn = cv.take<bf16>((size_t)B * N * Cp);
  if (!cv.ok) return SWG_E_WORKSPACE;
  if ((r = launch_cond_ln(cur, T, C0, hp * hp, nullptr, 0, 0, net->norm.g, net->norm.b, nullptr, xn, Cp, s))) return r;
  GemmEpi e = {};
  e.bias = na->ro.b; e.out_bf16 = rep; e.ldo = pp * Cp; e.n_store = pp * Cp;
  if ((r = launch_gemm(xn, Cp, (const bf16*)na->ro.w, T, pp * Cp, Cp, e, s))) return r;
  GemmEpi e1 = {};
  e1.bias = na->adj_fc1.b; e1.act = 1; e1.out_bf16 = hid; e1.ldo = Cp; e1.n_store = Cp;
  if ((r = launch_gemm(rep, Cp, (const bf16*)na->adj_fc1.w, P, C0, Cp, e1, s))) return r;
  if ((r = launch_na_head_adj(hid, Cp, B, N, p, C0, na->Ca, na->adj_fc2_w, na->adj_fc2_b, F, s))) return r;
  if ((r = launch_sym_precond_ch(B, na->Ca, N, F, coef ? adj : nullptr, node_flags, coef, out_adj, sym_err, s))) return r;
  if ((r = launch_na_pool(rep, Cp, B, N, p, node_flags, pooled, s))) return r;
  GemmEpi e2 = {};
  e2.bias = na->node_fc1.b; e2.act = 1; e2.out_bf16 = hidn; e2.ldo = Cp; e2.n_store = Cp;
  if ((r = launch_gemm(pooled, Cp, (const bf16*)na->node_fc1.w, B * N, C0, Cp, e2, s))) return r;
  return launch_na_head_node(hidn, Cp, B, N, C0, na->Cn, na->node_fc2_w, na->node_fc2_b, node, node_flags, coef,
                             out_node, s);
}

int check_nodeadj(const swg_net* net, const swg_nodeadj* na) {
  if (!net || !na) return SWG_E_NULL;
  if (na->Ca <= 0 || na->Cn <= 0 || na->Ca > 8 || na->Cn > 16) return SWG_E_SHAPE;
  const int C0 = net->C0, Cp = kp(C0), pp = net->p * net->p;
  if (C0 > 128) return SWG_E_UNSUPPORTED;
  if (!na->adj_fc2_w || !na->adj_fc2_b || !na->node_fc2_w || !na->node_fc2_b || !na->ro.b) return SWG_E_NULL;
  if (na->ro.N != pp * Cp || na->ro.K != C0 || na->ro.Kp != Cp || !na->ro.w) return SWG_E_SHAPE;
  int r;
  if ((r = check_lin(na->adj_fc1, C0, C0))) return r;
  return check_lin(na->node_fc1, C0, C0);
}

}  // namespace

extern "C" {

int swg_version(void) { return 1; }

size_t swg_workspace_bytes(const swg_net* net, int32_t B) {
  NetPlan pl;
  if (B <= 0 || plan_net(net, B, &pl)) return 0;
  return pl.persistent + pl.scratch + 4096;
}

int32_t swg_fused_op_count(const swg_net* net, int32_t B) {
  if (!net || !net->layers || B <= 0) return SWG_E_NULL;
  int32_t n = 0;
  for (int li = 0; li < net->n_layers; ++li) {
    const swg_layer& ly = net->layers[li];
    if (ly.up && expand_fused_ok(ly.up, B)) ++n;
    for (int j = 0; j < ly.n_blocks; ++j) {
      const swg_block* b = &ly.blocks[j];
      if (fused_tail_ok(b, B) || fused_tail_hd20_ok(b)) n += 2;
      else if (mlp_stream_ok(b, B)) ++n;
    }
    if (ly.down && merge_fused_ok(ly.down, B)) ++n;
  }
  return n;
}

int swg_linear_bf16(int32_t M, int32_t N, int32_t Kp, const void* a, int32_t lda, const void* w, const float* bias,
                    int32_t act, const float* residual, float* out_f32, void* out_bf16, int32_t ldo, void* stream) {
  if (!a || !w || (!out_f32 && !out_bf16)) return SWG_E_NULL;
  if (M <= 0 || N <= 0) return SWG_E_SHAPE;
  GemmEpi e = {};
  e.bias = bias; e.act = act; e.residual = residual; e.ldr = ldo; e.out_f32 = out_f32; e.out_bf16 = (bf16*)out_bf16;
  e.ldo = ldo; e.n_store = N;
  return launch_gemm((const bf16*)a, lda, (const bf16*)w, M, N, Kp, e, (hipStream_t)stream);
}

int swg_noise_embed(const swg_net* net, int32_t Bc, const float* c_noise, float* cond, void* ws, size_t ws_bytes,
                    void* stream) {
  if (Bc <= 0) return SWG_E_SHAPE;
  return embed_impl(net, Bc, c_noise, cond, ws, ws_bytes, (hipStream_t)stream);
}

int swg_patch_embed_fwd(const swg_net* net, int32_t B, const float* x, const float* self_cond, const float* c_in,
                        int32_t c_in_stride, const float* cond, int32_t cond_stride, float* tok, void* stream) {
  if (!net || !x || !cond || !tok || !net->pe_w || !net->pe_b || !net->pe_ln.g || !net->pe_ln.b) return SWG_E_NULL;
  if (B <= 0) return SWG_E_SHAPE;
  return launch_patch_embed(net, B, x, self_cond, c_in, c_in_stride, cond, cond_stride, tok, (hipStream_t)stream);
}

int swg_wattn_fwd(const swg_block* blk, int32_t B, const float* x_in, const float* cond, int32_t cond_stride,
                  float* x_out, void* ws, size_t ws_bytes, void* stream) {
  if (B <= 0) return SWG_E_SHAPE;
  return wattn_impl(blk, B, x_in, cond, cond_stride, x_out, ws, ws_bytes, (hipStream_t)stream);
}

int swg_block_fwd(const swg_block* blk, int32_t B, const float* x_in, const float* cond, int32_t cond_stride,
                  float* x_out, void* ws, size_t ws_bytes, void* stream) {
  if (B <= 0) return SWG_E_SHAPE;
  return block_impl(blk, B, x_in, cond, cond_stride, x_out, ws, ws_bytes, (hipStream_t)stream);
}

int swg_ln_mlp_fwd(const swg_block* blk, int32_t B, const float* x_in, float* x_out, void* ws, size_t ws_bytes,
                   void* stream) {
  if (B <= 0) return SWG_E_SHAPE;
  return mlp_impl(blk, B, x_in, x_out, ws, ws_bytes, (hipStream_t)stream);
}

int swg_patch_merge_fwd(const swg_merge* m, int32_t B, const float* x_in, float* x_out, void* ws, size_t ws_bytes,
                        void* stream) {
  if (B <= 0) return SWG_E_SHAPE;
  return merge_impl(m, B, x_in, x_out, ws, ws_bytes, (hipStream_t)stream);
}

int swg_patch_expand_fwd(const swg_breakup* u, int32_t B, const float* x, const float* skip, float* x_out, void* ws,
                         size_t ws_bytes, void* stream) {
  if (B <= 0) return SWG_E_SHAPE;
  return expand_impl(u, B, x, skip, x_out, ws, ws_bytes, (hipStream_t)stream);
}

int swg_readout_sym_precond_fwd(const swg_net* net, int32_t B, const float* tok, const float* x,
                                const float* node_flags, const float* coef, float* out, float* sym_err, void* ws,
                                size_t ws_bytes, void* stream) {
  if (B <= 0) return SWG_E_SHAPE;
  if (coef && !x) return SWG_E_NULL;
  return readout_impl(net, B, tok, x, node_flags, coef, out, sym_err, ws, ws_bytes, (hipStream_t)stream);
}

int swg_precond_forward_cond(const swg_net* net, int32_t B, int32_t mode, const float* x, const float* node_flags,
                             const float* labels, int32_t labels_uniform, const float* cond_rows,
                             const float* self_cond, float* out, float* sym_err, void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!net || !x || !node_flags || !labels || !out) return SWG_E_NULL;
  if (B <= 0 || (mode != 0 && mode != 1)) return SWG_E_SHAPE;
  TrunkBufs tb;
  int r = carve_trunk(net, B, labels_uniform, ws, ws_bytes, 0, &tb);
  if (r) return r;
  if ((r = launch_edm_coef(labels, B, labels_uniform, mode, tb.coef, tb.cnoise, s))) return r;
  if (cond_rows) {
    tb.cond = const_cast<float*>(cond_rows);  // read-only from here on
  } else if ((r = embed_impl(net, tb.Bc, tb.cnoise, tb.cond, tb.sw, tb.sb, s))) {
    return r;
  }
  if ((r = launch_patch_embed(net, B, x, self_cond, tb.coef + 2, 4, tb.cond, tb.cstride, tb.A, s))) return r;
  const float* cur = nullptr;
  if ((r = run_trunk(net, B, tb, &cur, s))) return r;
  return readout_impl(net, B, cur, x, node_flags, tb.coef, out, sym_err, tb.sw, tb.sb, s);
}

int swg_precond_forward(const swg_net* net, int32_t B, int32_t mode, const float* x, const float* node_flags,
                        const float* labels, int32_t labels_uniform, const float* self_cond, float* out,
                        float* sym_err, void* ws, size_t ws_bytes, void* stream) {
  return swg_precond_forward_cond(net, B, mode, x, node_flags, labels, labels_uniform, nullptr, self_cond, out, sym_err,
                                  ws, ws_bytes, stream);
}

int swg_cond_table(const swg_net* net, int32_t n, int32_t mode, const float* labels, float* cond_rows, void* ws,
                   size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!net || !labels || !cond_rows) return SWG_E_NULL;
  if (n <= 0 || (mode != 0 && mode != 1)) return SWG_E_SHAPE;
  Carver cv(ws, ws_bytes);
  float* coef = cv.take<float>((size_t)n * 4);
  float* cnoise = cv.take<float>((size_t)n);
  if (!cv.ok) return SWG_E_WORKSPACE;
  int r;
  if ((r = launch_edm_coef(labels, n, 0, mode, coef, cnoise, s))) return r;
  return embed_impl(net, n, cnoise, cond_rows, cv.p, cv.left, s);
}

size_t swg_nodeadj_workspace_bytes(const swg_net* net, const swg_nodeadj* na, int32_t B) {
  NetPlan pl;
  if (B <= 0 || !na || plan_net(net, B, &pl)) return 0;
  return pl.persistent + smax(pl.scratch, ws_na_tail(net, na, B)) + 4096;
}

int swg_nodeadj_forward(const swg_net* net, const swg_nodeadj* na, int32_t B, int32_t mode, const float* adj,
                        const float* node, const float* node_flags, const float* labels, int32_t labels_uniform,
                        const float* sc_adj, const float* sc_node, float* out_adj, float* out_node, float* sym_err,
                        void* ws, size_t ws_bytes, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (!net || !na || !adj || !node || !node_flags || !labels || !out_adj || !out_node) return SWG_E_NULL;
  if (B <= 0 || (mode != 0 && mode != 1)) return SWG_E_SHAPE;
  int r;
  if ((r = check_nodeadj(net, na))) return r;
  TrunkBufs tb;
  if ((r = carve_trunk(net, B, labels_uniform, ws, ws_bytes, ws_na_tail(net, na, B), &tb))) return r;
  if ((r = launch_edm_coef(labels, B, labels_uniform, mode, tb.coef, tb.cnoise, s))) return r;
  if ((r = embed_impl(net, tb.Bc, tb.cnoise, tb.cond, tb.sw, tb.sb, s))) return r;
  if ((r = launch_na_embed(net, na, B, adj, node, sc_adj, sc_node, node_flags, tb.coef + 2, 4, tb.cond, tb.cstride,
                           tb.A, s)))
    return r;
  const float* cur = nullptr;
  if ((r = run_trunk(net, B, tb, &cur, s))) return r;

  return na_readout_impl(net, na, B, mode, cur, adj, node, node_flags, mode ? tb.coef : nullptr, out_adj, out_node, sym_err,
                         tb.sw, tb.sb, s);
}

int swg_nodeadj_embed_fwd(const swg_net* net, const swg_nodeadj* na, int32_t B, const float* adj, const float* node,
                          const float* sc_adj, const float* sc_node, const float* node_flags, const float* c_in,
                          int32_t c_in_stride, const float* cond, int32_t cond_stride, float* tok, void* stream) {
  if (!net || !na || !adj || !node || !node_flags || !cond || !tok || !net->pe_w || !net->pe_b || !net->pe_ln.g ||
      !net->pe_ln.b)
    return SWG_E_NULL;
  if (B <= 0) return SWG_E_SHAPE;
  return launch_na_embed(net, na, B, adj, node, sc_adj, sc_node, node_flags, c_in, c_in_stride, cond, cond_stride, tok,
                         (hipStream_t)stream);
}

int swg_nodeadj_readout_fwd(const swg_net* net, const swg_nodeadj* na, int32_t B, const float* tok, const float* adj,
                            const float* node, const float* node_flags, const float* coef, float* out_adj,
                            float* out_node, float* sym_err, void* ws, size_t ws_bytes, void* stream) {
  if (!tok || !node_flags || !out_adj || !out_node || (coef && (!adj || !node))) return SWG_E_NULL;
  if (B <= 0) return SWG_E_SHAPE;
  int r;
  if ((r = check_nodeadj(net, na))) return r;
  if (!net->norm.g || !net->norm.b) return SWG_E_NULL;
  return na_readout_impl(net, na, B, coef ? 1 : 0, tok, adj, node, node_flags, coef, out_adj, out_node, sym_err, ws,
                         ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
