// Window-attention half of a Swin block (model/swin_gnn/swin_gnn.py:238-272), 6x6 or 8x8 windows, head_dim 32:
//     x1 = silu(shift + x (1 + scale));  x_out = x1 + proj( softmax(q k^T * hd^-1/2 + bias + mask) v ),  qkv = Wqkv LN1(x1)
// window_partition / cyclic shift / window_reverse are pure addressing (gather list per window).
//
// gfx950 mapping
//  * ONE WAVEFRONT PER WINDOW, 4 windows per workgroup.  The wave loads its window's token rows straight into MFMA
//    operand layout (lane = token l&15, 8 consecutive channels per 32-deep k-step), applies the noise conditioning
//    and LayerNorm1 in registers and keeps the bf16 fragments for all heads.
//  * Per head everything stays in the register file; each product is oriented so that its accumulator is directly
//    the operand of the next one (the k-slot permutation an accumulator implies is applied to the OTHER operand):
//        V   = A Wv^T (tokens on rows)     Q^T = Wq A^T, K^T = Wk A^T (features on rows, tokens on lanes)
//        S^T = K Q^T  (softmax over rows = in-lane + row swaps)      O^T = V^T P^T      out^T += Wp_h O^T
//  * The weights are ONE stream of 1-KiB MFMA fragments (per head: Wv, Wq, Wk [, Wproj columns] slabs in fragment
//    order) flowing through an LDS ring of 24-fragment units by LDS-DMA.  The four waves walk the stream in lockstep:
//    one barrier per unit, taken one unit AHEAD of the reads so that the fragment prefetch never stops at a unit
//    boundary; counted vmcnt waits keep NBUF-3 units in flight across the barrier.
//  * Every fragment is one pinned issue group {3 MFMAs sharing the fragment, the ds_read of the fragment AHEAD
//    groups later, every 4th group one LDS-DMA piece}: with one or two waves per SIMD the compiler's own order
//    (read, wait, MFMA) exposes an LDS latency per fragment.
//  * FUSE_PROJ (C = 96, 192): the projection accumulates per head in registers and the kernel writes x_out.
//    C = 384: the projection accumulator (288 VGPRs) does not fit beside the LN1 fragments (144), so the kernel
//    writes o (bf16, head-dim slots in accumulator order pi) and the projection is a GEMM whose epilogue recomputes
//    the shortcut (gemm.hip, GemmEpi::cond).
#include "common.h"

namespace {

struct RingAttnArgs {
  const float* x;
  const bf16* xn;       // PRE_LN: LayerNorm1(x1) rows, bf16 [T, ldxn] (k_row<cond_ln>); else null
  int ldxn;
  float* out;           // FUSE_PROJ: x_out fp32 [T, C]
  bf16* o;              // !FUSE_PROJ: o bf16 [T, ldo]
  int ldo;
  const float* cond;
  int cond_stride, cond_off;
  const float* ln_g;
  const float* ln_b;
  const bf16* img;      // fragment stream, engine.py wattn_ring_image()
  const float* qkv_b;   // [3C]
  const float* proj_b;  // [C]
  const float* rpb;     // [(2w-1)^2, heads]
  int B, res, shift, heads;
  const float* rpb_tiles;  // BT: bias in accumulator layout, [heads][NTM][NTM][64][4] (swg_rpb_tiles)
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

__host__ __device__ constexpr int ring_unit(int frags_head) { return frags_head % 24 == 0 ? 24 : frags_head; }

template <int C, bool FUSE_PROJ, int NBUF, int LB, int AHEAD, bool PRE_LN, int WIN, bool BT = false, int PACK = 1, int HD = 32>
__global__ __launch_bounds__(256, LB) void k_wattn_ring(RingAttnArgs a) {
  // HD < 32 (head_dim 20: the dim-60 networks): C is the channel count rounded up to the 32-deep k-step (the rows of
  // k_row<cond_ln> are zero-padded to it), every head occupies a 32-slot group whose slots HD..31 are zero in the
  // weight image (engine.py wattn_ring_image), and o has heads x 32 columns for a projection GEMM over the padded slots.
  static_assert(HD == 32 || (PRE_LN && !FUSE_PROJ && !BT), "padded heads: attention core on pre-normalised rows only");
  const int CO = HD == 32 ? C : a.heads * 32;        // q / k / v slot groups
  // window geometry: NTOK tokens in NTM 16-token MFMA tiles, KSP 32-deep k-steps over the keys; the last tile holds
  // REM valid tokens (WIN = 6: 36 tokens = 3 tiles, REM 4; WIN = 8: 64 tokens = 4 full tiles; WIN = 4: one tile).
  // PACK > 1 (WIN = 2): PACK consecutive windows share the wave's one tile (token slot i = window i / NTOKW, token
  // i % NTOKW); the scores between different windows of the tile are excluded like padding keys.
  constexpr int NTOKW = WIN * WIN, NTOK = NTOKW * PACK;
  constexpr int NTM = (NTOK + 15) / 16, KSP = (NTM + 1) / 2, REM = NTOK - (NTM - 1) * 16;
  static_assert(REM % 4 == 0, "key padding is per lane group");
  static_assert(PACK == 1 || (NTM == 1 && !BT && PRE_LN && !FUSE_PROJ), "packed windows: one tile, attention core only");
  constexpr int KS = C / 32;
  constexpr int NT = C / 16;
  constexpr int SLAB = 2 * KS;                       // fragments per slab (== NT)
  constexpr int NSLAB = FUSE_PROJ ? 4 : 3;           // V, Q, K [, P]
  constexpr int FRAGS_HEAD = NSLAB * SLAB;
  // fragments per ring unit (one barrier per unit): 24, or the whole head where 24 does not divide it (C = 192
  // without the projection slab: 36)
  constexpr int UNIT = ring_unit(FRAGS_HEAD);
  constexpr int UNIT_BYTES = UNIT * 1024;
  constexpr int UH = FRAGS_HEAD / UNIT;              // units per head
  constexpr int RING_BYTES = NBUF * UNIT_BYTES;
  constexpr int PIECES = UNIT / 4;                   // LDS-DMA pieces per wave per unit
  constexpr int DMA_EVERY = UNIT / PIECES;
  static_assert(FRAGS_HEAD % UNIT == 0 && UNIT % 4 == 0 && NBUF >= 3 && AHEAD < UNIT, "stream shape");
  static_assert(FRAGS_HEAD % (AHEAD + 1) == 0, "the register fragment ring must line up across heads");
  static_assert((NBUF - 2) * PIECES < 64, "vmcnt is 6 bits");
  constexpr int W2 = 2 * WIN - 1, NB = W2 * W2;
  // 6x6 windows: the 36 bias values of a lane (3 x 3 tile pairs x 4 accumulator rows) are read straight into the
  // accumulators of S^T = K Q^T through 36 head-invariant LDS offsets kept in registers; entry NB of a head's table is
  // the key padding (-3e38).  8x8 windows (64 offsets) keep the index arithmetic: C = 192 is at the register limit.
  // BT (6x6 windows with swg_block.rpb_tiles): the nine bias tiles of a head are loaded from global memory straight
  // into the S accumulators' registers at the top of the head (inline asm, counted vmcnt: the loads are older than
  // every LDS-DMA piece of the head's slabs, so they have landed when that many operations are outstanding) -- no LDS
  // reads for the bias at all.  The 36 ds_read_b32 per head with their 2-way bank conflicts were 10 % of the C = 384
  // kernel (ablation: 712 -> 643 us per pass without them).
  constexpr bool BIAS_REGS = (WIN == 6) && !BT;
  constexpr int NBT = BT ? NTM * NTM : 0;                       // bias loads per head
  constexpr int NAFTER = (3 * SLAB - 2) / DMA_EVERY + 1;        // LDS-DMA pieces issued between them and S
  constexpr int NBP = NB + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_qkvb = reinterpret_cast<float*>(smem + RING_BYTES);  // [3C], q part pre-scaled
  float* s_pb = s_qkvb + 3 * CO;                                // [C]
  float* s_rpb = s_pb + CO;                                     // [heads][NB] * log2(e)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int nwx = a.res / WIN, nW = nwx * nwx;
  const int total = a.B * nW;
  const int wi_0 = (blockIdx.x * 4 + wave) * PACK;   // the wave's first window
  const bool live = wi_0 < total;
  const int wi_g = min(wi_0, total - 1);
  const int b = wi_g / nW, wi = wi_g - b * nW;
  const int wy = wi / nwx, wx = wi - wy * nwx;
  const int Ltok = a.res * a.res;
  // region ids differ inside a window only in the last window row / column of a shifted block
  const bool need_mask = __builtin_amdgcn_readfirstlane(
      (int)(a.shift > 0 && (PACK > 1 || wy == nwx - 1 || wx == nwx - 1))) != 0;
  const int nunits = UH * a.heads;
  // gridDim.y head groups (launches with few windows: each workgroup streams only its heads' part of the weights)
  const int hpg = a.heads / (int)gridDim.y, h0 = (int)blockIdx.y * hpg;
#ifdef SWG_STAMPS
  const bool dbg = (blockIdx.x == 8 || blockIdx.x == gridDim.x / 2) && (wave == 0 || wave == 2);
  const int dbgslot = (blockIdx.x == 8 ? 0 : 2) + (wave == 0 ? 0 : 1);
#endif
  STAMP(0);

  const unsigned smem0 = lds_addr(smem);
  const char* img_lane = reinterpret_cast<const char*>(a.img) + lane * 16;
  // LDS-DMA piece i (of PIECES per wave) of stream unit `unit` into the ring at byte offset slot_off
#define STAGE_PIECE(unit, slot_off, i)                                                  \
  lds_dma16(img_lane + (size_t)(unit) * UNIT_BYTES + ((i) * 4 + wave) * 1024,            \
            smem0 + (slot_off) + ((i) * 4 + wave) * 1024)

  constexpr float LOG2E = 1.44269504088896340736f;
  // head_dim^-0.5 * log2(e): scores live in the log2 domain
  constexpr float QSCALE = (HD == 32 ? 0.17677669529663687f : HD == 20 ? 0.22360679774997896f : 0.f) * LOG2E;
  static_assert(HD == 32 || HD == 20, "head dims with a tabulated scale");

  // ---- per-lane token bookkeeping; the softmax tables are built after the LayerNorm block so that they are not
  // live across its register peak ----
  int tokq[NTM];
  const int b_w = b, wy_w = wy, wx_w = wx;
  auto coords = [&](int i, int& rel_y, int& rel_x, int& rid, int& tok) {
    int b = b_w, wy = wy_w, wx = wx_w;            // the token's window: the wave's, or window i / NTOKW of its pack
    if (PACK > 1) {
      const int sub = i / NTOKW;
      i -= sub * NTOKW;
      const int wg = min(wi_0 + sub, total - 1);
      b = wg / nW;
      const int w = wg - b * nW;
      wy = w / nwx;
      wx = w - wy * nwx;
    }
    const int iy = i / WIN, ix = i - iy * WIN;
    const int ys = wy * WIN + iy, xs = wx * WIN + ix;
    int y = ys + a.shift, x = xs + a.shift;  // shifted[y] = x[(y+s) mod H]  (roll by -s, :248)
    if (y >= a.res) y -= a.res;
    if (x >= a.res) x -= a.res;
    tok = b * Ltok + y * a.res + x;
    rid = 0;
    if (a.shift > 0) {
      const int ry = ys < a.res - WIN ? 0 : (ys < a.res - a.shift ? 1 : 2);
      const int rx = xs < a.res - WIN ? 0 : (xs < a.res - a.shift ? 1 : 2);
      rid = 3 * ry + rx;
    }
    rel_y = iy;
    rel_x = ix;
  };
#pragma unroll
  for (int mt = 0; mt < NTM; ++mt) {
    int iy, ix, rid, tok;
    coords(min(mt * 16 + l15, NTOK - 1), iy, ix, rid, tok);
    tokq[mt] = tok;
  }

  // ---- operand fragments of the window's rows.  PRE_LN: the rows were conditioned + normalised by k_row<cond_ln>
  // (bf16), one 16-byte load per fragment.  Otherwise conditioning + SiLU + LayerNorm1 run here (:238-244).
  // The weight ring is primed only after the row loads: LDS-DMA issued ahead of them would delay them (loads
  // return in order).
  bf16x8 xa[NTM][KS];
  // Projection accumulators.  With in-kernel conditioning (!PRE_LN) they START as shortcut + proj bias: accumulator
  // row 4*lg+r of out tile nt = 2*ks+hh is channel 32ks + 8lg + 4hh + r (rho order of the image rows), the channels
  // this lane conditions for k-step ks, so the epilogue is a plain store.
  f32x4 acc[FUSE_PROJ ? NT : 1][NTM];
  if (FUSE_PROJ && PRE_LN) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < NTM; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float* crow = a.cond + (size_t)b * a.cond_stride + a.cond_off;
  if (PRE_LN) {
#pragma unroll
    for (int mt = 0; mt < NTM; ++mt) {
      const bf16* row = a.xn + (size_t)tokq[mt] * a.ldxn + 8 * lg;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) xa[mt][ks] = *reinterpret_cast<const bf16x8*>(row + 32 * ks);
    }
  }
  if (HD == 32) {
    for (int i = tid; i < 3 * C; i += 256) s_qkvb[i] = a.qkv_b[i] * (i < C ? QSCALE : 1.0f);
  } else {   // slot part * CO + h * 32 + d  <-  qkv bias part * (heads * HD) + h * HD + d, zero for the padded slots
    for (int i = tid; i < 3 * CO; i += 256) {
      const int part = i / CO, r = i - part * CO, hh = r >> 5, d = r & 31;
      s_qkvb[i] = d < HD ? a.qkv_b[(part * a.heads + hh) * HD + d] * (part == 0 ? QSCALE : 1.0f) : 0.f;
    }
  }
  if (FUSE_PROJ)
    for (int i = tid; i < C; i += 256) s_pb[i] = a.proj_b[i];
  for (int i = tid; i < NBP * a.heads; i += 256) {
    const int hh = i / NBP, idx = i - hh * NBP;
    s_rpb[i] = idx < NB ? a.rpb[idx * a.heads + hh] * LOG2E : -3.0e38f;
  }
  if (!PRE_LN) {
#pragma unroll
    for (int mt = 0; mt < NTM; ++mt) {
      const float* row = a.x + (size_t)tokq[mt] * C;
      float x1[KS][8];
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int c0 = 32 * ks + 8 * lg;
        const float4 p0 = *reinterpret_cast<const float4*>(row + c0);
        const float4 p1 = *reinterpret_cast<const float4*>(row + c0 + 4);
        const float4 sc0 = *reinterpret_cast<const float4*>(crow + c0);
        const float4 sc1 = *reinterpret_cast<const float4*>(crow + c0 + 4);
        const float4 sh0 = *reinterpret_cast<const float4*>(crow + C + c0);
        const float4 sh1 = *reinterpret_cast<const float4*>(crow + C + c0 + 4);
        const float xv[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
        const float sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
        float pb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (FUSE_PROJ) {
          const float4 pb0 = *reinterpret_cast<const float4*>(a.proj_b + c0);
          const float4 pb1 = *reinterpret_cast<const float4*>(a.proj_b + c0 + 4);
          pb[0] = pb0.x; pb[1] = pb0.y; pb[2] = pb0.z; pb[3] = pb0.w;
          pb[4] = pb1.x; pb[5] = pb1.y; pb[6] = pb1.z; pb[7] = pb1.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = silu_f(sh[j] + xv[j] * (sc[j] + 1.f));
          x1[ks][j] = v;
          s += v;
          if (FUSE_PROJ) acc[FUSE_PROJ ? 2 * ks + (j >> 2) : 0][mt][j & 3] = v + pb[j];
        }
      }
      s = rows_sum(s);
      const float mean = s * (1.0f / C);
      float q = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = x1[ks][j] - mean; q += d * d; }
      q = rows_sum(q);
      const float rstd = ln_rstd(q * (1.0f / C));
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int c0 = 32 * ks + 8 * lg;
        const float4 g0 = *reinterpret_cast<const float4*>(a.ln_g + c0);
        const float4 g1 = *reinterpret_cast<const float4*>(a.ln_g + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(a.ln_b + c0);
        const float4 b1 = *reinterpret_cast<const float4*>(a.ln_b + c0 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (bf16)((x1[ks][j] - mean) * rstd * gg[j] + bb[j]);
        xa[mt][ks] = f;
      }
    }
  }
  asm volatile("" ::: "memory");
#pragma unroll
  for (int u = 0; u < NBUF - 1; ++u)
#pragma unroll
    for (int i = 0; i < PIECES; ++i) STAGE_PIECE(min(h0 * UH + u, nunits - 1), u * UNIT_BYTES, i);
  int qrel[NTM], krel[NTM][4], rids[NTM];
  int boff[BIAS_REGS ? NTM : 1][BIAS_REGS ? NTM : 1][4];
  const float kpad = (4 * lg < REM) ? 0.f : -3.0e38f;  // last key tile: keys 4*lg + r exist only below REM
#pragma unroll
  for (int mt = 0; mt < NTM; ++mt) {
    int iy, ix, rid, tok;
    coords(min(mt * 16 + l15, NTOK - 1), iy, ix, rid, tok);
    qrel[mt] = 4 * ((iy + WIN - 1) * W2 + (ix + WIN - 1));
    rids[mt] = rid;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = mt * 16 + 4 * lg + r;
      int jy, jx, jr, jt;
      coords(min(j, NTOK - 1), jy, jx, jr, jt);
      krel[mt][r] = 4 * (jy * W2 + jx);
      rids[mt] |= jr << (4 + 4 * r);
    }
  }
  if (BIAS_REGS) {
#pragma unroll
    for (int kt = 0; kt < NTM; ++kt)
#pragma unroll
      for (int qt = 0; qt < NTM; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          boff[BIAS_REGS ? kt : 0][BIAS_REGS ? qt : 0][r] =
              (kt * 16 + 4 * lg + r < NTOK) ? qrel[qt] - krel[kt][r] : 4 * NB;
  }

  // unit 0 has landed everywhere (and the tables are visible): start the fragment ring
  wait_vmcnt<(NBUF - 2) * PIECES>();
  __syncthreads();
  STAMP(2);
  bf16x8 fr[AHEAD + 1];
#pragma unroll
  for (int p = 0; p < AHEAD; ++p) fr[p] = *reinterpret_cast<const bf16x8*>(smem + lane * 16 + p * 1024);
  int unit = h0 * UH;  // stream unit of the head's first fragment
  int slot = 0;      // its ring slot (unit % NBUF)

  // One issue group: fragment G of the head (compile-time G).  At a unit start the barrier is taken for the NEXT
  // unit (counted wait: NBUF-3 units stay in flight) and the slot of the previous unit becomes the refill target.
#define GROUP(G, BODY)                                                                                              \
  {                                                                                                                 \
    if ((G) % UNIT == 0) {                                                                                          \
      /* the first boundary after the head's bias loads: they are younger than the unit the wait is about */       \
      if ((G) == UNIT) wait_vmcnt<(NBUF - 3) * PIECES + NBT>();                                                     \
      else wait_vmcnt<(NBUF - 3) * PIECES>();                                                                       \
      __builtin_amdgcn_s_barrier();                                                                                 \
      asm volatile("" ::: "memory");                                                                                \
    }                                                                                                               \
    const bf16x8 w = fr[(G) % (AHEAD + 1)];                                                                         \
    BODY;                                                                                                           \
    {                                                                                                               \
      const int GN = (G) + AHEAD;                                                                                   \
      fr[GN % (AHEAD + 1)] = *reinterpret_cast<const bf16x8*>(ub[GN / UNIT] + (GN % UNIT) * 1024);                  \
    }                                                                                                               \
    if ((G) % DMA_EVERY == 1) {                                                                                     \
      const int UI = (G) / UNIT;                                                                                    \
      STAGE_PIECE(min(unit + UI + NBUF - 1, nunits - 1), (UI > 0) ? uoff[UI > 0 ? UI - 1 : 0] : prev_off,           \
                  ((G) % UNIT) / DMA_EVERY);                                                                        \
    }                                                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                              \
  }

#pragma nounroll
  for (int h = h0; h < h0 + hpg; ++h) {
    // ring slots of the head's units and of the next head's first unit (the fragment prefetch runs AHEAD into it)
    unsigned uoff[UH + 1];
    const char* ub[UH + 1];
    const unsigned prev_off = (unsigned)((slot == 0 ? NBUF : slot) - 1) * UNIT_BYTES;
    {
      int sl = slot;
#pragma unroll
      for (int k = 0; k <= UH; ++k) {
        uoff[k] = (unsigned)sl * UNIT_BYTES;
        ub[k] = smem + uoff[k] + lane * 16;
        sl = (sl + 1 == NBUF) ? 0 : sl + 1;
      }
    }

    if (h < 3) STAMP(3 + 8 * h);
    // ---- V [tokens x 32] = A Wv^T (tokens on rows: directly the A operand of O^T = V^T P^T) ----
    bf16x8 qf[NTM], kf[NTM], vf[2][KSP];
    f32x4 bz[BT ? NTM * NTM : 1];
    {
      f32x4 va[NTM][2];
#pragma unroll
      for (int mt = 0; mt < NTM; ++mt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) va[mt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < SLAB; ++p) {
        GROUP(p, _Pragma("unroll") for (int mt = 0; mt < NTM; ++mt) va[mt][p & 1] =
                     __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[mt][p >> 1], w, va[mt][p & 1], 0, 0, 0));
        if (BT && p == 0) {   // after the unit-boundary wait of group 0, before the first DMA piece of the head
          const float* bt = a.rpb_tiles + ((size_t)h * (NTM * NTM * 64) + lane) * 4;
#pragma unroll
          for (int t = 0; t < NBT; ++t)
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bz[BT ? t : 0]) : "v"(bt + t * 256) : "memory");
          __builtin_amdgcn_sched_barrier(0);
        }
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const float bv = s_qkvb[2 * CO + h * 32 + dt * 16 + l15];
#pragma unroll
        for (int ks = 0; ks < KSP; ++ks) {
          bf16x8 f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            f[j] = (bf16)(va[2 * ks][dt][j] + bv);
            f[4 + j] = (2 * ks + 1 < NTM) ? (bf16)(va[(2 * ks + 1 < NTM) ? 2 * ks + 1 : 0][dt][j] + bv) : (bf16)0.f;
          }
          vf[dt][ks] = f;
        }
      }
    }
    if (h < 3) STAMP(4 + 8 * h);
      // ---- Q^T, K^T [32 x tokens] = W A^T ; k-slot j of lane group lg is head dim (j<4 ? 4lg+j : 16+4lg+j-4) ----
    {
      f32x4 qa[2][NTM];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int mt = 0; mt < NTM; ++mt) qa[dt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < SLAB; ++p) {
        GROUP(SLAB + p, _Pragma("unroll") for (int mt = 0; mt < NTM; ++mt) qa[p & 1][mt] =
                            __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xa[mt][p >> 1], qa[p & 1][mt], 0, 0, 0));
      }
      const float4 bq0 = *reinterpret_cast<const float4*>(s_qkvb + h * 32 + 4 * lg);
      const float4 bq1 = *reinterpret_cast<const float4*>(s_qkvb + h * 32 + 16 + 4 * lg);
#pragma unroll
      for (int mt = 0; mt < NTM; ++mt) {
        bf16x8 f;
        f[0] = (bf16)fmaf(qa[0][mt][0], QSCALE, bq0.x); f[1] = (bf16)fmaf(qa[0][mt][1], QSCALE, bq0.y);
        f[2] = (bf16)fmaf(qa[0][mt][2], QSCALE, bq0.z); f[3] = (bf16)fmaf(qa[0][mt][3], QSCALE, bq0.w);
        f[4] = (bf16)fmaf(qa[1][mt][0], QSCALE, bq1.x); f[5] = (bf16)fmaf(qa[1][mt][1], QSCALE, bq1.y);
        f[6] = (bf16)fmaf(qa[1][mt][2], QSCALE, bq1.z); f[7] = (bf16)fmaf(qa[1][mt][3], QSCALE, bq1.w);
        qf[mt] = f;
      }
    }
    if (h < 3) STAMP(5 + 8 * h);
    {
      f32x4 ka[2][NTM];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int mt = 0; mt < NTM; ++mt) ka[dt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int p = 0; p < SLAB; ++p) {
        GROUP(2 * SLAB + p, _Pragma("unroll") for (int mt = 0; mt < NTM; ++mt) ka[p & 1][mt] =
                                __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xa[mt][p >> 1], ka[p & 1][mt], 0, 0, 0));
      }
      const float4 bk0 = *reinterpret_cast<const float4*>(s_qkvb + CO + h * 32 + 4 * lg);
      const float4 bk1 = *reinterpret_cast<const float4*>(s_qkvb + CO + h * 32 + 16 + 4 * lg);
#pragma unroll
      for (int mt = 0; mt < NTM; ++mt) {
        bf16x8 g;
        g[0] = (bf16)(ka[0][mt][0] + bk0.x); g[1] = (bf16)(ka[0][mt][1] + bk0.y);
        g[2] = (bf16)(ka[0][mt][2] + bk0.z); g[3] = (bf16)(ka[0][mt][3] + bk0.w);
        g[4] = (bf16)(ka[1][mt][0] + bk1.x); g[5] = (bf16)(ka[1][mt][1] + bk1.y);
        g[6] = (bf16)(ka[1][mt][2] + bk1.z); g[7] = (bf16)(ka[1][mt][3] + bk1.w);
        kf[mt] = g;
      }
    }

    if (h < 3) STAMP(6 + 8 * h);
      // ---- S^T = K Q^T (log2 domain), softmax over keys (rows) per query (lane) ----
    float p[NTM][NTM][4];
    const char* rp = reinterpret_cast<const char*>(s_rpb + h * NBP);
    if (BT) {
      static_assert(!BT || NTM == 3, "nine tiles");
      // the loads are older than the NAFTER pieces issued since: done once at most that many operations are pending
      asm volatile("s_waitcnt vmcnt(%9)"
                   : "+v"(bz[0]), "+v"(bz[BT ? 1 : 0]), "+v"(bz[BT ? 2 : 0]), "+v"(bz[BT ? 3 : 0]), "+v"(bz[BT ? 4 : 0]),
                     "+v"(bz[BT ? 5 : 0]), "+v"(bz[BT ? 6 : 0]), "+v"(bz[BT ? 7 : 0]), "+v"(bz[BT ? 8 : 0])
                   : "n"(NAFTER));
#pragma unroll
      for (int kt = 0; kt < NTM; ++kt)
#pragma unroll
        for (int qt = 0; qt < NTM; ++qt) {
          f32x4 z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt], qf[qt], bz[BT ? kt * NTM + qt : 0], 0, 0, 0);
          p[kt][qt][0] = z[0]; p[kt][qt][1] = z[1]; p[kt][qt][2] = z[2]; p[kt][qt][3] = z[3];
        }
    } else if (BIAS_REGS) {
#pragma unroll
      for (int kt = 0; kt < NTM; ++kt)
#pragma unroll
        for (int qt = 0; qt < NTM; ++qt) {
          f32x4 z;
#ifdef RING_DIAG_NOBIAS   // diagnostic only (wrong results)
          z = (f32x4){0.f, 0.f, 0.f, 0.f};
#else
#pragma unroll
          for (int r = 0; r < 4; ++r)
            z[r] = *reinterpret_cast<const float*>(rp + boff[BIAS_REGS ? kt : 0][BIAS_REGS ? qt : 0][r]);
#endif
          z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt], qf[qt], z, 0, 0, 0);
          p[kt][qt][0] = z[0]; p[kt][qt][1] = z[1]; p[kt][qt][2] = z[2]; p[kt][qt][3] = z[3];
        }
    } else {
#pragma unroll
      for (int kt = 0; kt < NTM; ++kt)
#pragma unroll
        for (int qt = 0; qt < NTM; ++qt) {
          f32x4 z = {0.f, 0.f, 0.f, 0.f};
          z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt], qf[qt], z, 0, 0, 0);
          p[kt][qt][0] = z[0]; p[kt][qt][1] = z[1]; p[kt][qt][2] = z[2]; p[kt][qt][3] = z[3];
        }
#pragma unroll
      for (int kt = 0; kt < NTM; ++kt)
#pragma unroll
        for (int qt = 0; qt < NTM; ++qt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float sv = p[kt][qt][r] + *reinterpret_cast<const float*>(rp + (qrel[qt] - krel[kt][r]));
            if (kt == NTM - 1) sv += kpad;
            if (PACK > 1 && (4 * lg + r) / NTOKW != l15 / NTOKW) sv = -3.0e38f;   // another window of the pack
            p[kt][qt][r] = sv;
          }
    }
    if (need_mask) {  // wave-uniform: only the border windows of shifted blocks pay for the region mask (:207-226)
#pragma unroll
      for (int kt = 0; kt < NTM; ++kt)
#pragma unroll
        for (int qt = 0; qt < NTM; ++qt)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (((rids[kt] >> (4 + 4 * r)) & 15) != (rids[qt] & 15)) p[kt][qt][r] += -100.0f * LOG2E;
    }
    if (h < 3) STAMP(7 + 8 * h);
    float inv_sum[NTM];
#ifdef RING_DIAG_NOSOFTMAX   // diagnostic only (wrong results): no max / exp / sum
#pragma unroll
    for (int qt = 0; qt < NTM; ++qt) inv_sum[qt] = 1.0f;
#else
#pragma unroll
    for (int qt = 0; qt < NTM; ++qt) {
      float mx = -3.0e38f;
#pragma unroll
      for (int kt = 0; kt < NTM; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, p[kt][qt][r]);
      mx = rows_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NTM; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(p[kt][qt][r] - mx);  // padding keys: exp2(-huge) = 0
          p[kt][qt][r] = e;
          sum += e;
        }
      sum = rows_sum(sum);
      inv_sum[qt] = fast_rcp(sum);
    }
#endif

    if (h < 3) STAMP(8 + 8 * h);
    // ---- O^T = V^T P^T ; channel slot h*32 + 8*lg + e of a row holds head dim pi(lg, e) ----
    bf16x8 of[NTM];
#pragma unroll
    for (int qt = 0; qt < NTM; ++qt) {
      bf16x8 pf[KSP];
#pragma unroll
      for (int ks = 0; ks < KSP; ++ks) {
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          f[j] = (bf16)p[2 * ks][qt][j];
          f[4 + j] = (2 * ks + 1 < NTM) ? (bf16)p[(2 * ks + 1 < NTM) ? 2 * ks + 1 : 0][qt][j] : (bf16)0.f;
        }
        pf[ks] = f;
      }
      f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KSP; ++ks) {
        o0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[0][ks], pf[ks], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[1][ks], pf[ks], o1, 0, 0, 0);
      }
      const float is = inv_sum[qt];
      bf16x8 f;
      f[0] = (bf16)(o0[0] * is); f[1] = (bf16)(o0[1] * is); f[2] = (bf16)(o0[2] * is); f[3] = (bf16)(o0[3] * is);
      f[4] = (bf16)(o1[0] * is); f[5] = (bf16)(o1[1] * is); f[6] = (bf16)(o1[2] * is); f[7] = (bf16)(o1[3] * is);
      of[qt] = f;
      if (!FUSE_PROJ) {
        if (live && qt * 16 + l15 < NTOK && (PACK == 1 || wi_0 + l15 / NTOKW < total))
          *reinterpret_cast<bf16x8*>(a.o + (size_t)tokq[qt] * a.ldo + h * 32 + 8 * lg) = f;
      }
    }
    if (h < 3) STAMP(9 + 8 * h);
      if (FUSE_PROJ) {
      // ---- out^T += Wp_h O^T (Wproj columns carry the head-dim slot order, rows the rho order, in the image) ----
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p2 = 0; p2 < SLAB; ++p2) {
        GROUP(3 * SLAB + p2, _Pragma("unroll") for (int mt = 0; mt < NTM; ++mt) acc[p2][mt] =
                                 __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, of[mt], acc[p2][mt], 0, 0, 0));
      }
    }
    unit += UH;
    slot += UH;
    if (slot >= NBUF) slot -= NBUF;
    }
#undef GROUP
#undef STAGE_PIECE
  STAMP(27);
  wait_vmcnt<0>();  // trailing refills have landed before the workgroup's LDS is released

  if (FUSE_PROJ && !PRE_LN) {
    // ---- epilogue: scatter back to token order (shortcut and bias are already in the accumulators) ----
#pragma unroll
    for (int mt = 0; mt < NTM; ++mt) {
      if (!live || mt * 16 + l15 >= NTOK) continue;
      float* orow = a.out + (size_t)tokq[mt] * C;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *reinterpret_cast<f32x4*>(orow + 32 * (nt >> 1) + 8 * lg + 4 * (nt & 1)) = acc[FUSE_PROJ ? nt : 0][mt];
    }
  }
  if (FUSE_PROJ && PRE_LN) {
    // ---- epilogue: + proj bias + shortcut x1 (recomputed from the cached row), scatter back to token order.
    // Accumulator row 4*lg+r of tile nt = 2*ks+hh is channel 32ks+8lg+4hh+r (rho order of the image rows).
#pragma unroll
    for (int mt = 0; mt < NTM; ++mt) {
      if (!live || mt * 16 + l15 >= NTOK) continue;
      float* orow = a.out + (size_t)tokq[mt] * C;
      const float* xrow = a.x + (size_t)tokq[mt] * C;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = 32 * (nt >> 1) + 8 * lg + 4 * (nt & 1);
        const float4 bb = *reinterpret_cast<const float4*>(s_pb + c);
        const float4 xv = *reinterpret_cast<const float4*>(xrow + c);
        const float4 sc = *reinterpret_cast<const float4*>(crow + c);
        const float4 sf = *reinterpret_cast<const float4*>(crow + C + c);
        float4 o;
        o.x = acc[nt][mt][0] + bb.x + silu_f(sf.x + xv.x * (sc.x + 1.f));
        o.y = acc[nt][mt][1] + bb.y + silu_f(sf.y + xv.y * (sc.y + 1.f));
        o.z = acc[nt][mt][2] + bb.z + silu_f(sf.z + xv.z * (sc.z + 1.f));
        o.w = acc[nt][mt][3] + bb.w + silu_f(sf.w + xv.w * (sc.w + 1.f));
        *reinterpret_cast<float4*>(orow + c) = o;
      }
    }
  }
}

template <int C, bool FUSE_PROJ, int NBUF, int LB, int AHEAD, bool PRE_LN, int WIN, bool BT = false, int PACK = 1, int HD = 32>
int launch_t(const RingAttnArgs& a, int nwg, hipStream_t s) {
  constexpr int UNIT = ring_unit((FUSE_PROJ ? 4 : 3) * 2 * (C / 32));
  // Few windows (the deep stages of the small-graph networks): split the heads over gridDim.y so that the launch has
  // up to 256 workgroups, each streaming a fraction of the weights (attention-core instances only: o is per head).
  int hg = 1;
  if (!FUSE_PROJ)
    while (nwg * hg * 2 <= 256 && a.heads % (hg * 2) == 0 && a.heads / (hg * 2) >= 3) hg *= 2;
  const int CO = HD == 32 ? C : a.heads * 32;
  const size_t lds = (size_t)NBUF * UNIT * 1024 + (size_t)(4 * CO + ((2 * WIN - 1) * (2 * WIN - 1) + 1) * a.heads) * sizeof(float);
  static std::atomic<uint64_t> lds_ok{0};  // dynamic LDS above 64 KiB needs an explicit opt-in, per device
  if (int e = ensure_dyn_lds(lds_ok, (const void*)k_wattn_ring<C, FUSE_PROJ, NBUF, LB, AHEAD, PRE_LN, WIN, BT, PACK, HD>, (int)((int)lds))) return e;
  hipLaunchKernelGGL((k_wattn_ring<C, FUSE_PROJ, NBUF, LB, AHEAD, PRE_LN, WIN, BT, PACK, HD>), dim3(nwg, hg), dim3(256), lds, s, a);
  return 0;
}

}  // namespace

// The C = 192 instances live in a translation unit of their own (wattn_ring192.hip: this file compiled with
// -DRING_TU_192 and -mllvm -amdgpu-mfma-vgpr-form=1).  With MFMA results in the architectural VGPRs the attention-core
// instance fits 233 registers with no accumulation registers at all and loses 190 v_accvgpr moves per window
// (361 -> 333 us per pass); the same flag costs the C = 384 instances 1 % (and k_mlp_stream<384,proj> spills under it),
// and the flag is per translation unit.
int launch_wattn_ring_c192(const void* args, int nwg, int win, bool fuse_proj, hipStream_t s);
// head_dim 20 (C = 60 / 120 / 240, rows padded to CK = 64 / 128 / 256): wattn_ring_hd20.hip (-DRING_TU_HD20)
int launch_wattn_ring_hd20(const void* args, int nwg, int CK, int win, hipStream_t s);

#if defined(RING_TU_HD20)
namespace {
template <int CK>
int launch_hd20(const RingAttnArgs& a, int nwg, int win, hipStream_t s) {
  if (win == 8) return launch_t<CK, false, 4, 1, 5, true, 8, false, 1, 20>(a, nwg, s);
  if (win == 4) return launch_t<CK, false, 4, 1, 5, true, 4, false, 1, 20>(a, nwg, s);
  if (win == 2) return launch_t<CK, false, 4, 1, 5, true, 2, false, 4, 20>(a, nwg, s);
  return SWG_E_UNSUPPORTED;
}
}  // namespace
int launch_wattn_ring_hd20(const void* args, int nwg, int CK, int win, hipStream_t s) {
  const RingAttnArgs& a = *static_cast<const RingAttnArgs*>(args);
  if (CK == 64) return launch_hd20<64>(a, nwg, win, s);
  if (CK == 128) return launch_hd20<128>(a, nwg, win, s);
  if (CK == 256) return launch_hd20<256>(a, nwg, win, s);
  return SWG_E_UNSUPPORTED;
}
#elif defined(RING_TU_192)
int launch_wattn_ring_c192(const void* args, int nwg, int win, bool fuse_proj, hipStream_t s) {
  const RingAttnArgs& a = *static_cast<const RingAttnArgs*>(args);
  // small windows (attention core on pre-normalised rows only): 4x4 = one MFMA tile per wave, 2x2 = four windows per tile
  if (win == 4) return launch_t<192, false, 4, 1, 5, true, 4>(a, nwg, s);
  if (win == 2) return launch_t<192, false, 4, 1, 5, true, 2, false, 4>(a, nwg, s);
  if (!a.xn && !fuse_proj)   // conditioning + LayerNorm1 inside the kernel (no k_row<cond_ln> pass in front)
    return win == 8 ? launch_t<192, false, 4, 1, 5, false, 8>(a, nwg, s) : launch_t<192, false, 4, 1, 5, false, 6>(a, nwg, s);
  if (win == 8)
    return fuse_proj ? launch_t<192, true, 4, 1, 5, true, 8>(a, nwg, s) : launch_t<192, false, 4, 1, 5, true, 8>(a, nwg, s);
  if (a.rpb_tiles && !fuse_proj) return launch_t<192, false, 4, 1, 5, true, 6, true>(a, nwg, s);
  return fuse_proj ? launch_t<192, true, 4, 1, 5, true, 6>(a, nwg, s) : launch_t<192, false, 4, 1, 5, true, 6>(a, nwg, s);
}
#else
// 6x6 / 8x8 windows: every form.  4x4 (one MFMA tile per wave) and 2x2 windows (four windows per tile): the attention
// core on pre-normalised rows only.
bool wattn_ring_supported(int C, int heads, int win, int res) {
  if (res % win) return false;
  // head_dim 20 (the dim-60 networks): attention core with every head padded to a 32-slot group
  if ((C == 60 || C == 120 || C == 240) && heads * 20 == C) return win == 8 || win == 4 || win == 2;
  if (!((C == 192 || C == 384) && heads * 32 == C)) return false;
  if (win == 6 || win == 8) return true;
  return (win == 4 || win == 2) && !wattn_ring_fuses_proj(C) && wattn_ring_wants_xn(C);
}
// C = 384: the projection accumulators do not fit beside the LN1 fragments.  C = 192: they fit, but the kernel runs
// faster as the attention core alone (36 instead of 48 fragments per head, no 144 accumulator registers, no shortcut
// epilogue) with the projection folded into the streamed MLP kernel (mlp_stream.hip PROJ form), as at C = 384.
// SWG_RING192_FUSE_PROJ=1 restores the fused-projection form (A/B knob of scripts/; engine.py reads the same variable
// when it builds the weight image).
bool wattn_ring_fuses_proj(int C) {
  static const bool fuse192 = [] {
    const char* e = getenv("SWG_RING192_FUSE_PROJ");
    return e && e[0] == '1';
  }();
  return C == 192 && fuse192;
}

// C = 384 (and the fused-projection form at C = 192) take the rows conditioned + normalised by k_row<cond_ln>.  The
// attention-core instance at C = 192 has the registers to do it in its prologue (SWG_RING192_INKERNEL_LN, A/B knob).
bool wattn_ring_wants_xn(int C) {
  static const bool inkernel192 = [] {
    const char* e = getenv("SWG_RING192_INKERNEL_LN");
    return e && e[0] == '1';
  }();
  return !(C == 192 && inkernel192 && !wattn_ring_fuses_proj(C));
}

// FUSE_PROJ geometries write x_out (`out`); C = 384 writes o (bf16 [T, ldo]) for the projection GEMM.  C >= 192 take
// the rows already conditioned + normalised (xn, bf16 [T, ldxn], from k_row<cond_ln>): in-kernel LayerNorm of a
// 48 x C tile per wave is a long latency-bound prologue that nothing overlaps at one workgroup per CU.
int launch_wattn_ring(const swg_block* blk, int B, const float* x, const bf16* xn, int ldxn, const float* cond,
                      int cond_stride, float* out, bf16* o, int ldo, hipStream_t s) {
  const int C = blk->C;
  if (!wattn_ring_supported(C, blk->heads, blk->win, blk->res) || !blk->attn_img) return SWG_E_UNSUPPORTED;
  const bool hd20 = blk->heads * 20 == C;
  if (hd20 && ldo < blk->heads * 32) return SWG_E_SHAPE;
  if (wattn_ring_fuses_proj(C) ? !out : (!o || ldo % 8)) return SWG_E_NULL;
  if (wattn_ring_wants_xn(C) && (!xn || ldxn % 8)) return SWG_E_NULL;
  const int nW = (blk->res / blk->win) * (blk->res / blk->win);
  RingAttnArgs a = {x, xn, ldxn, out, o, ldo, cond, cond_stride, blk->cond_off, blk->ln1.g, blk->ln1.b,
                    (const bf16*)blk->attn_img, blk->qkv.b, blk->proj.b, blk->rpb, B, blk->res, blk->shift, blk->heads,
                    blk->win == 6 ? blk->rpb_tiles : nullptr};
  const int nwg = cdiv(B * nW, blk->win == 2 ? 16 : 4);
  const double T = (double)B * blk->res * blk->res;
  const double gemm = wattn_ring_fuses_proj(C) ? 8.0 : 6.0;
  // algorithmic bytes per token and channel: fused projection: x in (4, shortcut) + out (4) [+ 2 for the pre-normalised
  // bf16 rows]; attention core only: normalised rows in (2) + o out (2).  Weights: one pass of the stream (2 B each).
  const double row_bytes = wattn_ring_fuses_proj(C) ? (xn ? 10.0 : 8.0) : 4.0;
  SwgProfScope ps(hd20 ? KID_WATTN_RING_HD20 : C == 384 ? KID_WATTN_RING_384 : KID_WATTN_RING, gemm * T * C * C + 4.0 * T * (blk->win * blk->win) * C,
                  row_bytes * T * C + gemm * C * C, s);
  int r;
  if (hd20) {
    r = launch_wattn_ring_hd20(&a, nwg, round_up(C, 32), blk->win, s);
  } else if (blk->win == 4 || blk->win == 2) {
    if (C == 192) r = launch_wattn_ring_c192(&a, nwg, blk->win, false, s);
    else r = blk->win == 4 ? launch_t<384, false, 5, 1, 5, true, 4>(a, nwg, s)
                           : launch_t<384, false, 5, 1, 5, true, 2, false, 4>(a, nwg, s);
  } else if (blk->win == 8) {
    if (C == 192) r = launch_wattn_ring_c192(&a, nwg, 8, wattn_ring_fuses_proj(C), s);
    else r = launch_t<384, false, 5, 1, 5, true, 8>(a, nwg, s);
  } else {
    if (C == 192) r = launch_wattn_ring_c192(&a, nwg, 6, wattn_ring_fuses_proj(C), s);
    else r = a.rpb_tiles ? launch_t<384, false, 5, 1, 5, true, 6, true>(a, nwg, s)
                         : launch_t<384, false, 5, 1, 5, true, 6>(a, nwg, s);
  }
  if (r) return r;
  SWG_LAUNCH_CHECK();
  return 0;
}

// ---- swg_rpb_tiles: relative-position bias as accumulator-layout tiles (derived cache of swg_block.rpb_tiles) ----
namespace {
__global__ void k_rpb_tiles(const float* __restrict__ rpb, int heads, int win, int ntm, float* __restrict__ tiles) {
  const int ntok = win * win, w2 = 2 * win - 1;
  const int total = heads * ntm * ntm * 64;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    const int ln = e & 63, t = e >> 6;
    const int qt = t % ntm, kt = (t / ntm) % ntm, hh = t / (ntm * ntm);
    const int i = min(qt * 16 + (ln & 15), ntok - 1);
    const int iy = i / win, ix = i - iy * win;
    float v[4];
    for (int r = 0; r < 4; ++r) {
      const int j = kt * 16 + 4 * (ln >> 4) + r;
      const int jy = j / win, jx = j - jy * win;
      v[r] = j < ntok ? rpb[((iy - jy + win - 1) * w2 + (ix - jx + win - 1)) * heads + hh] * 1.44269504088896340736f
                      : -3.0e38f;
    }
    *reinterpret_cast<float4*>(tiles + (size_t)e * 4) = make_float4(v[0], v[1], v[2], v[3]);
  }
}
}  // namespace

extern "C" int swg_rpb_tiles(const float* rpb, int32_t heads, int32_t win, float* tiles, void* stream) {
  if (!rpb || !tiles) return SWG_E_NULL;
  if (heads <= 0 || win <= 0 || win > 8) return SWG_E_SHAPE;
  const int ntm = (win * win + 15) / 16;
  hipLaunchKernelGGL(k_rpb_tiles, dim3(cdiv(heads * ntm * ntm * 64, 256)), dim3(256), 0, (hipStream_t)stream, rpb, heads,
                     win, ntm, tiles);
  SWG_LAUNCH_CHECK();
  return 0;
}
#endif  // RING_TU_192
