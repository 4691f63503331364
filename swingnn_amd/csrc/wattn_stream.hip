// Window-attention half of a Swin block for the WIDE stages (C = 384, 6x6 windows, head_dim 32), first of two
// kernels (model/swin_gnn/swin_gnn.py:238-268):
//     x1 = silu(shift + x (1 + scale));  o = softmax(q k^T * hd^-1/2 + bias + mask) v,  qkv = Wqkv LN1(x1)
// It writes o (bf16, token order, head-dim slots in the accumulator order pi) for the projection GEMM, whose epilogue
// recomputes the shortcut x1 (gemm.hip, GemmEpi::cond).  Same data flow as wattn_fused.hip -- one wavefront per
// window, token on the lane, every accumulator is directly the next product's operand -- but organised for one
// wave per SIMD with 512 registers:
//  * the projection is NOT fused: its accumulator (48 tokens x 384 channels fp32 = 288 VGPRs) does not fit next to
//    the window's LN1 fragments (144 VGPRs);
//  * the head's Wv / Wq / Wk row slices ("slabs", 32 x C bf16 = 24 fragments of 1 KiB in fragment order) stream
//    through an NBUF-deep LDS ring by LDS-DMA with counted waits, one barrier per slab;
//  * a slab's product is issued as pinned groups {3 MFMAs sharing one weight fragment, the ds_read of the fragment
//    AHEAD groups later, every 4th group one LDS-DMA piece} (see mlp_stream.hip for why the order is pinned).
#include <stdlib.h>

#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

struct StreamAttnArgs {
  const float* x;
  bf16* o;              // [T, ldo] bf16
  int ldo;
  const float* cond;
  int cond_stride, cond_off;
  const float* ln_g;
  const float* ln_b;
  const bf16* img;      // slab image, engine.py wattn_stream_image()
  const float* qkv_b;   // [3C]
  const float* rpb;     // [(2w-1)^2, heads]
  int B, res, shift, heads;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int C, int NBUF, int ABL>
__global__ __launch_bounds__(256, 1) void k_wattn_stream(StreamAttnArgs a) {
  constexpr int WIN = 6, NTOK = 36, NTM = 3, KSP = 2;
  constexpr int KS = C / 32;
  constexpr int NFRAG = 2 * KS;             // fragments per slab: (k-step, 16-row half of the 32 head dims)
  constexpr int SLAB_BYTES = NFRAG * 1024;
  constexpr int PIECES = SLAB_BYTES / 4096;  // LDS-DMA pieces per wave per slab
  constexpr int AHEAD = 6;
  constexpr int DMA_EVERY = NFRAG / PIECES;
  static_assert(SLAB_BYTES % 4096 == 0 && NFRAG % PIECES == 0 && AHEAD < NFRAG, "slab shape");
  static_assert((NBUF - 2) * PIECES < 64, "vmcnt is 6 bits");
  constexpr int W2 = 2 * WIN - 1, NB = W2 * W2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_qkvb = reinterpret_cast<float*>(smem + NBUF * SLAB_BYTES);  // [3C], q part pre-scaled
  float* s_rpb = s_qkvb + 3 * C;                                       // [heads][NB] * log2(e)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int nwx = a.res / WIN, nW = nwx * nwx;
  const int total = a.B * nW;
  int wi_g = blockIdx.x * 4 + wave;
  const bool live = wi_g < total;
  wi_g = min(wi_g, total - 1);
  const int b = wi_g / nW, wi = wi_g - b * nW;
  const int wy = wi / nwx, wx = wi - wy * nwx;
  const int Ltok = a.res * a.res;
  // region ids differ inside a window only in the last window row / column of a shifted block
  const bool need_mask = __builtin_amdgcn_readfirstlane((int)(a.shift > 0 && (wy == nwx - 1 || wx == nwx - 1))) != 0;
  const int nslab = 3 * a.heads;

  const unsigned smem0 = lds_addr(smem);
  const char* img_lane = reinterpret_cast<const char*>(a.img) + lane * 16;
#define STAGE_PIECE(slab, slot_off, i)                                                  \
  lds_dma16(img_lane + (size_t)(slab) * SLAB_BYTES + ((i) * 4 + wave) * 1024,            \
            smem0 + (slot_off) + ((i) * 4 + wave) * 1024)

  constexpr float LOG2E = 1.44269504088896340736f;
  constexpr float QSCALE = 0.17677669529663687f * LOG2E;  // 32^-0.5 * log2(e): scores live in the log2 domain
  for (int i = tid; i < 3 * C; i += 256) s_qkvb[i] = a.qkv_b[i] * (i < C ? QSCALE : 1.0f);
  for (int i = tid; i < NB * a.heads; i += 256) {
    const int hh = i / NB, idx = i - hh * NB;
    s_rpb[i] = a.rpb[idx * a.heads + hh] * LOG2E;
  }
#pragma unroll
  for (int sl = 0; sl < NBUF - 1; ++sl)
#pragma unroll
    for (int i = 0; i < PIECES; ++i) STAGE_PIECE(min(sl, nslab - 1), sl * SLAB_BYTES, i);

  // ---- per-lane token bookkeeping (as wattn_fused.hip); the softmax tables are built after the LayerNorm block so
  // that they are not live across its register peak ----
  int tokq[NTM];
  auto coords = [&](int i, int& rel_y, int& rel_x, int& rid, int& tok) {
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

  // ---- rows in operand layout: conditioning + SiLU, LayerNorm1 -> bf16 fragments (:238-244) ----
  bf16x8 xa[NTM][KS];
  const float* crow = a.cond + (size_t)b * a.cond_stride + a.cond_off;
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
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = silu_f(sh[j] + xv[j] * (sc[j] + 1.f));
        x1[ks][j] = v;
        s += v;
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
  asm volatile("" ::: "memory");
  int qrel[NTM], krel[NTM][4], rids[NTM];
  const float kpad = (lg == 0) ? 0.f : -3.0e38f;  // last key tile: keys 32 + 4*lg + r exist only for lg == 0
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
  __syncthreads();  // s_qkvb / s_rpb visible (the LDS-DMA is inline asm: nothing is drained here)

  // One slab product: ACC[dt][mt] (+)= W-slab(dt rows) x xa(mt)   (SWAP: ACC[mt][dt] = xa(mt) x W-slab(dt), for V).
  // `q` is the running slab number, `bo` the byte offset of its ring slot; the slot used one slab ago is refilled
  // with slab q + NBUF - 1 piece by piece (past the end the last slab is re-copied into a dead slot, which keeps
  // the vmcnt arithmetic uniform).
  int q = 0;
  unsigned bo = 0;
#define SLAB_GEMM(ACC, SWAP)                                                                                      \
  {                                                                                                               \
    if (ABL & 1) wait_vmcnt<0>(); else wait_vmcnt<(NBUF - 2) * PIECES>();                                         \
    __builtin_amdgcn_s_barrier();                                                                                 \
    asm volatile("" ::: "memory");                                                                                \
    const unsigned prev = (bo == 0) ? (NBUF - 1) * SLAB_BYTES : bo - SLAB_BYTES;                                  \
    const int next_slab = min(q + NBUF - 1, nslab - 1);                                                           \
    const char* cb = smem + bo + lane * 16;                                                                       \
    bf16x8 fr[AHEAD + 1];                                                                                         \
    _Pragma("unroll") for (int p = 0; p < AHEAD; ++p) fr[p] = *reinterpret_cast<const bf16x8*>(cb + p * 1024);   \
    __builtin_amdgcn_sched_barrier(0);                                                                            \
    _Pragma("unroll") for (int p = 0; p < NFRAG; ++p) {                                                           \
      const int ks = p >> 1, dt = p & 1;                                                                          \
      const bf16x8 w = fr[p % (AHEAD + 1)];                                                                       \
      _Pragma("unroll") for (int mt = 0; mt < NTM; ++mt) {                                                        \
        if (SWAP) ACC[mt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[mt][ks], w, ACC[mt][dt], 0, 0, 0);     \
        else ACC[dt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xa[mt][ks], ACC[dt][mt], 0, 0, 0);          \
      }                                                                                                           \
      if (p + AHEAD < NFRAG)                                                                                      \
        fr[(p + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const bf16x8*>(cb + (p + AHEAD) * 1024);                \
      if (p % DMA_EVERY == 1 && !(ABL & 1)) STAGE_PIECE(next_slab, prev, p / DMA_EVERY);                                     \
      __builtin_amdgcn_sched_barrier(0);                                                                          \
    }                                                                                                             \
    bo = (bo == (NBUF - 1) * SLAB_BYTES) ? 0u : bo + SLAB_BYTES;                                                  \
    ++q;                                                                                                          \
  }

#pragma nounroll
  for (int h = 0; h < a.heads; ++h) {
    // ---- V [tokens x 32] = A Wv^T (tokens on rows: directly the A operand of O^T = V^T P^T) ----
    bf16x8 qf[NTM], kf[NTM], vf[2][KSP];
    {
      f32x4 va[NTM][2];
#pragma unroll
      for (int mt = 0; mt < NTM; ++mt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) va[mt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      SLAB_GEMM(va, true);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        const float bv = s_qkvb[2 * C + h * 32 + dt * 16 + l15];
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
    // ---- Q^T, K^T [32 x tokens] = W A^T ; k-slot j of lane group lg is head dim (j<4 ? 4lg+j : 16+4lg+j-4) ----
    {
      f32x4 qa[2][NTM];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int mt = 0; mt < NTM; ++mt) qa[dt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      SLAB_GEMM(qa, false);
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
    {
      f32x4 ka[2][NTM];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int mt = 0; mt < NTM; ++mt) ka[dt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      SLAB_GEMM(ka, false);
      const float4 bk0 = *reinterpret_cast<const float4*>(s_qkvb + C + h * 32 + 4 * lg);
      const float4 bk1 = *reinterpret_cast<const float4*>(s_qkvb + C + h * 32 + 16 + 4 * lg);
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

    if (ABL & 2) {
#pragma unroll
      for (int qt = 0; qt < NTM; ++qt) {
        u32x4 f = __builtin_bit_cast(u32x4, qf[qt]) ^ __builtin_bit_cast(u32x4, kf[qt]) ^ __builtin_bit_cast(u32x4, vf[qt & 1][0]) ^ __builtin_bit_cast(u32x4, vf[0][1]) ^ __builtin_bit_cast(u32x4, vf[1][qt & 1]);
        if (live && qt * 16 + l15 < NTOK)
          *reinterpret_cast<u32x4*>(a.o + (size_t)tokq[qt] * a.ldo + h * 32 + 8 * lg) = f;
      }
      continue;
    }
    // ---- S^T = K Q^T (log2 domain), softmax over keys (rows) per query (lane) ----
    float p[NTM][NTM][4];
#pragma unroll
    for (int kt = 0; kt < NTM; ++kt)
#pragma unroll
      for (int qt = 0; qt < NTM; ++qt) {
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt], qf[qt], z, 0, 0, 0);
        p[kt][qt][0] = z[0]; p[kt][qt][1] = z[1]; p[kt][qt][2] = z[2]; p[kt][qt][3] = z[3];
      }
    const char* rp = reinterpret_cast<const char*>(s_rpb + h * NB);
#pragma unroll
    for (int kt = 0; kt < NTM; ++kt)
#pragma unroll
      for (int qt = 0; qt < NTM; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sv = p[kt][qt][r] + *reinterpret_cast<const float*>(rp + (qrel[qt] - krel[kt][r]));
          if (kt == NTM - 1) sv += kpad;
          p[kt][qt][r] = sv;
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
    float inv_sum[NTM];
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
      inv_sum[qt] = __frcp_rn(sum);
    }

    // ---- O^T = V^T P^T ; o rows to memory: channel slot h*32 + 8*lg + e holds head dim pi(lg, e) ----
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
      if (live && qt * 16 + l15 < NTOK)
        *reinterpret_cast<bf16x8*>(a.o + (size_t)tokq[qt] * a.ldo + h * 32 + 8 * lg) = f;
    }
  }
#undef SLAB_GEMM
#undef STAGE_PIECE
  wait_vmcnt<0>();  // trailing refills have landed before the workgroup's LDS is released
}

}  // namespace

bool wattn_stream_supported(int C, int heads, int win) { return C == 384 && heads * 32 == C && win == 6; }

int launch_wattn_stream(const swg_block* blk, int B, const float* x, const float* cond, int cond_stride, bf16* o,
                        int ldo, hipStream_t s) {
  const int C = blk->C;
  if (!wattn_stream_supported(C, blk->heads, blk->win) || !blk->attn_img) return SWG_E_UNSUPPORTED;
  if (blk->res % blk->win || ldo % 8) return SWG_E_SHAPE;
  constexpr int NBUF = 4;
  const int w2 = 2 * blk->win - 1;
  const int nW = (blk->res / blk->win) * (blk->res / blk->win);
  StreamAttnArgs a = {x, o, ldo, cond, cond_stride, blk->cond_off, blk->ln1.g, blk->ln1.b, (const bf16*)blk->attn_img,
                      blk->qkv.b, blk->rpb, B, blk->res, blk->shift, blk->heads};
  const size_t lds = (size_t)NBUF * (2 * (C / 32) * 1024) + (size_t)(3 * C + w2 * w2 * blk->heads) * sizeof(float);
  const double T = (double)B * blk->res * blk->res;
  SwgProfScope ps(KID_FUSED_WATTN, 6.0 * T * C * C + 4.0 * T * 36 * C, 6.0 * T * C + 6.0 * C * C, s);
  static const int abl = getenv("SWG_ATTN_ABL") ? atoi(getenv("SWG_ATTN_ABL")) : 0;  // TEMP
#define LAUNCH_ABL(A)                                                                                           \
  {                                                                                                             \
    hipError_t e = hipFuncSetAttribute((const void*)k_wattn_stream<384, NBUF, A>,                               \
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                   \
    if (e != hipSuccess) return (int)e;                                                                         \
    hipLaunchKernelGGL((k_wattn_stream<384, NBUF, A>), dim3(cdiv(B * nW, 4)), dim3(256), lds, s, a);            \
  }
  if (abl == 1) LAUNCH_ABL(1) else if (abl == 2) LAUNCH_ABL(2) else if (abl == 3) LAUNCH_ABL(3) else LAUNCH_ABL(0)
  SWG_LAUNCH_CHECK();
  return 0;
}
