// Fused window-attention half of a Swin block (model/swin_gnn/swin_gnn.py:238-272) in ONE pass over the
// residual stream:
//     x1 = silu(shift + x (1 + scale));  x_out = x1 + proj( softmax(q k^T * hd^-1/2 + bias + mask) v ),  qkv = Wqkv LN1(x1)
// window_partition / cyclic shift / window_reverse are pure addressing (gather list per window).
//
// gfx950 mapping (head_dim 32, C in {96, 192}, windows of up to 64 tokens)
//  * ONE WAVEFRONT PER WINDOW, 4 windows per workgroup.  The wave loads its window's token rows straight into
//    MFMA operand layout (lane = token l&15, 8 consecutive channels per 32-deep k-step), applies the noise
//    conditioning and LayerNorm1 in registers (row statistics = 2 xor-shuffles) and keeps x1 for the residual.
//  * Per head, everything stays in the register file; each product is oriented so that its accumulator is
//    directly the operand of the next one (the k-slot permutation an accumulator implies is applied to the
//    OTHER operand: to Q/K consistently, to V by computing it with swapped roles, to Wproj in the host image):
//        Q^T = Wq A^T, K^T = Wk A^T        (features on rows, tokens on lanes)
//        S^T = K Q^T                       (keys on rows, queries on lanes -> softmax over rows is in-lane + 2 shuffles)
//        V   = A Wv^T                      (tokens on rows)  ->  O^T = V^T P^T     ->  out^T += Wp_h O^T
//  * The four waves walk the heads in lockstep; the head's weight slices (Wq,Wk,Wv rows and the Wproj columns,
//    host-prepared as a swizzled fragment image) stream through LDS by LDS-DMA, double buffered, one barrier
//    per head.  Relative-position bias table and biases live in LDS; the shift mask is computed in-register
//    from region ids.
#include "common.h"

namespace {

struct FusedAttnArgs {
  const float* x;
  float* out;
  const float* cond;
  int cond_stride, cond_off;
  const float* ln_g;
  const float* ln_b;
  const bf16* img;      // per-head weight image, engine.py wattn_image()
  const float* qkv_b;   // [3C]
  const float* proj_b;  // [C]
  const float* rpb;     // [(2w-1)^2, heads]
  int B, res, win, shift, heads;
};

__device__ __forceinline__ int slot_swz(int row, int ch) { return ch ^ ((-(row >> 2)) & 3); }

template <int C, int NTM, bool KEEPX, int LB>
__global__ __launch_bounds__(256, LB) void k_wattn_fused(FusedAttnArgs a) {
  constexpr int KS = C / 32;             // k-steps over channels
  constexpr int NT = C / 16;             // output feature tiles of proj
  constexpr int KSP = (NTM + 1) / 2;     // 32-key steps of P.V
  constexpr int PART = 32 * C;           // elements of one Wq/Wk/Wv head slice, and of the Wproj head slice
  constexpr int HEAD_BYTES = 4 * PART * 2;
  constexpr int DMA_PER_WAVE = HEAD_BYTES / 4096;
  static_assert(HEAD_BYTES % 4096 == 0, "head slice must be whole 1-KiB pieces per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* s_qkvb = reinterpret_cast<float*>(smem + 2 * HEAD_BYTES);  // [3C]
  float* s_pb = s_qkvb + 3 * C;                                     // [C]
  float* s_rpb = s_pb + C;                                          // [(2w-1)^2 * heads]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int n = a.win * a.win;
  const int nwx = a.res / a.win, nW = nwx * nwx;
  const int total = a.B * nW;
  int wi_g = blockIdx.x * 4 + wave;
  const bool live = wi_g < total;
  wi_g = min(wi_g, total - 1);
  const int b = wi_g / nW, wi = wi_g - b * nW;
  const int wy = wi / nwx, wx = wi - wy * nwx;
  const int Ltok = a.res * a.res;
  // region ids differ inside a window only in the last window row / column of a shifted block
  const bool need_mask = __builtin_amdgcn_readfirstlane((int)(a.shift > 0 && (wy == nwx - 1 || wx == nwx - 1))) != 0;
  const int w2 = 2 * a.win - 1;

  auto stage = [&](int h, int buf) {
    const char* src = reinterpret_cast<const char*>(a.img) + (size_t)h * HEAD_BYTES;
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
      const int piece = i * 4 + wave;
      lds_dma16(src + piece * 1024 + lane * 16, lds_addr(smem) + buf * HEAD_BYTES + piece * 1024);
    }
  };
  stage(0, 0);
  // scores are kept in the log2 domain: q carries hd^-1/2 * log2(e), the bias table is pre-multiplied by log2(e)
  constexpr float LOG2E = 1.44269504088896340736f;
  constexpr float QSCALE = 0.17677669529663687f * LOG2E;  // 32^-0.5 * log2(e)
  const int nb = w2 * w2;
  for (int i = tid; i < 3 * C; i += 256) s_qkvb[i] = a.qkv_b[i] * (i < C ? QSCALE : 1.0f);
  for (int i = tid; i < C; i += 256) s_pb[i] = a.proj_b[i];
  for (int i = tid; i < nb * a.heads; i += 256) {
    const int hh = i / nb, idx = i - hh * nb;  // LDS layout [head][idx]
    s_rpb[i] = a.rpb[idx * a.heads + hh] * LOG2E;
  }

  // ---- per-lane token bookkeeping ----
  // query side: token i = mt*16 + l15 ; key side: token j = kt*16 + 4*lg + r
  int tokq[NTM];      // global token index of the lane's query/row token (clamped for padding rows)
  int qrel[NTM];      // 4 * ((iy+w-1)*(2w-1) + (ix+w-1)): byte offset base into the head's bias column
  int krel[NTM][4];   // 4 * (jy*(2w-1) + jx)
  int rids[NTM];      // region ids, packed: query id in bits 0..3, key r in bits 4+4r.. (used only when shift > 0)
  float kpad[4];      // additive mask of the last key tile: 0 for real keys, -huge for padding
  auto coords = [&](int i, int& rel_y, int& rel_x, int& rid, int& tok) {
    const int iy = i / a.win, ix = i - iy * a.win;
    const int ys = wy * a.win + iy, xs = wx * a.win + ix;
    int y = ys + a.shift, x = xs + a.shift;  // shifted[y] = x[(y+s) mod H]  (roll by -s, :248)
    if (y >= a.res) y -= a.res;
    if (x >= a.res) x -= a.res;
    tok = b * Ltok + y * a.res + x;
    rid = 0;
    if (a.shift > 0) {
      const int ry = ys < a.res - a.win ? 0 : (ys < a.res - a.shift ? 1 : 2);
      const int rx = xs < a.res - a.win ? 0 : (xs < a.res - a.shift ? 1 : 2);
      rid = 3 * ry + rx;
    }
    rel_y = iy;
    rel_x = ix;
  };
#pragma unroll
  for (int mt = 0; mt < NTM; ++mt) {
    int iy, ix, rid, tok;
    coords(min(mt * 16 + l15, n - 1), iy, ix, rid, tok);
    tokq[mt] = tok;
    qrel[mt] = 4 * ((iy + a.win - 1) * w2 + (ix + a.win - 1));
    rids[mt] = rid;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = mt * 16 + 4 * lg + r;
      int jy, jx, jr, jt;
      coords(min(j, n - 1), jy, jx, jr, jt);
      krel[mt][r] = 4 * (jy * w2 + jx);
      rids[mt] |= jr << (4 + 4 * r);
      if (mt == NTM - 1) kpad[r] = (j < n) ? 0.f : -3.0e38f;
    }
  }

  // ---- load rows in operand layout, conditioning + SiLU (x1 = shortcut), LayerNorm1 -> bf16 fragments ----
  bf16x8 xa[NTM][KS];
  float x1[KEEPX ? NTM : 1][KS][8];  // shortcut rows (KEEPX) / per-tile scratch
  const float* crow = a.cond + (size_t)b * a.cond_stride + a.cond_off;
#pragma unroll
  for (int mt = 0; mt < NTM; ++mt) {
    const int xi = KEEPX ? mt : 0;
    const float* row = a.x + (size_t)tokq[mt] * C;
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
        const float v = silu_f(sh[j] + xv[j] * (sc[j] + 1.f));  // :238-240
        x1[xi][ks][j] = v;
        s += v;
      }
    }
    s = rows_sum(s);
    const float mean = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = x1[xi][ks][j] - mean; q += d * d; }
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
      for (int j = 0; j < 8; ++j) f[j] = (bf16)((x1[xi][ks][j] - mean) * rstd * gg[j] + bb[j]);
      xa[mt][ks] = f;
    }
  }

  f32x4 acc[NT][NTM];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int mt = 0; mt < NTM; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // lane's fragment byte offsets inside a head slice: rows dt*16+l15 of a 32-row part, proj rows nt*16+l15
  int qkv_off[2];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt) {
    const int r = dt * 16 + l15;
    qkv_off[dt] = (r * 4 + slot_swz(r, lg)) * 16;
  }
  int p_off[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int r = nt * 16 + l15;
    p_off[nt] = 3 * PART * 2 + (r * 4 + slot_swz(r, lg)) * 16;
  }

  for (int h = 0; h < a.heads; ++h) {
    const int buf = h & 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's LDS-DMA of head h has landed
    __syncthreads();                                  // ... everyone's has; the other buffer is free
    if (h + 1 < a.heads) stage(h + 1, buf ^ 1);
    const char* cb = smem + buf * HEAD_BYTES;

    // ---- V [tokens x 32] = A Wv^T first (its fragments are small), then Q^T, K^T [32 x tokens] = W A^T ----
    bf16x8 qf[NTM], kf[NTM], vf[2][KSP];
    {
      f32x4 va[NTM][2];
#pragma unroll
      for (int mt = 0; mt < NTM; ++mt)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) va[mt][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 wv[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
          wv[dt] = *reinterpret_cast<const bf16x8*>(cb + 2 * PART * 2 + qkv_off[dt] + ks * 2048);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int mt = 0; mt < NTM; ++mt)
            va[mt][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[mt][ks], wv[dt], va[mt][dt], 0, 0, 0);
      }
      // V rows (tokens 4lg+r of tile mt) give exactly the key-slot order the P^T accumulators have
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
    {
      f32x4 qa[2][NTM], ka[2][NTM];
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int mt = 0; mt < NTM; ++mt) {
          qa[dt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          ka[dt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 wq[2], wk[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          wq[dt] = *reinterpret_cast<const bf16x8*>(cb + qkv_off[dt] + ks * 2048);
          wk[dt] = *reinterpret_cast<const bf16x8*>(cb + PART * 2 + qkv_off[dt] + ks * 2048);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int mt = 0; mt < NTM; ++mt) {
            qa[dt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[dt], xa[mt][ks], qa[dt][mt], 0, 0, 0);
            ka[dt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wk[dt], xa[mt][ks], ka[dt][mt], 0, 0, 0);
          }
      }
      // bias (q: pre-scaled, log2 domain) and bf16 fragments.  k-slot j of lane group lg is head dim
      // (j<4 ? 4lg+j : 16+4lg+j-4) for Q and K alike.
      const float4 bq0 = *reinterpret_cast<const float4*>(s_qkvb + h * 32 + 4 * lg);
      const float4 bq1 = *reinterpret_cast<const float4*>(s_qkvb + h * 32 + 16 + 4 * lg);
      const float4 bk0 = *reinterpret_cast<const float4*>(s_qkvb + C + h * 32 + 4 * lg);
      const float4 bk1 = *reinterpret_cast<const float4*>(s_qkvb + C + h * 32 + 16 + 4 * lg);
#pragma unroll
      for (int mt = 0; mt < NTM; ++mt) {
        bf16x8 f, g;
        f[0] = (bf16)fmaf(qa[0][mt][0], QSCALE, bq0.x); f[1] = (bf16)fmaf(qa[0][mt][1], QSCALE, bq0.y);
        f[2] = (bf16)fmaf(qa[0][mt][2], QSCALE, bq0.z); f[3] = (bf16)fmaf(qa[0][mt][3], QSCALE, bq0.w);
        f[4] = (bf16)fmaf(qa[1][mt][0], QSCALE, bq1.x); f[5] = (bf16)fmaf(qa[1][mt][1], QSCALE, bq1.y);
        f[6] = (bf16)fmaf(qa[1][mt][2], QSCALE, bq1.z); f[7] = (bf16)fmaf(qa[1][mt][3], QSCALE, bq1.w);
        g[0] = (bf16)(ka[0][mt][0] + bk0.x); g[1] = (bf16)(ka[0][mt][1] + bk0.y);
        g[2] = (bf16)(ka[0][mt][2] + bk0.z); g[3] = (bf16)(ka[0][mt][3] + bk0.w);
        g[4] = (bf16)(ka[1][mt][0] + bk1.x); g[5] = (bf16)(ka[1][mt][1] + bk1.y);
        g[6] = (bf16)(ka[1][mt][2] + bk1.z); g[7] = (bf16)(ka[1][mt][3] + bk1.w);
        qf[mt] = f;
        kf[mt] = g;
      }
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
    const char* rp = reinterpret_cast<const char*>(s_rpb + h * nb);
#pragma unroll
    for (int kt = 0; kt < NTM; ++kt)
#pragma unroll
      for (int qt = 0; qt < NTM; ++qt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float sv = p[kt][qt][r] + *reinterpret_cast<const float*>(rp + (qrel[qt] - krel[kt][r]));
          if (kt == NTM - 1) sv += kpad[r];
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

    // ---- O^T = V^T P^T, then out^T += Wp_h O^T (Wproj columns carry the head-dim slot order in the image) ----
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
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bf16x8 wp = *reinterpret_cast<const bf16x8*>(cb + p_off[nt]);
#pragma unroll
      for (int mt = 0; mt < NTM; ++mt)
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp, of[mt], acc[nt][mt], 0, 0, 0);
    }
  }

  // ---- epilogue: + proj bias + shortcut (registers), scatter back to token order ----
  // proj image rows are ordered so that accumulator row 4*lg+r of tile nt = 2*ks+hh is channel 32ks+8lg+4hh+r.
#pragma unroll
  for (int mt = 0; mt < NTM; ++mt) {
    if (!live || mt * 16 + l15 >= n) continue;
    float* orow = a.out + (size_t)tokq[mt] * C;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ks = nt >> 1, hh = nt & 1;
      const int c = 32 * ks + 8 * lg + 4 * hh;
      const float4 bb = *reinterpret_cast<const float4*>(s_pb + c);
      float4 sh;  // shortcut x1 = silu(shift + x (1 + scale)): from registers, or recomputed from the (cached) row
      if (KEEPX) {
        sh = make_float4(x1[KEEPX ? mt : 0][ks][4 * hh + 0], x1[KEEPX ? mt : 0][ks][4 * hh + 1],
                         x1[KEEPX ? mt : 0][ks][4 * hh + 2], x1[KEEPX ? mt : 0][ks][4 * hh + 3]);
      } else {
        const float4 xv = *reinterpret_cast<const float4*>(a.x + (size_t)tokq[mt] * C + c);
        const float4 sc = *reinterpret_cast<const float4*>(crow + c);
        const float4 sf = *reinterpret_cast<const float4*>(crow + C + c);
        sh = make_float4(silu_f(sf.x + xv.x * (sc.x + 1.f)), silu_f(sf.y + xv.y * (sc.y + 1.f)),
                         silu_f(sf.z + xv.z * (sc.z + 1.f)), silu_f(sf.w + xv.w * (sc.w + 1.f)));
      }
      float4 o;
      o.x = acc[nt][mt][0] + bb.x + sh.x;
      o.y = acc[nt][mt][1] + bb.y + sh.y;
      o.z = acc[nt][mt][2] + bb.z + sh.z;
      o.w = acc[nt][mt][3] + bb.w + sh.w;
      *reinterpret_cast<float4*>(orow + c) = o;
    }
  }
}

template <int C, int NTM, bool KEEPX, int LB>
int launch_t(const FusedAttnArgs& a, size_t lds, int nwg, hipStream_t s) {
  static bool attr_set = false;
  if (!attr_set && lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute((const void*)k_wattn_fused<C, NTM, KEEPX, LB>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL((k_wattn_fused<C, NTM, KEEPX, LB>), dim3(nwg), dim3(256), lds, s, a);
  return 0;
}

}  // namespace

bool wattn_fused_supported(int C, int heads, int win) {
  return C == 96 && heads * 32 == C && win == 6;  // 64-token windows need 2 waves per window (todo)
}

int launch_wattn_fused(const swg_block* blk, int B, const float* x, const float* cond, int cond_stride, float* out,
                       hipStream_t s) {
  const int C = blk->C;
  if (!wattn_fused_supported(C, blk->heads, blk->win) || !blk->attn_img) return SWG_E_UNSUPPORTED;
  const int n = blk->win * blk->win, NTM = cdiv(n, 16);
  const int w2 = 2 * blk->win - 1;
  const int nW = (blk->res / blk->win) * (blk->res / blk->win);
  FusedAttnArgs a = {x, out, cond, cond_stride, blk->cond_off, blk->ln1.g, blk->ln1.b, (const bf16*)blk->attn_img,
                     blk->qkv.b, blk->proj.b, blk->rpb, B, blk->res, blk->win, blk->shift, blk->heads};
  const size_t lds = (size_t)2 * (4 * 32 * C * 2) + (size_t)(4 * C + w2 * w2 * blk->heads) * sizeof(float);
  const int nwg = cdiv(B * nW, 4);
  const double T = (double)B * blk->res * blk->res;
  SwgProfScope ps(KID_FUSED_WATTN, 8.0 * T * C * C + 4.0 * T * n * C, 8.0 * T * C + 8.0 * C * C, s);
  int r;
  if (C == 96 && NTM == 3) r = launch_t<96, 3, false, 2>(a, lds, nwg, s);
  else return SWG_E_UNSUPPORTED;
  if (r) return r;
  SWG_LAUNCH_CHECK();
  return 0;
}
