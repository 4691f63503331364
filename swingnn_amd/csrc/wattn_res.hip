// Window-attention half of a Swin block for the NARROW stage (C = 96, 3 heads, 4x4 / 6x6 / 8x8 windows; swin_gnn.py:238-272):
//     x1 = silu(shift + x (1 + scale));  x_out = x1 + proj( softmax(q k^T * hd^-1/2 + bias + mask) v ),  qkv = Wqkv LN1(x1)
// Same per-window data flow and pinned fragment groups as wattn_ring.hip, but all weights of the block (3 heads x 24
// fragments = 72 KiB) stay RESIDENT in LDS: persistent workgroups of 8 waves copy them once by LDS-DMA, then every
// wave walks over windows on its own -- no barrier, no DMA and no counted wait inside the window loop.  With two
// waves per SIMD that drift apart, one wave's VALU phases (conditioning + LayerNorm1, softmax) overlap the other's
// MFMA phases; at this width the kernel is bound by exactly those VALU phases and by the fp32 token stream.
#include "common.h"

// tuning switches (A/B builds of scripts/build_variant.sh).  Measured on one box, grid B = 112, us per pass (4 launches):
// round-2 form 494; LayerNorm1 g / b from LDS 476 (default); + contiguous run of windows per wave 487; + the graph's
// conditioning row staged in a per-wave LDS slot 588 (the staged row is re-read per token tile -- 36 instead of 12
// reads -- and sits in the dependent chain of the SiLU; the global reads were hoisted out of the tile loop).
#ifndef RES_WALK_CONTIG
#define RES_WALK_CONTIG 0   // 1: every wave walks a contiguous run of windows; 0: windows dealt round-robin to the waves
#endif
#ifndef RES_COND_LDS
#define RES_COND_LDS 0      // 1: (1 + scale | shift) of the wave's graph staged in LDS; 0: read through the address unit
#endif

namespace {

struct ResAttnArgs {
  const float* x;
  float* out;
  const float* cond;
  int cond_stride, cond_off;
  const float* ln_g;
  const float* ln_b;
  const bf16* img;      // fragment stream, engine.py wattn_ring_image(fuse_proj=True)
  const float* qkv_b;   // [3C]
  const float* proj_b;  // [C]
  const float* rpb;     // [(2w-1)^2, heads]
  int B, res, shift, heads;
};

template <int C, int NW, int AHEAD, int WIN>
__global__ __launch_bounds__(64 * NW) void k_wattn_res(ResAttnArgs a) {
  // window geometry: NTOK tokens in NTM 16-token MFMA tiles, KSP 32-deep k-steps over the keys; the last tile holds
  // REM valid tokens (WIN = 6: 36 tokens = 3 tiles, REM 4; WIN = 8: 64 = 4 tiles; WIN = 4: 16 = 1 tile)
  constexpr int NTOK = WIN * WIN, NTM = (NTOK + 15) / 16, KSP = (NTM + 1) / 2, REM = NTOK - (NTM - 1) * 16;
  static_assert(WIN <= 8 && REM % 4 == 0, "in-window coordinates are packed in 3 bits; key padding is per lane group");
  constexpr int KS = C / 32;
  constexpr int NT = C / 16;
  constexpr int SLAB = 2 * KS;           // fragments per slab (== NT)
  constexpr int FRAGS_HEAD = 4 * SLAB;   // V, Q, K, P
  constexpr int HEAD_BYTES = FRAGS_HEAD * 1024;
  static_assert(FRAGS_HEAD % (AHEAD + 1) == 0, "the register fragment ring must line up across heads");
  constexpr int W2 = 2 * WIN - 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int wbytes = a.heads * HEAD_BYTES;                            // resident weights (+ AHEAD KiB read-ahead slack)
  float* s_qkvb = reinterpret_cast<float*>(smem + wbytes + AHEAD * 1024);  // [3C], q part pre-scaled
  float* s_pb = s_qkvb + 3 * C;                                       // [C]
  // relative-position bias (:121-124) in ACCUMULATOR layout, one f32x4 per lane and (head, key tile, query tile):
  // element r of lane (lg, l15) is bias[key kt*16+4lg+r][query qt*16+l15] * log2(e); keys beyond the window carry
  // -3e38.  One conflict-free ds_read_b128 gives the initial accumulator of S^T = K Q^T: no per-score add, no index
  // tables in registers.
  f32x4* s_bias = reinterpret_cast<f32x4*>(s_pb + C);                 // [heads][NTM][NTM][64]
  // LayerNorm1 weight / bias come from LDS (one broadcast ds_read_b128 per 4 channels) instead of through the address
  // unit once per window: 12 of the 45 vector-memory reads of a window.  (RES_COND_LDS: also the conditioning row of
  // the wave's graph as (1 + scale | shift), in a per-wave slot: measured slower, off.)
  float* s_lng = reinterpret_cast<float*>(s_bias + a.heads * NTM * NTM * 64);   // [C]
  float* s_lnb = s_lng + C;                                                   // [C]
#if RES_COND_LDS
  float* s_cond = s_lnb + C + (threadIdx.x >> 6) * (2 * C);                   // [NW][2C], this wave's slot
#endif

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int nwx = a.res / WIN, nW = nwx * nwx;
  const int total = a.B * nW;
  const int Ltok = a.res * a.res;
  constexpr float LOG2E = 1.44269504088896340736f;
  constexpr float QSCALE = 0.17677669529663687f * LOG2E;  // 32^-0.5 * log2(e): scores live in the log2 domain

  // ---- once per workgroup: weights -> LDS by LDS-DMA (1 KiB pieces round-robin over the waves), small tables ----
  {
    const unsigned smem0 = lds_addr(smem);
    const char* img_lane = reinterpret_cast<const char*>(a.img) + lane * 16;
    const int npieces = wbytes >> 10;
    for (int pc = wave; pc < npieces; pc += NW) lds_dma16(img_lane + (size_t)pc * 1024, smem0 + pc * 1024);
  }
  for (int i = tid; i < 3 * C; i += 64 * NW) s_qkvb[i] = a.qkv_b[i] * (i < C ? QSCALE : 1.0f);
  for (int i = tid; i < C; i += 64 * NW) {
    s_pb[i] = a.proj_b[i];
    s_lng[i] = a.ln_g[i];
    s_lnb[i] = a.ln_b[i];
  }
  for (int e = tid; e < a.heads * NTM * NTM * 64; e += 64 * NW) {
    const int ln = e & 63, t = e >> 6;
    const int qt = t % NTM, kt = (t / NTM) % NTM, hh = t / (NTM * NTM);
    const int i = min(qt * 16 + (ln & 15), NTOK - 1);
    const int iy = i / WIN, ix = i - iy * WIN;
    f32x4 v;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = kt * 16 + 4 * (ln >> 4) + r;
      const int jy = j / WIN, jx = j - jy * WIN;
      v[r] = j < NTOK ? a.rpb[((iy - jy + WIN - 1) * W2 + (ix - jx + WIN - 1)) * a.heads + hh] * LOG2E : -3.0e38f;
    }
    s_bias[e] = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // One issue group: fragment G of head h (compile-time G): {3 MFMAs, the ds_read of the fragment AHEAD later}.
#define GROUP(G, BODY)                                                                                  \
  {                                                                                                     \
    const bf16x8 w = fr[(G) % (AHEAD + 1)];                                                             \
    BODY;                                                                                               \
    fr[((G) + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const bf16x8*>(cb + ((G) + AHEAD) * 1024);      \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  }

  // ---- once per wave: in-window coordinates of the lane's query token i = mt*16 + l15 (key side: token
  // j = kt*16 + 4*lg + r, only needed by the region mask of border windows and by the bias table above) ----
  int iyx[NTM];
#pragma unroll
  for (int mt = 0; mt < NTM; ++mt) {
    const int i = min(mt * 16 + l15, NTOK - 1);
    const int iy = i / WIN, ix = i - iy * WIN;
    iyx[mt] = iy | (ix << 8);
  }

  // every wave walks a CONTIGUOUS run of windows (mostly one graph: its conditioning row is staged once per run)
  const int nwaves = gridDim.x * NW, gw = blockIdx.x * NW + wave;
#if RES_WALK_CONTIG
  const int per = total / nwaves, rem = total - per * nwaves;
  const int w_lo = gw * per + min(gw, rem), w_hi = w_lo + per + (gw < rem ? 1 : 0), w_step = 1;
#else
  const int w_lo = gw, w_hi = total, w_step = nwaves;
#endif
  int b_cached = -1;
  for (int wi_g = w_lo; wi_g < w_hi; wi_g += w_step) {
#ifdef SWG_STAMPS
    const int witer = (wi_g - w_lo) / w_step;
    const bool dbg = (witer == 3) && (blockIdx.x == 0 || blockIdx.x == 131) && (wave == 0 || wave == 5);
    const int dbgslot = (blockIdx.x == 0 ? 0 : 2) + (wave == 0 ? 0 : 1);
#endif
    STAMP(0);
    const int b = wi_g / nW, wi = wi_g - b * nW;
    const int wy = wi / nwx, wx = wi - wy * nwx;
    // region ids differ inside a window only in the last window row / column of a shifted block
    const bool need_mask = __builtin_amdgcn_readfirstlane((int)(a.shift > 0 && (wy == nwx - 1 || wx == nwx - 1))) != 0;
    // token rows of this window: shifted[y] = x[(y+s) mod H]  (roll by -s, :248)
    int tokq[NTM];
#pragma unroll
    for (int mt = 0; mt < NTM; ++mt) {
      int y = wy * WIN + (iyx[mt] & 255) + a.shift, x = wx * WIN + (iyx[mt] >> 8) + a.shift;
      if (y >= a.res) y -= a.res;
      if (x >= a.res) x -= a.res;
      tokq[mt] = b * Ltok + y * a.res + x;
    }
    // ---- rows in operand layout: conditioning + SiLU, LayerNorm1 -> bf16 fragments (:238-244) ----
    bf16x8 xa[NTM][KS];
    // The projection accumulators START as shortcut + proj bias: accumulator row 4*lg+r of out tile nt = 2*ks+hh is
    // channel 32ks + 8lg + 4hh + r (rho order of the image rows), i.e. exactly the channels this lane conditions
    // below for k-step ks, so the epilogue is a plain store (no second pass over x, no second SiLU).
    f32x4 acc[NT][NTM];
#if RES_COND_LDS
    if (b != b_cached) {   // wave-uniform: stage (1 + scale | shift) of graph b in the wave's LDS slot
      const float* crow = a.cond + (size_t)b * a.cond_stride + a.cond_off;
      for (int i = lane; i < 2 * C; i += 64) s_cond[i] = crow[i] + (i < C ? 1.0f : 0.0f);
      b_cached = b;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own wave only: no barrier needed
    }
    const float* cnd = s_cond;
    constexpr float ONE = 0.0f;
#else
    const float* cnd = a.cond + (size_t)b * a.cond_stride + a.cond_off;
    constexpr float ONE = 1.0f;
    (void)b_cached;
#endif
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
        const float4 sc0 = *reinterpret_cast<const float4*>(cnd + c0);        // 1 + scale (scale when read from memory)
        const float4 sc1 = *reinterpret_cast<const float4*>(cnd + c0 + 4);
        const float4 sh0 = *reinterpret_cast<const float4*>(cnd + C + c0);
        const float4 sh1 = *reinterpret_cast<const float4*>(cnd + C + c0 + 4);
        const float xv[8] = {p0.x, p0.y, p0.z, p0.w, p1.x, p1.y, p1.z, p1.w};
        const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
        const float sh[8] = {sh0.x, sh0.y, sh0.z, sh0.w, sh1.x, sh1.y, sh1.z, sh1.w};
        const float4 pb0 = *reinterpret_cast<const float4*>(s_pb + c0);
        const float4 pb1 = *reinterpret_cast<const float4*>(s_pb + c0 + 4);
        const float pb[8] = {pb0.x, pb0.y, pb0.z, pb0.w, pb1.x, pb1.y, pb1.z, pb1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = silu_f(fmaf(xv[j], sc[j] + ONE, sh[j]));
          x1[ks][j] = v;
          s += v;
          acc[2 * ks + (j >> 2)][mt][j & 3] = v + pb[j];
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
        const float4 g0 = *reinterpret_cast<const float4*>(s_lng + c0);
        const float4 g1 = *reinterpret_cast<const float4*>(s_lng + c0 + 4);
        const float4 b0 = *reinterpret_cast<const float4*>(s_lnb + c0);
        const float4 b1 = *reinterpret_cast<const float4*>(s_lnb + c0 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        bf16x8 f;
#pragma unroll
        for (int j = 0; j < 8; ++j) f[j] = (bf16)((x1[ks][j] - mean) * rstd * gg[j] + bb[j]);
        xa[mt][ks] = f;
      }
    }
    asm volatile("" ::: "memory");
    STAMP(1);
    bf16x8 fr[AHEAD + 1];
#pragma unroll
    for (int p = 0; p < AHEAD; ++p) fr[p] = *reinterpret_cast<const bf16x8*>(smem + lane * 16 + p * 1024);

#pragma nounroll
    for (int h = 0; h < a.heads; ++h) {
      const char* cb = smem + h * HEAD_BYTES + lane * 16;  // fragments are contiguous across heads
      STAMP(3 + 8 * h);
      // ---- V [tokens x 32] = A Wv^T (tokens on rows: directly the A operand of O^T = V^T P^T) ----
      bf16x8 qf[NTM], kf[NTM], vf[2][KSP];
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
        }
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
      STAMP(4 + 8 * h);
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
      STAMP(5 + 8 * h);
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

      STAMP(6 + 8 * h);
        // ---- S^T = K Q^T (log2 domain), softmax over keys (rows) per query (lane) ----
      f32x4 p[NTM][NTM];
      {
        const f32x4* bp = s_bias + (size_t)h * (NTM * NTM * 64) + lane;
#pragma unroll
        for (int kt = 0; kt < NTM; ++kt)
#pragma unroll
          for (int qt = 0; qt < NTM; ++qt) p[kt][qt] = bp[(kt * NTM + qt) * 64];
#pragma unroll
        for (int kt = 0; kt < NTM; ++kt)
#pragma unroll
          for (int qt = 0; qt < NTM; ++qt)
            p[kt][qt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt], qf[qt], p[kt][qt], 0, 0, 0);
      }
      if (need_mask) {  // wave-uniform: only the border windows of shifted blocks pay for the region mask (:207-226)
        auto region = [&](int iy, int ix) {
          const int ys = wy * WIN + iy, xs = wx * WIN + ix;
          const int ry = ys < a.res - WIN ? 0 : (ys < a.res - a.shift ? 1 : 2);
          const int rx = xs < a.res - WIN ? 0 : (xs < a.res - a.shift ? 1 : 2);
          return 3 * ry + rx;
        };
        int rq[NTM];
#pragma unroll
        for (int qt = 0; qt < NTM; ++qt) rq[qt] = region(iyx[qt] & 255, iyx[qt] >> 8);
#pragma unroll
        for (int kt = 0; kt < NTM; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = min(kt * 16 + 4 * lg + r, NTOK - 1);
            const int jy = j / WIN, jx = j - jy * WIN;
            const int rk = region(jy, jx);
#pragma unroll
            for (int qt = 0; qt < NTM; ++qt)
              if (rk != rq[qt]) p[kt][qt][r] += -100.0f * LOG2E;
          }
      }
      STAMP(7 + 8 * h);
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
        inv_sum[qt] = fast_rcp(sum);
      }

      STAMP(8 + 8 * h);
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
      }
      STAMP(9 + 8 * h);
        {
        // ---- out^T += Wp_h O^T (Wproj columns carry the head-dim slot order, rows the rho order, in the image) ----
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p2 = 0; p2 < SLAB; ++p2) {
          GROUP(3 * SLAB + p2, _Pragma("unroll") for (int mt = 0; mt < NTM; ++mt) acc[p2][mt] =
                                   __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, of[mt], acc[p2][mt], 0, 0, 0));
        }
      }
    }

    STAMP(27);
    // ---- epilogue: scatter back to token order (shortcut and bias are already in the accumulators) ----
#pragma unroll
    for (int mt = 0; mt < NTM; ++mt) {
      if (mt * 16 + l15 >= NTOK) continue;
      float* orow = a.out + (size_t)tokq[mt] * C;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        *reinterpret_cast<f32x4*>(orow + 32 * (nt >> 1) + 8 * lg + 4 * (nt & 1)) = acc[nt][mt];
    }
  }
#undef GROUP
}

}  // namespace

bool wattn_res_supported(int C, int heads, int win, int res) {
  return C == 96 && heads == 3 && (win == 4 || win == 6 || win == 8) && res % win == 0;
}

namespace {
#ifndef RES_NW8
#define RES_NW8 4   // waves per workgroup for 8x8 windows (A/B switch)
#endif
template <int WIN>
int launch_res_t(const ResAttnArgs& a, int total, hipStream_t s) {
  // 8x8 windows (64 tokens = 4 token tiles): with 8 waves per workgroup (two per SIMD, 256 registers each) the
  // instance spills 256 B per lane; 4 waves (one per SIMD, the 512-register budget) hold everything in registers.
  constexpr int C = 96, NW = (WIN == 8) ? RES_NW8 : 8, AHEAD = 3, NTM = (WIN * WIN + 15) / 16;
  const size_t lds = (size_t)a.heads * (4 * 2 * (C / 32)) * 1024 + AHEAD * 1024 + (size_t)(4 * C) * sizeof(float) +
                     (size_t)a.heads * NTM * NTM * 1024 +   // weights + read-ahead slack + biases + bias tiles
                     (size_t)(2 * C + NW * 2 * C) * sizeof(float);  // LayerNorm1 g / b + per-wave conditioning rows
  const int nwg = min(cdiv(total, NW), 256);
  static std::atomic<uint64_t> lds_ok{0};  // dynamic LDS above 64 KiB needs an explicit opt-in, per device
  if (int e = ensure_dyn_lds(lds_ok, (const void*)k_wattn_res<C, NW, AHEAD, WIN>, (int)((int)lds))) return e;
  hipLaunchKernelGGL((k_wattn_res<C, NW, AHEAD, WIN>), dim3(nwg), dim3(64 * NW), lds, s, a);
  return 0;
}
}  // namespace

int launch_wattn_res(const swg_block* blk, int B, const float* x, const float* cond, int cond_stride, float* out,
                     hipStream_t s) {
  if (!wattn_res_supported(blk->C, blk->heads, blk->win, blk->res) || !blk->attn_img) return SWG_E_UNSUPPORTED;
  constexpr int C = 96;
  const int nW = (blk->res / blk->win) * (blk->res / blk->win);
  ResAttnArgs a = {x, out, cond, cond_stride, blk->cond_off, blk->ln1.g, blk->ln1.b, (const bf16*)blk->attn_img,
                   blk->qkv.b, blk->proj.b, blk->rpb, B, blk->res, blk->shift, blk->heads};
  const int total = B * nW;
  const double T = (double)B * blk->res * blk->res;
  SwgProfScope ps(KID_WATTN_RES, 8.0 * T * C * C + 4.0 * T * (blk->win * blk->win) * C, 8.0 * T * C + 8.0 * C * C, s);
  int r;
  if (blk->win == 4) r = launch_res_t<4>(a, total, s);
  else if (blk->win == 8) r = launch_res_t<8>(a, total, s);
  else r = launch_res_t<6>(a, total, s);
  if (r) return r;
  SWG_LAUNCH_CHECK();
  return 0;
}
