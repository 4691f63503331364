// MFMA forms of the two 4x4-patch kernels at the ends of the network (p = 4, C0 = 96 -- the grid / DD geometry):
//   * PatchEmbed (swin_gnn.py:562-577 + self-cond concat :769-771 + c_in of precond.py:47): a [tokens x 32 taps] x
//     [32 x C] product is exactly ONE 16x16x32 k-step per 16 tokens x 16 channels;
//   * final LayerNorm + collapsed read-out (:748, :697-701): [tokens x C] x [C x 16 pixels].
// Both are HBM-bound (the fp32 token stream is 4*C bytes per token); the scalar-FMA versions in rowops.hip were
// VALU/LDS-bound at a quarter of the HBM rate.  The matrix core works on bf16, so every fp32 operand is split into
// hi + lo bf16 parts and the product is taken as hi*hi + lo*hi + hi*lo (relative error ~2^-16: fp32-grade for this
// path, whose other layers round activations to bf16 once).
// Persistent waves: weights / LN constants are converted to fragments once per wave, then the wave walks over
// 16-token tiles.  Token on the lane (l & 15), 8 consecutive taps / channels per lane group (l >> 4).
#include "common.h"

namespace {

struct EmbedMfmaArgs {
  const float* x; const float* sc; const float* c_in; int c_in_stride;
  const float* cond; int cond_stride, cond_off;
  const float* w; const float* bias; const float* g; const float* b;
  float* tok;
  int B, N, in_ch, self_cond;
};

__device__ __forceinline__ void split_bf16(const float (&f)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const bf16 h = (bf16)f[e];
    hi[e] = h;
    lo[e] = (bf16)(f[e] - (float)h);
  }
}

// channel held by accumulator row i of output tile nt: plain order, so that the four lane groups of one store
// instruction write 64 contiguous bytes per token (whole 32-byte sectors; the rho order of engine.py, 8 consecutive
// channels per lane over a tile pair, made every instruction write half sectors 32 bytes apart)
__device__ __forceinline__ int rho_chan(int nt, int i) { return 16 * nt + i; }

template <int C>
__global__ __launch_bounds__(256, 1) void k_patch_embed_mfma(EmbedMfmaArgs a) {
  constexpr int NT = C / 16;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int kin = a.in_ch * 16;

  bf16x8 wh[NT], wl[NT];
  f32x4 cb[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c = rho_chan(nt, l15);
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * lg + e;
      f[e] = (k < kin) ? a.w[c * kin + k] : 0.f;
    }
    split_bf16(f, wh[nt], wl[nt]);
    cb[nt] = *reinterpret_cast<const f32x4*>(a.bias + rho_chan(nt, 4 * lg));
  }

  const int hp = a.N >> 2, L = hp * hp, rows = a.B * L;
  const int ntiles = (rows + 15) >> 4;
  // this lane's taps: channel ci = lg >> 1 of cat([self_cond, x]), patch rows ii0, ii0 + 1
  const int ci = lg >> 1, ii0 = (lg & 1) * 2;
  const bool has = ci < a.in_ch;
  const bool is_x = (!a.self_cond) || ci == a.in_ch - 1;
  const float* src = is_x ? a.x : a.sc;
  const bool live = has && src != nullptr;

  // Few, long-lived waves (two workgroups per CU), each with a CONTIGUOUS run of tiles: the per-graph epilogue
  // constants  y = silu(A * xhat + B),  A = g (1 + scale),  B = shift + beta (1 + scale)  then stay in registers and are
  // reloaded only when a lane's token moves to the next graph.  (Per-tile reloads of scale / shift / g / beta were 30 of
  // the 45 vector-memory instructions of a tile and kept the address unit as busy as the token stream itself; waves
  // of 4 tiles paid their start-up -- weight gather, first tap fetch -- once per 4 tiles.)
  const int nwaves = gridDim.x * 4, gw = blockIdx.x * 4 + wave;
  const int tpw = (ntiles + nwaves - 1) / nwaves;
  const int t0 = gw * tpw, t1 = min(ntiles, t0 + tpw);
  f32x4 cA[NT], cB[NT];
  int cur_bi = -1;
  // the taps are fetched D tiles ahead (a load issued after the stores of the previous tile completes after them:
  // vector-memory operations retire in issue order); D = 1 and D = 4 measured the same, the kernel is bound by its
  // ~650 instructions per tile with the store issue on top
  constexpr int D = 2;
  float4 n0[D], n1[D];
  float nsc[D];
  auto fetch = [&](int tile, float4& q0, float4& q1, float& qs) {
    if (!live || tile >= t1) return;
    const int row = min(tile * 16 + l15, rows - 1);
    const int bi = row / L, rem = row - bi * L;
    const int Y = rem / hp, X = rem - Y * hp;
    const float* p0 = src + ((size_t)bi * a.N + (Y * 4 + ii0)) * a.N + X * 4;
    q0 = *reinterpret_cast<const float4*>(p0);
    q1 = *reinterpret_cast<const float4*>(p0 + a.N);
    qs = (is_x && a.c_in) ? a.c_in[(size_t)bi * a.c_in_stride] : 1.0f;
  };
#pragma unroll
  for (int d = 0; d < D; ++d) {
    n0[d] = make_float4(0.f, 0.f, 0.f, 0.f); n1[d] = n0[d]; nsc[d] = 1.0f;
    fetch(t0 + d, n0[d], n1[d], nsc[d]);
  }
  for (int tb = t0; tb < t1; tb += D)
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int tile = tb + d;
    if (tile >= t1) break;
    const int row = min(tile * 16 + l15, rows - 1);
    const int bi = row / L;
    const float4 r0 = n0[d], r1 = n1[d];
    const float sc = nsc[d];
    fetch(tile + D, n0[d], n1[d], nsc[d]);
    if (bi != cur_bi) {
      cur_bi = bi;
      const float* crow = a.cond + (size_t)bi * a.cond_stride + a.cond_off;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = rho_chan(nt, 4 * lg);
        const f32x4 scl = *reinterpret_cast<const f32x4*>(crow + c);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(crow + C + c);
        const f32x4 cg = *reinterpret_cast<const f32x4*>(a.g + c);
        const f32x4 cbeta = *reinterpret_cast<const f32x4*>(a.b + c);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          cA[nt][r] = cg[r] * (scl[r] + 1.f);
          cB[nt][r] = sh[r] + cbeta[r] * (scl[r] + 1.f);
        }
      }
    }
    const float f[8] = {sc * r0.x, sc * r0.y, sc * r0.z, sc * r0.w, sc * r1.x, sc * r1.y, sc * r1.z, sc * r1.w};
    bf16x8 xh, xl;
    split_bf16(f, xh, xl);
    f32x4 acc[NT];
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[nt], xh, cb[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[nt], xl, acc[nt], 0, 0, 0);
      acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[nt], xh, acc[nt], 0, 0, 0);
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) s += (acc[nt][0] + acc[nt][1]) + (acc[nt][2] + acc[nt][3]);
    s = rows_sum(s);
    const float mean = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = acc[nt][r] - mean; q += d * d; }
    q = rows_sum(q);
    const float rstd = ln_rstd(q * (1.0f / C));
    if (tile * 16 + l15 < rows) {
      float* orow = a.tok + (size_t)row * C;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        f32x4 y;
#pragma unroll
        for (int r = 0; r < 4; ++r) y[r] = silu_f(cB[nt][r] + ((acc[nt][r] - mean) * rstd) * cA[nt][r]);
        *reinterpret_cast<f32x4*>(orow + rho_chan(nt, 4 * lg)) = y;
      }
    }
  }
}

struct ReadoutMfmaArgs {
  const float* tok; const float* g; const float* b; const float* w; float bias; float* F;
  int B, N;
};

template <int C>
__global__ __launch_bounds__(256, 4) void k_final_readout_mfma(ReadoutMfmaArgs a) {
  constexpr int KS = C / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lg = lane >> 4;

  bf16x8 wh[KS], wl[KS];  // A operand: row = output pixel o = l15 (4x4 patch, o = 4*i + j), k = channel
  float cg[KS][8], cbeta[KS][8];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    float f[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = 32 * ks + 8 * lg + e;
      f[e] = a.w[c * 16 + l15];
      cg[ks][e] = a.g[c];
      cbeta[ks][e] = a.b[c];
    }
    split_bf16(f, wh[ks], wl[ks]);
  }
  const int hp = a.N >> 2, L = hp * hp, rows = a.B * L;
  const int ntiles = (rows + 15) >> 4;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const int row = min(tile * 16 + l15, rows - 1);
    const float* xr = a.tok + (size_t)row * C + 8 * lg;
    float v[KS][8];
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 p0 = *reinterpret_cast<const float4*>(xr + 32 * ks);
      const float4 p1 = *reinterpret_cast<const float4*>(xr + 32 * ks + 4);
      v[ks][0] = p0.x; v[ks][1] = p0.y; v[ks][2] = p0.z; v[ks][3] = p0.w;
      v[ks][4] = p1.x; v[ks][5] = p1.y; v[ks][6] = p1.z; v[ks][7] = p1.w;
#pragma unroll
      for (int e = 0; e < 8; ++e) s += v[ks][e];
    }
    s = rows_sum(s);
    const float mean = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int e = 0; e < 8; ++e) { const float d = v[ks][e] - mean; q += d * d; }
    q = rows_sum(q);
    const float rstd = ln_rstd(q * (1.0f / C));
    f32x4 acc = {a.bias, a.bias, a.bias, a.bias};
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float y[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) y[e] = (v[ks][e] - mean) * rstd * cg[ks][e] + cbeta[ks][e];
      bf16x8 yh, yl;
      split_bf16(y, yh, yl);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], yh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[ks], yl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[ks], yh, acc, 0, 0, 0);
    }
    if (tile * 16 + l15 < rows) {
      // accumulator rows 4*lg .. 4*lg+3 of this token = pixels (i = lg, j = 0..3) of its patch: one float4 store
      const int bi = row / L, rem = row - bi * L;
      const int Y = rem / hp, X = rem - Y * hp;
      *reinterpret_cast<f32x4*>(a.F + ((size_t)bi * a.N + (Y * 4 + lg)) * a.N + X * 4) = acc;
    }
  }
}

// PatchBreakup tail (swin_gnn.py:385-403) in ONE pass for Co = 96 (Cin = 384):
//   y row [Cin] -> norm LayerNorm(Cin) -> 4 sub-vectors of Co -> 2x2 scatter -> post_norm LayerNorm(Co) -> post_linear (no
//   bias) -> fp32 token rows.  The 96x96 weight lives in registers as 18 MFMA fragments per wave (persistent waves);
//   a wave takes 16 source tokens at a time, token on the lane.  Replaces k_breakup_ln2 + a [4T,96]x[96,96] GEMM whose
//   bf16 operand round trip cost more than the product.
struct BreakupPostArgs {
  const float* y; const float* g1; const float* b1; const float* g2; const float* b2;
  const bf16* w;   // post_linear weight, bf16 [Co, ldw]
  int ldw;
  float* out;      // [4 * rows, Co]
  int rows, res;
};

template <int CO>
__global__ __launch_bounds__(256, 2) void k_breakup_post(BreakupPostArgs a) {
  constexpr int KS = CO / 32, NT = CO / 16, CIN = 4 * CO;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  bf16x8 wf[NT][KS];  // A operand: row = out channel rho(nt*16 + l15), k-slots = input channels 32ks + 8lg ..+7
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
      wf[nt][ks] = *reinterpret_cast<const bf16x8*>(a.w + (size_t)rho_chan(nt, l15) * a.ldw + 32 * ks + 8 * lg);
  // the affine parameters of both LayerNorms (2 * Cin + 2 * Co floats) are read from LDS: as global loads they were 96
  // of a tile's 144 vector-memory instructions
  __shared__ __align__(16) float cst[2 * CIN + 2 * CO];
  for (int i = threadIdx.x; i < CIN; i += 256) { cst[i] = a.g1[i]; cst[CIN + i] = a.b1[i]; }
  for (int i = threadIdx.x; i < CO; i += 256) { cst[2 * CIN + i] = a.g2[i]; cst[2 * CIN + CO + i] = a.b2[i]; }
  __syncthreads();
  const int cst_lane = 8 * lg;
  const int ntiles = (a.rows + 15) >> 4;
  const int L = a.res * a.res;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const int row = min(tile * 16 + l15, a.rows - 1);
    const float* src = a.y + (size_t)row * CIN + 8 * lg;
    float v[4][KS][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const f32x4 p0 = *reinterpret_cast<const f32x4*>(src + k * CO + 32 * ks);
        const f32x4 p1 = *reinterpret_cast<const f32x4*>(src + k * CO + 32 * ks + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[k][ks][e] = p0[e]; v[k][ks][4 + e] = p1[e]; s += p0[e] + p1[e]; }
      }
    const float mean = rows_sum(s) * (1.0f / CIN);
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[k][ks][e] - mean; q += d * d; }
    const float rstd = ln_rstd(rows_sum(q) * (1.0f / CIN));
    const int bi = row / L, rem = row - bi * L;
    const int yy = rem / a.res, xx = rem - yy * a.res;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // (an opaque offset per quarter keeps the compiler from hoisting all 240 loop-invariant LDS reads into registers)
      int zoff = cst_lane;
      asm volatile("" : "+v"(zoff));
      const float* lg1 = cst + zoff; const float* lb1 = cst + CIN + zoff;
      const float* lg2 = cst + 2 * CIN + zoff; const float* lb2 = cst + 2 * CIN + CO + zoff;
      // norm (its affine parameters are indexed by the position in the Cin row), then the sub-vector statistics
      float s2 = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const f32x4 ga = *reinterpret_cast<const f32x4*>(lg1 + k * CO + 32 * ks);
        const f32x4 gb = *reinterpret_cast<const f32x4*>(lg1 + k * CO + 32 * ks + 4);
        const f32x4 ba = *reinterpret_cast<const f32x4*>(lb1 + k * CO + 32 * ks);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(lb1 + k * CO + 32 * ks + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[k][ks][e] = (v[k][ks][e] - mean) * rstd * ga[e] + ba[e];
          v[k][ks][4 + e] = (v[k][ks][4 + e] - mean) * rstd * gb[e] + bb[e];
          s2 += v[k][ks][e] + v[k][ks][4 + e];
        }
      }
      const float mean2 = rows_sum(s2) * (1.0f / CO);
      float q2 = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int e = 0; e < 8; ++e) { const float d = v[k][ks][e] - mean2; q2 += d * d; }
      const float rstd2 = ln_rstd(rows_sum(q2) * (1.0f / CO));
      f32x4 acc[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) acc[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const f32x4 ga = *reinterpret_cast<const f32x4*>(lg2 + 32 * ks);
        const f32x4 gb = *reinterpret_cast<const f32x4*>(lg2 + 32 * ks + 4);
        const f32x4 ba = *reinterpret_cast<const f32x4*>(lb2 + 32 * ks);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(lb2 + 32 * ks + 4);
        bf16x8 f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f[e] = (bf16)((v[k][ks][e] - mean2) * rstd2 * ga[e] + ba[e]);
          f[4 + e] = (bf16)((v[k][ks][4 + e] - mean2) * rstd2 * gb[e] + bb[e]);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[nt][ks], f, acc[nt], 0, 0, 0);
      }
      if (tile * 16 + l15 < a.rows) {
        // quarter k of token (b, y, x) is sub-token (2y + dy, 2x + dx), (dy, dx) = (k & 1, k >> 1): swin_gnn.py:389-397
        const size_t drow = (size_t)bi * 4 * L + (size_t)(2 * yy + (k & 1)) * 2 * a.res + (2 * xx + (k >> 1));
        float* orow = a.out + drow * CO;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) *reinterpret_cast<f32x4*>(orow + rho_chan(nt, 4 * lg)) = acc[nt];
      }
    }
  }
}

int persistent_grid(int ntiles, int g = 8) { return max(1, min(cdiv(ntiles, 4), 256 * g)); }

}  // namespace

bool embed_mfma_supported(const swg_net* net) {
  return net->p == 4 && net->C0 == 96 && net->N % 4 == 0 && net->in_ch >= 1 && net->in_ch <= 2;
}

int launch_patch_embed_mfma(const swg_net* net, int B, const float* x, const float* sc, const float* c_in,
                            int c_in_stride, const float* cond, int cond_stride, float* tok, hipStream_t s) {
  EmbedMfmaArgs a = {x, sc, c_in, c_in_stride, cond, cond_stride, net->pe_cond_off, net->pe_w, net->pe_b,
                     net->pe_ln.g, net->pe_ln.b, tok, B, net->N, net->in_ch, net->self_cond};
  const int hp = net->N / 4, rows = B * hp * hp;
  SwgProfScope ps(KID_PATCH_EMBED, 2.0 * rows * 96.0 * 16 * net->in_ch,
                  4.0 * B * (double)net->N * net->N * (sc ? 2 : 1) + 4.0 * rows * 96.0, s);
  // one workgroup per CU (measured: 105 us per launch of 56 graphs with 8 short-lived workgroups per CU, 75 us with one)
  hipLaunchKernelGGL((k_patch_embed_mfma<96>), dim3(persistent_grid(cdiv(rows, 16), 1)), dim3(256), 0, s, a);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_final_readout_mfma(const swg_net* net, int B, const float* tok, float* F, hipStream_t s) {
  ReadoutMfmaArgs a = {tok, net->norm.g, net->norm.b, net->ro_w, net->ro_b, F, B, net->N};
  const int hp = net->N / 4, rows = B * hp * hp;
  SwgProfScope ps(KID_READOUT, 2.0 * rows * 96.0 * 16, 4.0 * rows * 96.0 + 4.0 * B * (double)net->N * net->N, s);
  hipLaunchKernelGGL((k_final_readout_mfma<96>), dim3(persistent_grid(cdiv(rows, 16))), dim3(256), 0, s, a);
  SWG_LAUNCH_CHECK();
  return 0;
}

bool breakup_post_supported(int Co) { return Co == 96; }

int launch_breakup_post(const float* y, int B, int res, int Co, const float* g1, const float* b1, const float* g2,
                        const float* b2, const bf16* w, int ldw, float* out, hipStream_t s) {
  if (!breakup_post_supported(Co) || ldw % 8) return SWG_E_UNSUPPORTED;
  BreakupPostArgs a = {y, g1, b1, g2, b2, w, ldw, out, B * res * res, res};
  if (a.rows <= 0) return 0;
  SwgProfScope ps(KID_BREAKUP_POST, 2.0 * 4.0 * a.rows * (double)Co * Co, 2.0 * 4.0 * a.rows * 4.0 * Co, s);
  hipLaunchKernelGGL((k_breakup_post<96>), dim3(persistent_grid((a.rows + 15) / 16)), dim3(256), 0, s, a);
  SWG_LAUNCH_CHECK();
  return 0;
}
