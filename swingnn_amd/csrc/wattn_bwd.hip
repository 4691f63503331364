// Backward of the window-attention core (model/swin_gnn/swin_gnn.py:116-136 between the two Linear layers) on the
// packed qkv buffer in TOKEN order, the partition / shift folded into addressing exactly as in wattn.hip.
//
//   S = scale Q K^T + bias + mask, P = softmax(S), O = P V
//   dV = P^T dO,  dP = dO V^T,  dS = P (dP - rowsum(P dP)),  dQ = scale dS K,  dK = scale dS^T Q,
//   d(relative_position_bias_table)[idx(i,j), h] += dS[i,j]
//
// One wavefront per (window, head), all five contractions on v_mfma_f32_16x16x32_bf16 (head dim padded to 32, the
// window's tokens to NT = 16 NTL rows):
//  * Q, K, V, dO rows go global -> registers once (8 head-dim values per lane = an MFMA operand as loaded) and Q, K,
//    dO also to token-major LDS tiles.
//  * Phase A, per 16-query tile: S^T = K Q^T and dP^T = V dO^T put the QUERY on the lane (i = lane & 15) and four
//    consecutive keys in the accumulator registers, so the softmax statistics (max, sum, delta = sum_j P dP) are
//    in-lane plus two cross-group swaps.  dS in that layout IS the B operand of dQ^T = K^T dS^T (contraction slots
//    permuted: slot e of lane group g = key 16 jt + 4 g + e); the matching K^T operand is gathered from the token-major
//    LDS tile with ds_read_b64_tr_b16.  The bias-table gradient is scattered into an LDS copy of the table.
//  * Phase B, per 16-key tile: S and dP recomputed with swapped operands (KEY on the lane), P and dS rebuilt from
//    the per-query statistics (LDS), then dK^T = Q^T dS and dV^T = dO^T P the same way.
// A workgroup keeps its head fixed and walks over windows, so the bias-table gradient is flushed with one atomic per
// entry per workgroup.
#include "common.h"

namespace {

struct WbArgs {
  const bf16* qkv; int ldq;
  const bf16* dO; int ldo;
  bf16* dqkv; int lddq;
  const float* rpb;
  float* drpb;
  int B, res, win, shift, heads, C, hd;
  float scale;
};

constexpr int LDT = 48;     // LDS row stride (elements) of the token-major tiles: 8 rows x 32 B cover 64 banks once
constexpr int NBMAX = 232;  // >= (2*8-1)^2

typedef __attribute__((address_space(3))) bf16x4 lds_b4;
__device__ __forceinline__ bf16x4 tr4(const bf16* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)p); }

// 8 head-dim values of one row (two 8-byte pieces; `rem` = values left in the head from this lane's first column)
__device__ __forceinline__ bf16x8 load8(const bf16* p, bool valid, int rem) {
  uint2 lo = make_uint2(0u, 0u), hi = make_uint2(0u, 0u);
  if (valid && rem >= 4) lo = *reinterpret_cast<const uint2*>(p);
  if (valid && rem >= 8) hi = *reinterpret_cast<const uint2*>(p + 4);
  uint4 v = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return *reinterpret_cast<bf16x8*>(&v);
}
__device__ __forceinline__ bf16x8 pack8(const f32x4& a, const f32x4& b) {
  return (bf16x8){(bf16)a[0], (bf16)a[1], (bf16)a[2], (bf16)a[3], (bf16)b[0], (bf16)b[1], (bf16)b[2], (bf16)b[3]};
}
__device__ __forceinline__ bf16x8 cat8(const bf16x4& a, const bf16x4& b) {
  return (bf16x8){a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
}

template <int NTL, bool SH>
__global__ __launch_bounds__(64) void k_wattn_bwd(WbArgs a) {
  constexpr int NT = 16 * NTL;
  constexpr int KP = (NTL + 1) / 2;  // 32-deep contraction steps over the window's tokens
  __shared__ __attribute__((aligned(16))) bf16 Qs[NT * LDT];
  __shared__ __attribute__((aligned(16))) bf16 Ks[NT * LDT];
  __shared__ __attribute__((aligned(16))) bf16 Ds[NT * LDT];
  __shared__ float bias[NBMAX];
  __shared__ float dbias[NBMAX];
  __shared__ __attribute__((aligned(16))) float st_m[64];
  __shared__ __attribute__((aligned(16))) float st_inv[64];
  __shared__ __attribute__((aligned(16))) float st_delta[64];
  __shared__ __attribute__((aligned(16))) int rid[64];
  __shared__ __attribute__((aligned(16))) int pos[64];

  const int lane = threadIdx.x, l15 = lane & 15, lg = lane >> 4, q4 = l15 >> 2, p4 = l15 & 3;
  const int h = blockIdx.x;
  const int n = a.win * a.win, w2 = 2 * a.win - 1, nb = w2 * w2;
  const int nwx = a.res / a.win, nW = nwx * nwx, Ltok = a.res * a.res;
  const int hc = h * a.hd;
  const int cpos = (a.win - 1) * (w2 + 1);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const bf16x4 zb4 = {(bf16)0.f, (bf16)0.f, (bf16)0.f, (bf16)0.f};
  for (int t = lane; t < nb; t += 64) {
    bias[t] = a.rpb[(size_t)t * a.heads + h];
    dbias[t] = 0.f;
  }
  {  // index (iy - jy + w - 1) (2w - 1) + (ix - jx + w - 1) = cpos + pos[i] - pos[j]   (swin_gnn.py:88-97)
    const int t = min(lane, n - 1);
    const int ty = t / a.win;
    pos[lane] = ty * w2 + (t - ty * a.win);
  }
  // this lane's own rows: local token 16 t + l15 of every tile t
  int ry_[NTL], rx_[NTL], rpos[NTL];
  bool rvalid[NTL];
#pragma unroll
  for (int t = 0; t < NTL; ++t) {
    const int il = min(16 * t + l15, n - 1);
    ry_[t] = il / a.win;
    rx_[t] = il - ry_[t] * a.win;
    rpos[t] = ry_[t] * w2 + rx_[t];
    rvalid[t] = 16 * t + l15 < n;
  }
  const int rem = a.hd - 8 * lg;
  const int troff = (4 * lg + q4) * LDT + 4 * p4;  // transposed-read address of this lane inside a 16-row tile

  // dS summed over this workgroup's windows, in the phase-A register layout; scattered into the bias table once at the
  // end (an LDS atomic per element and window kept every wave waiting on the LDS queue: 47 % of the wave cycles)
  constexpr bool REGACC = NTL <= 3;  // 64-token windows keep the per-window LDS scatter: 64 more registers halve their occupancy
  f32x4 dsacc[REGACC ? NTL : 1][REGACC ? NTL : 1];
#pragma unroll
  for (int i = 0; i < (REGACC ? NTL : 1); ++i)
#pragma unroll
    for (int j = 0; j < (REGACC ? NTL : 1); ++j) dsacc[i][j] = zero4;

  for (int u = blockIdx.y; u < a.B * nW; u += gridDim.y) {
    const int b = u / nW, wi = u - b * nW;
    const int wy = wi / nwx, wx = wi - wy * nwx;
    size_t tok[NTL];
    int myrid[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const int ys = wy * a.win + ry_[t], xs = wx * a.win + rx_[t];
      int y = ys + a.shift, x = xs + a.shift;
      if (y >= a.res) y -= a.res;
      if (x >= a.res) x -= a.res;
      tok[t] = (size_t)b * Ltok + y * a.res + x;
      myrid[t] = 0;
      if (SH) {
        const int ry = ys < a.res - a.win ? 0 : (ys < a.res - a.shift ? 1 : 2);
        const int rx = xs < a.res - a.win ? 0 : (xs < a.res - a.shift ? 1 : 2);
        myrid[t] = 3 * ry + rx;
      }
    }
    bf16x8 fq[NTL], fk[NTL], fv[NTL], fd[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const bf16* row = a.qkv + tok[t] * a.ldq + hc + 8 * lg;
      fq[t] = load8(row, rvalid[t], rem);
      fk[t] = load8(row + a.C, rvalid[t], rem);
      fv[t] = load8(row + 2 * a.C, rvalid[t], rem);
      fd[t] = load8(a.dO + tok[t] * a.ldo + hc + 8 * lg, rvalid[t], rem);
    }
    __syncthreads();  // the previous window's LDS reads are done
#pragma unroll
    for (int t = 0; t < NTL; ++t) {
      const int o = (16 * t + l15) * LDT + 8 * lg;
      *reinterpret_cast<bf16x8*>(&Qs[o]) = fq[t];
      *reinterpret_cast<bf16x8*>(&Ks[o]) = fk[t];
      *reinterpret_cast<bf16x8*>(&Ds[o]) = fd[t];
      if (lg == 0) rid[16 * t + l15] = myrid[t];
    }
    __syncthreads();

    // ---------------- phase A: query on the lane -> statistics, dQ, bias-table gradient ----------------
#pragma unroll
    for (int it = 0; it < NTL; ++it) {
      f32x4 sv[NTL], dp[NTL];
      int pjs[REGACC ? 1 : NTL][4];
#pragma unroll
      for (int jt = 0; jt < NTL; ++jt) {
        sv[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fk[jt], fq[it], zero4, 0, 0, 0);
        dp[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fv[jt], fd[it], zero4, 0, 0, 0);
      }
      const int pi = cpos + rpos[it];
      float mx = -3.0e38f;
#pragma unroll
      for (int jt = 0; jt < NTL; ++jt) {
        const int4 pj = *reinterpret_cast<const int4*>(&pos[16 * jt + 4 * lg]);
        const int4 rj = *reinterpret_cast<const int4*>(&rid[16 * jt + 4 * lg]);
        const int pjv[4] = {pj.x, pj.y, pj.z, pj.w}, rjv[4] = {rj.x, rj.y, rj.z, rj.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int id = pi - pjv[r];
          if constexpr (!REGACC) pjs[jt][r] = pjv[r];
          float v = sv[jt][r] * a.scale + bias[id];
          if (SH && rjv[r] != myrid[it]) v += -100.0f;  // swin_gnn.py:226
          if (jt == NTL - 1 && 16 * jt + 4 * lg + r >= n) v = -3.0e38f;  // padding keys: only in the last tile
          sv[jt][r] = v;
          mx = fmaxf(mx, v);
        }
      }
      mx = rows_max(mx);
      float sum = 0.f;
#pragma unroll
      for (int jt = 0; jt < NTL; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __expf(sv[jt][r] - mx);
          sv[jt][r] = e;
          sum += e;
        }
      sum = rows_sum(sum);
      const float inv = rvalid[it] ? 1.0f / sum : 0.f;  // padding queries: P = dS = 0
      float delta = 0.f;
#pragma unroll
      for (int jt = 0; jt < NTL; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = sv[jt][r] * inv;
          sv[jt][r] = p;
          delta += p * dp[jt][r];
        }
      delta = rows_sum(delta);
      if (lg == 0) {
        st_m[16 * it + l15] = mx;
        st_inv[16 * it + l15] = inv;
        st_delta[16 * it + l15] = delta;
      }
#pragma unroll
      for (int jt = 0; jt < NTL; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float ds = sv[jt][r] * (dp[jt][r] - delta);
          dp[jt][r] = ds;
          if constexpr (REGACC) dsacc[it][jt][r] += ds;
          else atomicAdd(&dbias[pi - pjs[jt][r]], ds);
        }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        f32x4 acc = zero4;
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
          const int j0 = 2 * kp, j1 = 2 * kp + 1;
          const bf16x8 bop = pack8(dp[j0], j1 < NTL ? dp[j1 < NTL ? j1 : 0] : zero4);
          const bf16x4 a0 = tr4(&Ks[16 * j0 * LDT + troff + 16 * dt]);
          const bf16x4 a1 = j1 < NTL ? tr4(&Ks[16 * (j1 < NTL ? j1 : 0) * LDT + troff + 16 * dt]) : zb4;
          acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(a0, a1), bop, acc, 0, 0, 0);
        }
        if (rvalid[it] && 16 * dt + 4 * lg < a.hd)
          *reinterpret_cast<uint2*>(a.dqkv + tok[it] * a.lddq + hc + 16 * dt + 4 * lg) =
              pack_bf16x4(acc[0] * a.scale, acc[1] * a.scale, acc[2] * a.scale, acc[3] * a.scale);
      }
    }
    __syncthreads();  // statistics visible

    // ---------------- phase B: key on the lane -> dK, dV ----------------
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) {
      f32x4 sv[NTL], dp[NTL];
#pragma unroll
      for (int it = 0; it < NTL; ++it) {
        sv[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fq[it], fk[jt], zero4, 0, 0, 0);
        dp[it] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd[it], fv[jt], zero4, 0, 0, 0);
      }
      const int pj = rpos[jt] - cpos;
#pragma unroll
      for (int it = 0; it < NTL; ++it) {
        const int4 pi4 = *reinterpret_cast<const int4*>(&pos[16 * it + 4 * lg]);
        const int4 ri4 = *reinterpret_cast<const int4*>(&rid[16 * it + 4 * lg]);
        const float4 m4 = *reinterpret_cast<const float4*>(&st_m[16 * it + 4 * lg]);
        const float4 i4 = *reinterpret_cast<const float4*>(&st_inv[16 * it + 4 * lg]);
        const float4 d4 = *reinterpret_cast<const float4*>(&st_delta[16 * it + 4 * lg]);
        const int piv[4] = {pi4.x, pi4.y, pi4.z, pi4.w}, riv[4] = {ri4.x, ri4.y, ri4.z, ri4.w};
        const float mv[4] = {m4.x, m4.y, m4.z, m4.w}, iv[4] = {i4.x, i4.y, i4.z, i4.w}, dv_[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = sv[it][r] * a.scale + bias[piv[r] - pj];
          if (SH && riv[r] != myrid[jt]) v += -100.0f;
          float p = __expf(v - mv[r]) * iv[r];
          if (jt == NTL - 1 && !rvalid[jt]) p = 0.f;  // padding keys
          sv[it][r] = p;
          dp[it][r] = p * (dp[it][r] - dv_[r]);
        }
      }
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        f32x4 ak = zero4, av = zero4;
#pragma unroll
        for (int kp = 0; kp < KP; ++kp) {
          const int i0 = 2 * kp, i1 = 2 * kp + 1;
          const int i1c = i1 < NTL ? i1 : 0;
          const bf16x8 bds = pack8(dp[i0], i1 < NTL ? dp[i1c] : zero4);
          const bf16x8 bp = pack8(sv[i0], i1 < NTL ? sv[i1c] : zero4);
          const bf16x4 q0 = tr4(&Qs[16 * i0 * LDT + troff + 16 * dt]);
          const bf16x4 q1 = i1 < NTL ? tr4(&Qs[16 * i1c * LDT + troff + 16 * dt]) : zb4;
          const bf16x4 g0 = tr4(&Ds[16 * i0 * LDT + troff + 16 * dt]);
          const bf16x4 g1 = i1 < NTL ? tr4(&Ds[16 * i1c * LDT + troff + 16 * dt]) : zb4;
          ak = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(q0, q1), bds, ak, 0, 0, 0);
          av = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cat8(g0, g1), bp, av, 0, 0, 0);
        }
        if (rvalid[jt] && 16 * dt + 4 * lg < a.hd) {
          bf16* o = a.dqkv + tok[jt] * a.lddq + hc + 16 * dt + 4 * lg;
          *reinterpret_cast<uint2*>(o + a.C) = pack_bf16x4(ak[0] * a.scale, ak[1] * a.scale, ak[2] * a.scale, ak[3] * a.scale);
          *reinterpret_cast<uint2*>(o + 2 * a.C) = pack_bf16x4(av[0], av[1], av[2], av[3]);
        }
      }
      if (h == 0 && lg == 0 && rvalid[jt])  // K padding of the following GEMM operand
        for (int c = 3 * a.C; c < a.lddq; c += 4) *reinterpret_cast<uint2*>(a.dqkv + tok[jt] * a.lddq + c) = make_uint2(0u, 0u);
    }
  }
  if constexpr (REGACC)
#pragma unroll
  for (int it = 0; it < NTL; ++it)
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) {
      const int4 pj = *reinterpret_cast<const int4*>(&pos[16 * jt + 4 * lg]);
      const int pjv[4] = {pj.x, pj.y, pj.z, pj.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) atomicAdd(&dbias[cpos + rpos[it] - pjv[r]], dsacc[it][jt][r]);
    }
  __syncthreads();
  for (int t = lane; t < nb; t += 64) {
    const float v = dbias[t];
    if (v != 0.f) atomicAdd(a.drpb + (size_t)t * a.heads + h, v);
  }
}

}  // namespace

int launch_wattn_bwd(const bf16* qkv, int ldq, const bf16* dO, int ldo, int B, int res, int win, int shift, int heads,
                     int C, const float* rpb, bf16* dqkv, int lddq, float* drpb, hipStream_t s) {
  if (win <= 0 || res % win || C % heads) return SWG_E_SHAPE;
  const int n = win * win, hd = C / heads;
  if (n > 64 || hd > 32 || hd % 4 || ldq % 4 || ldo % 4 || lddq % 4 || C % 4 || win > 8) return SWG_E_UNSUPPORTED;
  if (shift < 0 || shift >= win) return SWG_E_SHAPE;
  if (!qkv || !dO || !dqkv || !rpb || !drpb) return SWG_E_NULL;
  WbArgs a = {qkv, ldq, dO, ldo, dqkv, lddq, rpb, drpb, B, res, win, shift, heads, C, hd, 1.0f / sqrtf((float)hd)};
  const int nW = (res / win) * (res / win);
  // A workgroup (= one wave) walks over windows of its head.  About `target` workgroups in all, and every one the same
  // number k of windows (gy = ceil(B nW / k)): with 4096 / heads rows the C = 384 stage of grid B = 80 gave its
  // workgroups 4 or 3 windows (1280 windows over 342 rows) and the kernel ended on a quarter-empty tail.
  static const int target = [] { const char* v = getenv("SWG_WBWD_WGS"); return v ? atoi(v) : 3072; }();
  const int k = max(1, (int)(((long)B * nW * heads + target / 2) / target));
  const int gy = cdiv(B * nW, k);
  const double T = (double)B * res * res;
  SwgProfScope ps(KID_WATTN_BWD, 10.0 * n * C * T, T * (3.0 * C * 2 + C * 2 + 3.0 * C * 2), s);
  const dim3 grid(heads, gy), block(64);
#define WB_LAUNCH(NTL)                                                                     \
  if (shift > 0) hipLaunchKernelGGL((k_wattn_bwd<NTL, true>), grid, block, 0, s, a);       \
  else hipLaunchKernelGGL((k_wattn_bwd<NTL, false>), grid, block, 0, s, a);
  switch (cdiv(n, 16)) {
    case 1: WB_LAUNCH(1); break;
    case 2: WB_LAUNCH(2); break;
    case 3: WB_LAUNCH(3); break;
    default: WB_LAUNCH(4); break;
  }
#undef WB_LAUNCH
  SWG_LAUNCH_CHECK();
  return 0;
}
