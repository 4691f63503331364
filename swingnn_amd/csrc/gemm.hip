// bf16 MFMA GEMM for the token-wise linear layers:  out[M,N] = A[M,Kp] @ W[N,Kp]^T  (nn.Linear layout).
//
// gfx950 design notes
//  * v_mfma_f32_16x16x32_bf16 with SWAPPED operands: the MFMA "A" operand is a W tile (rows = output
//    features n) and the "B" operand is an activation tile (rows = tokens m).  The accumulator then has the
//    token on the lane (m = lane & 15) and four CONSECUTIVE output features in its four registers
//    (n = 4*(lane>>4) + r), so the epilogue loads bias / residual and stores results as 16-byte (fp32) or
//    8-byte (bf16) vectors with no cross-lane movement.
//  * 128 x 128 block tile, 4 waves as 2(m) x 2(n), each wave 64 x 64 = 4 x 4 MFMA tiles (64 accumulator VGPRs).
//  * Operands go global -> registers -> LDS (ds_write_b128) with an XOR swizzle on the 16-byte chunk index so
//    that the ds_read_b128 fragment reads are bank-conflict free; LDS is double buffered and the global loads
//    of tile k+1 are in flight while tile k is multiplied (one barrier per K tile).
//  * blockIdx is remapped so that consecutive logical tiles (which share the activation row panel) land on
//    the same XCD and hit its L2.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int BM = 128, BN = 128;

template <int BK>
struct Swz {
  static constexpr int CH = BK / 8;  // 16-byte chunks per LDS row
  __device__ static __forceinline__ int chunk(int row, int ch) {
    if (BK == 64) return ch ^ (row & 7);
    return ch ^ ((-(row >> 2)) & 3);  // BK == 32
  }
};

// bijective XCD remap (8 XCDs, round-robin dispatch): logical tile id from hardware block id
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, slot = bid >> 3;
  const int start = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + slot;
}

// BMT: token rows per block tile.  128 (4 x 4 MFMA tiles per wave) is the default; 64 (2 x 4 per wave) doubles the
// workgroup count for shapes that would otherwise put one or two workgroups on a CU: with so few waves the
// register-staged pipeline exposes a memory latency per k-step (C = 768 fc2 at M = 8064: 378 workgroups, 70 us for
// 14 us of MFMA work).
template <int BK, int BMT = 128>
__global__ __launch_bounds__(256) void k_gemm_nt(const bf16* __restrict__ A, int lda, const bf16* __restrict__ W, int ldw,
                                                 int M, int N, int Kp, GemmEpi e, int tiles_n, int kslice,
                                                 size_t out_slice) {
  constexpr int CH = Swz<BK>::CH;
  constexpr int LDX = (BMT * CH) / 256;  // 16-byte chunks per thread of the activation tile
  constexpr int LDW = (BN * CH) / 256;   // ... of the weight tile
  constexpr int MT = BMT / 32;           // 16-token MFMA tiles per wave
  constexpr int KSTEPS = BK / 32;
  __shared__ __attribute__((aligned(16))) bf16 sX[2][BMT * BK];
  __shared__ __attribute__((aligned(16))) bf16 sW[2][BN * BK];

  const int nwg = gridDim.x;
  if (kslice > 0) {  // split-K (weight gradients): slice blockIdx.y of the reduction, own partial output
    const int z = blockIdx.y;
    A += (size_t)z * kslice;
    W += (size_t)z * kslice;
    Kp = min(kslice, Kp - z * kslice);
    e.out_f32 += (size_t)z * out_slice;
  }
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;  // n fastest: neighbours share the X panel
  const int m0 = tm * BMT, n0 = tn * BN;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;

  // global -> register staging (plain unrolled code: arrays captured by lambdas end up in scratch memory)
  bf16x8 rx[LDX], rw[LDW];
  const bf16* gx[LDX];
  const bf16* gw[LDW];
  int soff[LDW];  // the same (row, chunk) -> LDS offset map serves both tiles (LDX <= LDW)
#pragma unroll
  for (int i = 0; i < LDW; ++i) {
    const int id = tid + i * 256;
    const int row = id / CH, ch = id - row * CH;
    if (i < LDX) gx[i < LDX ? i : 0] = A + (size_t)min(m0 + row, M - 1) * lda + ch * 8;
    gw[i] = W + (size_t)min(n0 + row, N - 1) * ldw + ch * 8;
    soff[i] = row * BK + Swz<BK>::chunk(row, ch) * 8;
  }
#define G_LOAD(KT)                                                                            \
  _Pragma("unroll") for (int i = 0; i < LDW; ++i) {                                           \
    if (i < LDX) rx[i < LDX ? i : 0] = *reinterpret_cast<const bf16x8*>(gx[i < LDX ? i : 0] + (KT) * BK); \
    rw[i] = *reinterpret_cast<const bf16x8*>(gw[i] + (KT) * BK);                               \
  }
#define S_STORE(BUF)                                                                          \
  _Pragma("unroll") for (int i = 0; i < LDW; ++i) {                                           \
    if (i < LDX) *reinterpret_cast<bf16x8*>(&sX[BUF][soff[i]]) = rx[i < LDX ? i : 0];          \
    *reinterpret_cast<bf16x8*>(&sW[BUF][soff[i]]) = rw[i];                                     \
  }

  f32x4 acc[4][MT];  // [nt][mt]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment read offsets (elements) of this lane, per 32-deep k-step
  int fwo[4][KSTEPS], fxo[MT][KSTEPS];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      const int rw_ = wn * 64 + t * 16 + l15;
      fwo[t][ks] = rw_ * BK + Swz<BK>::chunk(rw_, ks * 4 + lg) * 8;
      if (t < MT) {
        const int rx_ = wm * (16 * MT) + t * 16 + l15;
        fxo[t < MT ? t : 0][ks] = rx_ * BK + Swz<BK>::chunk(rx_, ks * 4 + lg) * 8;
      }
    }

  const int nk = Kp / BK;
  G_LOAD(0);
  S_STORE(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) { G_LOAD(kt + 1); }
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) {
      bf16x8 fw[4], fx[MT];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        fw[t] = *reinterpret_cast<const bf16x8*>(&sW[buf][fwo[t][ks]]);
        if (t < MT) fx[t < MT ? t : 0] = *reinterpret_cast<const bf16x8*>(&sX[buf][fxo[t < MT ? t : 0][ks]]);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
          acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nt], fx[mt], acc[nt][mt], 0, 0, 0);
    }
    if (kt + 1 < nk) { S_STORE(buf ^ 1); }
    __syncthreads();
  }
#undef G_LOAD
#undef S_STORE

  // epilogue: lane owns token m = ... + l15 and features n = ... + 4*lg + {0..3}
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + wm * (16 * MT) + mt * 16 + l15;
    if (m >= M) continue;
    const float* crow = e.cond ? e.cond + (size_t)(m / e.cond_L) * e.cond_stride + e.cond_off : nullptr;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + 4 * lg;
      if (n >= e.n_store) continue;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N) {
        v = make_float4(acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]);
        if (e.bias) {
          const float4 bb = *reinterpret_cast<const float4*>(e.bias + n);
          v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        if (e.act == 1) { v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w); }
        if (e.residual) {
          float4 rr = *reinterpret_cast<const float4*>(e.residual + (size_t)m * e.ldr + n);
          if (crow) {
            const float4 sc = *reinterpret_cast<const float4*>(crow + n);
            const float4 sh = *reinterpret_cast<const float4*>(crow + N + n);
            rr.x = silu_f(sh.x + rr.x * (sc.x + 1.f)); rr.y = silu_f(sh.y + rr.y * (sc.y + 1.f));
            rr.z = silu_f(sh.z + rr.z * (sc.z + 1.f)); rr.w = silu_f(sh.w + rr.w * (sc.w + 1.f));
          }
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
      }
      if (e.out_f32) {
        if (n < N) *reinterpret_cast<float4*>(e.out_f32 + (size_t)m * e.ldo + n) = v;
      } else {
        *reinterpret_cast<uint2*>(e.out_bf16 + (size_t)m * e.ldo + n) = pack_bf16x4(v.x, v.y, v.z, v.w);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Large-shape form: 256 (tokens) x 128 (features) block tile, 4 waves as 2(m) x 2(n), each wave 128 x 64 = 8 x 4 MFMA
// tiles (128 accumulator VGPRs, two workgroups per CU).  Against the 128 x 128 form this halves the LDS bytes read
// per MFMA (12 fragment reads per 32 MFMAs instead of 8 per 16) and drops the register staging: operand tiles go
// global -> LDS by LDS-DMA (16 B per lane, 1 KiB per wave-instruction = 16 rows x 64 B), three stages deep with a
// counted vmcnt, one barrier per 32-deep k-step.  The DMA writes lane i to LDS chunk i, so the XOR swizzle of the
// fragment reads is applied on the GLOBAL side: lane (row, physical chunk) fetches logical chunk pc ^ swz(row).
// ------------------------------------------------------------------------------------------------------------------
constexpr int G2_BN = 128, G2_BK = 32;

template <int N>
__device__ __forceinline__ void g2_wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MT = MFMA token tiles per wave (8: 256-token block, 4: 128-token block); NB = LDS stages (NB - 1 tiles in flight)
template <int MT, int NB>
__global__ __launch_bounds__(256, 2) void k_gemm_nt_big(const bf16* __restrict__ A, int lda, const bf16* __restrict__ W,
                                                        int ldw, int M, int N, int Kp, GemmEpi e, int tiles_n) {
  constexpr int G2_BM = 32 * MT;
  constexpr int G2_STAGE = (G2_BM + G2_BN) * G2_BK;  // elements per stage: X tile then W tile
  constexpr int XI = MT / 2;                          // X wave-instructions per wave and stage (W: 2)
  constexpr int PER = XI + 2;                         // DMA instructions per wave and stage
  __shared__ __attribute__((aligned(1024))) bf16 sm[NB * G2_STAGE];
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * G2_BM, n0 = tn * G2_BN;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;

  // LDS-DMA sources: wave-instruction j of a stage covers tile rows 16 j .. 16 j + 15 (X: j < BM / 16, W: j < 8)
  const int lr = lane >> 2, pc = lane & 3;
  const bf16* gsrc[PER];
  unsigned ldst[PER];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int j = wave + 4 * i, row = 16 * j + lr;
    gsrc[i] = A + (size_t)min(m0 + row, M - 1) * lda + Swz<32>::chunk(row, pc) * 8;
    ldst[i] = j * 1024;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = wave + 4 * i, row = 16 * j + lr;
    gsrc[XI + i] = W + (size_t)min(n0 + row, N - 1) * ldw + Swz<32>::chunk(row, pc) * 8;
    ldst[XI + i] = G2_BM * G2_BK * 2 + j * 1024;
  }
  const unsigned sm_base = lds_addr(sm);
#define G2_ISSUE(KT, BUF)                                                                          \
  _Pragma("unroll") for (int i = 0; i < PER; ++i)                                                   \
      lds_dma16(gsrc[i] + (size_t)(KT) * G2_BK, sm_base + (BUF) * (G2_STAGE * 2) + ldst[i]);

  f32x4 acc[4][MT];  // [nt][mt]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  int fxo[MT], fwo[4];
#pragma unroll
  for (int t = 0; t < MT; ++t) {
    const int r = wm * (16 * MT) + t * 16 + l15;
    fxo[t] = r * G2_BK + Swz<32>::chunk(r, lg) * 8;
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int r = wn * 64 + t * 16 + l15;
    fwo[t] = G2_BM * G2_BK + r * G2_BK + Swz<32>::chunk(r, lg) * 8;
  }

  const int nk = Kp / G2_BK;
#pragma unroll
  for (int st = 0; st < NB - 1; ++st)
    if (st < nk) { G2_ISSUE(st, st); }
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    // stage kt must have landed; the up to NB - 2 younger stages stay in flight
    const int younger = min(nk - 1 - kt, NB - 2);
    if (younger >= 2) g2_wait_vm<2 * PER>();
    else if (younger == 1) g2_wait_vm<PER>();
    else g2_wait_vm<0>();
    __syncthreads();
    if (kt + NB - 1 < nk) {
      const int nb = buf == 0 ? NB - 1 : buf - 1;  // (kt + NB - 1) % NB
      G2_ISSUE(kt + NB - 1, nb);
    }
    const bf16* st = sm + buf * G2_STAGE;
    bf16x8 fw[4], fx[MT];
#pragma unroll
    for (int t = 0; t < 4; ++t) fw[t] = *reinterpret_cast<const bf16x8*>(st + fwo[t]);
#pragma unroll
    for (int t = 0; t < MT; ++t) fx[t] = *reinterpret_cast<const bf16x8*>(st + fxo[t]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[nt], fx[mt], acc[nt][mt], 0, 0, 0);
    buf = buf == NB - 1 ? 0 : buf + 1;
  }
#undef G2_ISSUE

#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = m0 + wm * (16 * MT) + mt * 16 + l15;
    if (m >= M) continue;
    const float* crow = e.cond ? e.cond + (size_t)(m / e.cond_L) * e.cond_stride + e.cond_off : nullptr;
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const int n = n0 + wn * 64 + nt * 16 + 4 * lg;
      if (n >= e.n_store) continue;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N) {
        v = make_float4(acc[nt][mt][0], acc[nt][mt][1], acc[nt][mt][2], acc[nt][mt][3]);
        if (e.bias) {
          const float4 bb = *reinterpret_cast<const float4*>(e.bias + n);
          v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        if (e.act == 1) { v.x = gelu_f(v.x); v.y = gelu_f(v.y); v.z = gelu_f(v.z); v.w = gelu_f(v.w); }
        if (e.residual) {
          float4 rr = *reinterpret_cast<const float4*>(e.residual + (size_t)m * e.ldr + n);
          if (crow) {
            const float4 sc = *reinterpret_cast<const float4*>(crow + n);
            const float4 sh = *reinterpret_cast<const float4*>(crow + N + n);
            rr.x = silu_f(sh.x + rr.x * (sc.x + 1.f)); rr.y = silu_f(sh.y + rr.y * (sc.y + 1.f));
            rr.z = silu_f(sh.z + rr.z * (sc.z + 1.f)); rr.w = silu_f(sh.w + rr.w * (sc.w + 1.f));
          }
          v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
        }
      }
      if (e.out_f32) {
        if (n < N) *reinterpret_cast<float4*>(e.out_f32 + (size_t)m * e.ldo + n) = v;
      } else {
        *reinterpret_cast<uint2*>(e.out_bf16 + (size_t)m * e.ldo + n) = pack_bf16x4(v.x, v.y, v.z, v.w);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Weight-gradient GEMM on TOKEN-MAJOR operands:  out[Mo, No] += sum_t P[t, m] Q[t, n]   (dW = dY^T X)
// P = dY (bf16 [rows, ldp]) and Q = X (bf16 [rows, ldq]) are read exactly as the forward / backward kernels wrote them:
// no transposed copies.  A 32-token x 128-feature tile of each operand goes to LDS row-major; the MFMA fragments,
// which need 8 consecutive TOKENS per lane, are gathered by ds_read_b64_tr_b16 (gfx950 transposing LDS read: 4 tokens
// x 16 features per 16-lane group, delivered feature-per-lane).  The contraction slots of lane group g are tokens
// {4 g .. 4 g + 3} and {16 + 4 g ..} (any order works as long as both operands use it), so the 32 lanes of one LDS pass
// read 8 CONSECUTIVE rows.
// The operand stream is the bound (SQ_WAIT_ANY 54 % of the wave cycles in a register-staged form with one tile in
// flight per workgroup), so tiles go global -> LDS by LDS-DMA, TD_NBUF - 1 of them in flight behind a counted vmcnt.
// One wave-instruction moves 4 rows x 256 B; LDS rows are 256 B with no padding, the bank spread comes from an XOR on
// the 32-byte chunk-pair index applied on the GLOBAL side (lane (row, physical chunk) fetches logical chunk
// pc ^ 2 (row & 7)): the 8 consecutive rows of one transposed-read pass land on 8 distinct chunk pairs.  Rows past the
// slice end are fetched from a clamped (valid) row and the P fragment is zeroed instead.
// 1-D grid over (token slice, tile): workgroup L runs on XCD L % 8, so slice = xcd + 8 (slot / tiles) keeps all tiles
// of one token slice on ONE XCD, back to back (operands fetched from HBM once, the other tiles hit that L2).  Slices
// write fp32 partials that k_splitk_reduce folds into `out`.
// ------------------------------------------------------------------------------------------------------------------
constexpr int TN_BT = 32;    // tokens per k-step
constexpr int TD_NBUF = 4;
constexpr int TD_OP = TN_BT * 128;          // elements per operand tile
constexpr int TD_STAGE = 2 * TD_OP;         // P tile then Q tile

__global__ __launch_bounds__(256, 2) void k_gemm_tn(const bf16* __restrict__ P, int ldp, const bf16* __restrict__ Q,
                                                        int ldq, int rows, int Mo, int No, float* __restrict__ out,
                                                        int tiles_n, int tiles, int nslices, int kslice, size_t out_slice,
                                                        int accumulate, float* __restrict__ colsum_p) {
  __shared__ __attribute__((aligned(1024))) bf16 sm[TD_NBUF * TD_STAGE];
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int sl = xcd + 8 * (slot / tiles);
  if (sl >= nslices) return;
  const int tile = slot % tiles;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int m0 = tm * 128, n0 = tn * 128;
  const int t_begin = sl * kslice, t_end = min(rows, t_begin + kslice);
  float* dst = out + (size_t)sl * out_slice;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l15 = lane & 15, lg = lane >> 4;
  const int q4 = l15 >> 2, p4 = l15 & 3;

  // DMA: wave-instruction j (0..7) of an operand tile covers rows 4 j .. 4 j + 3; this wave issues j = wave, wave + 4
  const int dr = lane >> 4, dpc = lane & 15;
  int drow[2], dcolp[2], dcolq[2];
  unsigned ddst[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = wave + 4 * i;
    drow[i] = 4 * j + dr;
    const int lc = dpc ^ (2 * (drow[i] & 7));
    dcolp[i] = min(m0 + lc * 8, ldp - 8);
    dcolq[i] = min(n0 + lc * 8, ldq - 8);
    ddst[i] = j * 1024;
  }
  const unsigned sm_base = lds_addr(sm);
#define TD_ISSUE(KT, BUF)                                                                                    \
  _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                             \
    const size_t t = (size_t)min(t_begin + (KT) * TN_BT + drow[i], rows - 1);                                 \
    lds_dma16(P + t * ldp + dcolp[i], sm_base + (BUF) * (TD_STAGE * 2) + ddst[i]);                            \
    lds_dma16(Q + t * ldq + dcolq[i], sm_base + (BUF) * (TD_STAGE * 2) + TD_OP * 2 + ddst[i]);                \
  }

  f32x4 acc[4][4];  // [mt][nt]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // optional by-product: colsum_p[m] += sum_t P[t, m] (the bias gradient of the same layer) as one more MFMA per row
  // tile against an all-ones operand, by the tn == 0 workgroups' wn == 0 waves (every column of the result is the sum)
  const bool do_colsum = colsum_p != nullptr && tn == 0 && wn == 0;
  f32x4 accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (bf16)1.0f;

  // transposed-read element offsets inside an operand tile, per 16-column block t and half (rows 4 lg + q, 16 + ...)
  int offa[4], offb[4];
  {
    const int r = 4 * lg + q4;  // (r & 7) is the same for r + 16
    const int sw = 2 * (r & 7);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int lca = 8 * wm + 2 * t + (p4 >> 1), lcb = 8 * wn + 2 * t + (p4 >> 1);
      offa[t] = r * 128 + (lca ^ sw) * 8 + (p4 & 1) * 4;
      offb[t] = r * 128 + (lcb ^ sw) * 8 + (p4 & 1) * 4;
    }
  }
  typedef __attribute__((address_space(3))) bf16x4 lds_b4;

  const int nk = (t_end - t_begin + TN_BT - 1) / TN_BT;
#pragma unroll
  for (int st = 0; st < TD_NBUF - 1; ++st)
    if (st < nk) { TD_ISSUE(st, st); }
  int buf = 0;
  for (int kt = 0; kt < nk; ++kt) {
    if (kt + 2 < nk) g2_wait_vm<8>();
    else if (kt + 1 < nk) g2_wait_vm<4>();
    else g2_wait_vm<0>();
    __syncthreads();
    if (kt + TD_NBUF - 1 < nk) {
      const int nb = buf == 0 ? TD_NBUF - 1 : buf - 1;  // (kt + TD_NBUF - 1) % TD_NBUF
      TD_ISSUE(kt + TD_NBUF - 1, nb);
    }
    const bf16* sp = sm + buf * TD_STAGE;
    const bf16* sq = sp + TD_OP;
    bf16x8 fa[4], fb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(sp + offa[t]));
      bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(sp + offa[t] + 16 * 128));
      const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(sq + offb[t]));
      const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_b4*)(sq + offb[t] + 16 * 128));
      fa[t] = (bf16x8){a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
      fb[t] = (bf16x8){b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
    }
    const int left = t_end - (t_begin + kt * TN_BT);  // valid tokens in this tile
    if (left < TN_BT) {                               // slice tail: slot e of group lg is token 4 lg + e (+ 16)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int tokn = 4 * lg + (e & 3) + (e >> 2) * 16;
          if (tokn >= left) fa[t][e] = (bf16)0.f;
        }
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], fb[nt], acc[mt][nt], 0, 0, 0);
    if (do_colsum) {
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) accb[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[mt], ones, accb[mt], 0, 0, 0);
    }
    buf = buf == TD_NBUF - 1 ? 0 : buf + 1;
  }
#undef TD_ISSUE
  if (do_colsum && l15 == 0) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = m0 + wm * 64 + mt * 16 + 4 * lg + r;
        if (m < Mo) unsafeAtomicAdd(colsum_p + m, accb[mt][r]);
      }
  }
#pragma unroll
  for (int mt = 0; mt < 4; ++mt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * 64 + mt * 16 + 4 * lg + r;
      if (m >= Mo) continue;
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        const int n = n0 + wn * 64 + nt * 16 + l15;
        if (n >= No) continue;
        float* o = dst + (size_t)m * No + n;
        if (accumulate == 2) unsafeAtomicAdd(o, acc[mt][nt][r]);  // token slices add into `out` directly (no-return fp32 atomic)
        else *o = accumulate ? *o + acc[mt][nt][r] : acc[mt][nt][r];
      }
    }
}

// out[i] += sum_s partial[s][i]
__global__ __launch_bounds__(256) void k_splitk_reduce(const float* __restrict__ partial, int S, size_t n4, size_t slice,
                                                       float* __restrict__ out) {
  for (size_t i = blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    float4 a = *reinterpret_cast<const float4*>(out + 4 * i);
    for (int z = 0; z < S; ++z) {
      const float4 p = *reinterpret_cast<const float4*>(partial + (size_t)z * slice + 4 * i);
      a.x += p.x; a.y += p.y; a.z += p.z; a.w += p.w;
    }
    *reinterpret_cast<float4*>(out + 4 * i) = a;
  }
}

}  // namespace

// Weight-gradient form: out[M,N] += A[M,Kp] @ W[N,Kp]^T with a LONG reduction (Kp = tokens) and a small output, so the
// reduction is split over blockIdx.y into fp32 partials that a second kernel folds into `out` (deterministic order).
// A row stride lda, W row stride ldw (both >= Kp, multiples of 8); out dense [M, N].
size_t gemm_splitk_ws_floats(int M, int N) {
  const int tiles = cdiv(M, BM) * cdiv(N, BN);
  return (size_t)(512 + tiles) * BM * BN;
}
int launch_gemm_splitk(const bf16* A, int lda, const bf16* W, int ldw, int M, int N, int Kp, float* out, float* partial,
                       size_t partial_floats, hipStream_t s) {
  if (M <= 0 || N <= 0 || Kp <= 0) return 0;
  if (Kp % 32 || lda % 8 || ldw % 8 || lda < Kp || ldw < Kp || N % 4) return SWG_E_SHAPE;
  if (!A || !W || !out) return SWG_E_NULL;
  const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
  const int tiles = tiles_m * tiles_n;
  int S = min(min(cdiv(512, tiles), max(Kp / 256, 1)), 64);
  int kslice = round_up(cdiv(Kp, S), 32);
  S = cdiv(Kp, kslice);
  GemmEpi e = {};
  e.ldo = N; e.n_store = N;
  SwgProfScope ps(KID_GEMM, 2.0 * M * (double)N * Kp, 2.0 * M * Kp + 2.0 * N * (double)Kp + 8.0 * M * (double)N, s);
  if (S <= 1) {
    e.out_f32 = out; e.residual = out; e.ldr = N;
    hipLaunchKernelGGL(k_gemm_nt<32>, dim3(tiles), dim3(256), 0, s, A, lda, W, ldw, M, N, Kp, e, tiles_n, 0, (size_t)0);
    SWG_LAUNCH_CHECK();
    return 0;
  }
  const size_t slice = (size_t)M * N;
  if (!partial || partial_floats < slice * S) return SWG_E_WORKSPACE;
  e.out_f32 = partial;
  hipLaunchKernelGGL(k_gemm_nt<32>, dim3(tiles, S), dim3(256), 0, s, A, lda, W, ldw, M, N, Kp, e, tiles_n, kslice, slice);
  SWG_LAUNCH_CHECK();
  const size_t n4 = slice / 4;
  hipLaunchKernelGGL(k_splitk_reduce, dim3((unsigned)min((size_t)1024, (n4 + 255) / 256)), dim3(256), 0, s, partial, S, n4,
                     slice, out);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_gemm_tn_splitk(const bf16* P, int ldp, const bf16* Q, int ldq, int rows, int Mo, int No, float* out,
                          float* partial, size_t partial_floats, hipStream_t s, float* colsum_p) {
  if (rows <= 0 || Mo <= 0 || No <= 0) return 0;
  if (ldp % 8 || ldq % 8 || No % 4 || ldp < 8 || ldq < 8) return SWG_E_SHAPE;
  if (!P || !Q || !out) return SWG_E_NULL;
  const int tiles_m = cdiv(Mo, 128), tiles_n = cdiv(No, 128);
  const int tiles = tiles_m * tiles_n;
  // Token slices: slice sl runs on XCD sl % 8 (see k_gemm_tn), 64 workgroup slots per XCD (32 CUs x 2).  With s8
  // slices per XCD an XCD works through tiles * s8 workgroups in ceil(tiles * s8 / 64) rounds of rows / (8 s8 * 32)
  // k-steps each, plus a fixed cost per workgroup (prologue + 64 KB of atomics) of about TN_FIXED k-steps: take the
  // s8 with the least modelled time.  (The former rule, 512 / tiles slices, left XCDs idle whenever that was not a
  // multiple of 8 and spilled a few workgroups into a second round: C = 768 fc1 159 us, C = 192 fc1 157 us.)
  static const int forced_s8 = [] { const char* v = getenv("SWG_TN_S8"); return v ? atoi(v) : 0; }();
  static const int fixed_k = [] { const char* v = getenv("SWG_TN_FIXED"); return v ? atoi(v) : 32; }();
  static const int min_ks = [] { const char* v = getenv("SWG_TN_MINK"); return v ? atoi(v) : 256; }();
  int best_s8 = 1;
  long best_cost = -1;
  for (int s8 = 1; s8 <= 32; ++s8) {
    const int ks = round_up(cdiv(rows, 8 * s8), TN_BT);
    if (ks < min_ks && s8 > 1) break;  // short slices: the atomics of many workgroups on one tile cost more than they save
    const long cost = (long)cdiv(tiles * s8, 64) * (ks / TN_BT + fixed_k);
    if (best_cost < 0 || cost < best_cost) { best_cost = cost; best_s8 = s8; }
  }
  if (forced_s8 > 0) best_s8 = forced_s8;
  int kslice = round_up(cdiv(rows, 8 * best_s8), TN_BT);
  const int S = cdiv(rows, kslice);
  SwgProfScope ps(KID_GEMM_TN, 2.0 * Mo * (double)No * rows, 2.0 * rows * (double)(min(ldp, tiles_m * 128) + min(ldq, tiles_n * 128)) +
                                                           8.0 * Mo * (double)No, s);
  if (S <= 1) {
    hipLaunchKernelGGL(k_gemm_tn, dim3(8 * tiles), dim3(256), 0, s, P, ldp, Q, ldq, rows, Mo, No, out, tiles_n, tiles, 1, kslice,
                       (size_t)0, 1, colsum_p);
    SWG_LAUNCH_CHECK();
    return 0;
  }
  // the S token slices add their tiles into `out` with fp32 atomics: no partial buffers, no reduction pass (the other
  // parameter-gradient kernels accumulate with atomics too; the summation order across slices is not fixed)
  hipLaunchKernelGGL(k_gemm_tn, dim3(tiles * round_up(S, 8)), dim3(256), 0, s, P, ldp, Q, ldq, rows, Mo, No, out, tiles_n,
                     tiles, S, kslice, (size_t)0, 2, colsum_p);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_gemm(const bf16* A, int lda, const bf16* W, int M, int N, int Kp, const GemmEpi& e, hipStream_t s) {
  if (M <= 0) return 0;
  if (Kp % 32 || lda % 8 || lda < Kp || N % 4 || e.ldo % 4 || (e.residual && e.ldr % 4)) return SWG_E_SHAPE;
  if (!e.out_f32 && !e.out_bf16) return SWG_E_NULL;
  GemmEpi ee = e;
  if (ee.out_f32) ee.n_store = N;
  if (ee.n_store < N || ee.n_store > ee.ldo || ee.n_store % 4) return SWG_E_SHAPE;
  const int tiles_m = cdiv(M, BM), tiles_n = cdiv(ee.n_store, BN);
  const int nwg = tiles_m * tiles_n;
  const double osz = ee.out_f32 ? 4.0 : 2.0;

  // BK = 32 halves the LDS footprint (4 workgroups per CU instead of 2): with many row panels in flight the extra
  // residency hides the short-K prologue / epilogue better than the longer k-step amortises its barrier (measured
  // per shape, scripts/bench_gemm.py: +5..25 % for M >= 32k, -3..10 % for M = 8k)
  // form: 0 = by shape (below), 1 = register-staged 128 x 128, 3 = LDS-DMA 256 x 128 (SWG_GEMM_FORM, read once: a
  // tuning knob for scripts/bench_gemm.py, not an API).  An LDS-DMA 128 x 128 form (4 stages) was measured too: equal
  // or up to 10 % slower than the register-staged one on every shape of the network, so it is not built.
  static const int forced = [] { const char* v = getenv("SWG_GEMM_FORM"); return v ? atoi(v) : 0; }();
  const int big_wg = cdiv(M, 256) * tiles_n;
  int form = forced;
  if (form == 0) {
    // measured per shape (scripts/bench_gemm.py, B = 112): the large form wins for wide outputs and for very long token
    // panels; short-K fp32 + residual shapes (epilogue-bound) keep the small form's higher workgroup residency
    form = (big_wg >= 384 && (ee.n_store >= 1024 || M >= 131072 || (ee.n_store >= 768 && M >= 65536))) ? 3 : 1;
  }
  const int kid = form == 3 ? KID_GEMM_NT_BIG : ((Kp % 64 == 0 && M < 16384) ? KID_GEMM_NT64 : KID_GEMM);
  SwgProfScope ps(kid, 2.0 * M * (double)N * Kp,
                  2.0 * M * Kp + 2.0 * N * (double)Kp + osz * M * (double)N + (ee.residual ? 4.0 * M * (double)N : 0.0), s);
  if (form == 3) {
    hipLaunchKernelGGL((k_gemm_nt_big<8, 3>), dim3(big_wg), dim3(256), 0, s, A, lda, W, Kp, M, N, Kp, ee, tiles_n);
    SWG_LAUNCH_CHECK();
    return 0;
  }
  // few workgroups (at most two per CU): 64-row tiles double their number.  Measured at B = 56 (scripts/bench_gemm.py,
  // SWG_GEMM_BM64 = 0 / 1): C = 768 proj 30.9 -> 24.5 us, fc2 70.9 -> 62.0 us, the 384 -> 768 merge 38.1 -> 33.5 us;
  // at 756 workgroups (M = 32256, N = 384) the 128-row tile is still the better one
  static const int small_m = [] { const char* v = getenv("SWG_GEMM_BM64"); return v ? atoi(v) : 1; }();
  if (small_m && nwg <= 512 && M >= 128) {
    const int nwg64 = cdiv(M, 64) * tiles_n;
    if (Kp % 64 == 0)
      hipLaunchKernelGGL((k_gemm_nt<64, 64>), dim3(nwg64), dim3(256), 0, s, A, lda, W, Kp, M, N, Kp, ee, tiles_n, 0, (size_t)0);
    else
      hipLaunchKernelGGL((k_gemm_nt<32, 64>), dim3(nwg64), dim3(256), 0, s, A, lda, W, Kp, M, N, Kp, ee, tiles_n, 0, (size_t)0);
    SWG_LAUNCH_CHECK();
    return 0;
  }
  if (Kp % 64 == 0 && M < 16384)
    hipLaunchKernelGGL(k_gemm_nt<64>, dim3(nwg), dim3(256), 0, s, A, lda, W, Kp, M, N, Kp, ee, tiles_n, 0, (size_t)0);
  else
    hipLaunchKernelGGL(k_gemm_nt<32>, dim3(nwg), dim3(256), 0, s, A, lda, W, Kp, M, N, Kp, ee, tiles_n, 0, (size_t)0);
  SWG_LAUNCH_CHECK();
  return 0;
}
