// Fused PatchMerging for the narrow stages (C = 96, 192): model/swin_gnn/swin_gnn.py:298-335,
//     out[b, Y, X, :] = Linear_{4C -> 2C}( LayerNorm_{4C}( cat( x[b, 2Y+dy, 2X+dx, :] : (dy,dx) = (0,0),(1,0),(0,1),(1,1) ) ) )
// in ONE pass over the fp32 token rows: the separate gather + LayerNorm pass wrote a bf16 operand that the GEMM read
// back (6 B per input element); here the normalised rows go straight into MFMA operand registers.
// Same skeleton as the fc1 half of mlp_stream.hip: 32 merged tokens per wave (two 16-token MFMA tiles) stay resident
// as bf16 fragments, the weight [2C, 4C] streams through a 3-deep LDS ring in 24-KiB chunks of NC output features
// (LDS-DMA, counted s_waitcnt, one barrier per chunk, fragment order so that every ds_read_b128 is conflict-free), the
// body is issued as pinned groups {2 MFMAs on one weight fragment, the ds_read of the fragment AHEAD groups later, every
// FPC/PIECES groups one DMA piece}.  The 32 x 2C outputs stay in accumulators until the stream is done.  One wave per
// SIMD (the 4C-wide rows alone are 192 registers at C = 192).
// Measured (B = 56; scripts/bench_merge.py, scripts/diag_merge_stamps.py): C = 96 114 -> 80 us, C = 192 63 -> 44 us
// against the LayerNorm pass + GEMM.  The workgroups run in lockstep, one per CU: while all of them load rows the HBM is
// saturated (the row phase is ~45 % of a workgroup's life, and the first in-loop DMA pieces queue behind it: 2-4 k
// ticks per piece instead of ~50), while they multiply (23 ticks per MFMA) it idles.  Rotating the chunk order per
// workgroup, priming the ring before the rows, touching the image's pages first and storing per chunk changed
// nothing; two workgroups per CU need <= 256 registers (76 spilled at C = 96).
#include "common.h"

namespace {

struct MergeArgs {
  const float* x;   // [B, res*res, C]
  const bf16* img;  // weight image, engine.py merge_image()
  const float* g;   // LayerNorm(4C) weight
  const float* b;   // LayerNorm(4C) bias
  float* out;       // [B * (res/2)^2, 2C]
  int T4, res;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int CS, int NC, int NBUF>
__global__ __launch_bounds__(256, 1) void k_merge_linear(MergeArgs a) {
  constexpr int K = 4 * CS, N = 2 * CS;
  constexpr int KS = K / 32;   // k-steps
  constexpr int KQ = CS / 32;  // k-steps per source token (quarter of the merged row)
  constexpr int HT = NC / 16;  // output tiles per chunk
  constexpr int NCH = N / NC;  // chunks
  constexpr int CHUNK_BYTES = NC * K * 2;
  constexpr int PIECES = CHUNK_BYTES / 4096;  // 1-KiB LDS-DMA pieces per wave per chunk
  constexpr int FPC = CHUNK_BYTES / 1024;     // fragments per chunk
  constexpr int AHEAD = 6;
  static_assert(FPC == KS * HT && CHUNK_BYTES % 4096 == 0 && FPC % PIECES == 0, "chunk shape");
  static_assert(NBUF == 3 && NCH >= 3 && PIECES * (NBUF - 1) < 64, "ring depth / vmcnt is 6 bits");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NBUF x CHUNK_BYTES | g [K] | b [K]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int tok0 = blockIdx.x * 128 + wave * 32;
  const unsigned smem0 = lds_addr(smem);
#ifdef SWG_STAMPS
  const bool dbg = wave == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 3 || blockIdx.x == gridDim.x - 2);
  const int dbgslot = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x / 3 ? 1 : 2);
#endif
  STAMP(0);
  const char* img_lane = reinterpret_cast<const char*>(a.img) + lane * 16;
#define STAGE_PIECE(chunk, buf_off, i)                                                        \
  lds_dma16(img_lane + (size_t)(chunk) * CHUNK_BYTES + ((i) * 4 + wave) * 1024,               \
            smem0 + (buf_off) + ((i) * 4 + wave) * 1024)

  // source rows of this lane's two merged tokens: quarter q of merged token (b, Y, X) is token (2Y + (q&1), 2X + (q>>1))
  const int r2 = a.res >> 1;
  const float* src[2][4];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int m = min(tok0 + mt * 16 + l15, a.T4 - 1);
    const int bi = m / (r2 * r2), rem = m - bi * r2 * r2;
    const int Y = rem / r2, X = rem - Y * r2;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      src[mt][q] = a.x + ((size_t)bi * a.res * a.res + (size_t)(2 * Y + (q & 1)) * a.res + (2 * X + (q >> 1))) * CS + 8 * lg;
  }
  // both tiles' rows are requested at once where they fit (C = 96: 192 registers); at C = 192 one tile at a time
  constexpr bool BOTH = (CS <= 96);
  f32x4 xr[BOTH ? 2 : 1][KS][2];
  auto load_rows = [&](int mt, int slot) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float* p = src[mt][ks / KQ] + (ks % KQ) * 32;
      xr[slot][ks][0] = *reinterpret_cast<const f32x4*>(p);
      xr[slot][ks][1] = *reinterpret_cast<const f32x4*>(p + 4);
    }
  };
  // LayerNorm weight / bias go through LDS (K-indexed, shared by the four waves); their loads are issued first so
  // that the wait in front of the LDS store does not cover the row loads behind them
  constexpr int GB = (K + 255) / 256;
  float gv[GB], bv[GB];
#pragma unroll
  for (int i = 0; i < GB; ++i) {
    const int c = min(tid + 256 * i, K - 1);
    gv[i] = a.g[c];
    bv[i] = a.b[c];
  }
  load_rows(0, 0);
  if (BOTH) load_rows(1, BOTH ? 1 : 0);
  STAMP(1);
  float* sg = reinterpret_cast<float*>(smem + NBUF * CHUNK_BYTES);
#pragma unroll
  for (int i = 0; i < GB; ++i) {
    const int c = tid + 256 * i;
    if (c < K) { sg[c] = gv[i]; sg[K + c] = bv[i]; }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  STAMP(2);
  bf16x8 xa[2][KS];
  auto layer_norm = [&](int mt, int slot) {
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int h = 0; h < 2; ++h) s += (xr[slot][ks][h][0] + xr[slot][ks][h][1]) + (xr[slot][ks][h][2] + xr[slot][ks][h][3]);
    const float mean = rows_sum(s) * (1.0f / K);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const float d = xr[slot][ks][h][e] - mean; q += d * d; }
    const float rstd = ln_rstd(rows_sum(q) * (1.0f / K));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(sg + 32 * ks + 8 * lg);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(sg + 32 * ks + 8 * lg + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(sg + K + 32 * ks + 8 * lg);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(sg + K + 32 * ks + 8 * lg + 4);
      bf16x8 f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f[e] = (bf16)((xr[slot][ks][0][e] - mean) * rstd * g0[e] + b0[e]);
        f[4 + e] = (bf16)((xr[slot][ks][1][e] - mean) * rstd * g1[e] + b1[e]);
      }
      xa[mt][ks] = f;
    }
  };
  if (BOTH) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) STAGE_PIECE(0, 0, i);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) STAGE_PIECE(1, CHUNK_BYTES, i);
    STAMP(3);
    layer_norm(0, 0);
    STAMP(4);
    layer_norm(1, BOTH ? 1 : 0);
  } else {
    layer_norm(0, 0);
    STAMP(3);
    __builtin_amdgcn_sched_barrier(0);  // the second tile's rows reuse the first tile's registers: no hoisting
    load_rows(1, 0);  // requested BEFORE the ring is primed: loads return in order
#pragma unroll
    for (int i = 0; i < PIECES; ++i) STAGE_PIECE(0, 0, i);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) STAGE_PIECE(1, CHUNK_BYTES, i);
    STAMP(4);
    layer_norm(1, 0);
  }

  // All N outputs of the wave's 32 tokens stay in accumulators and are stored after the last chunk: a store inside the
  // chunk loop sits between the LDS-DMA pieces in the in-order vector-memory queue, so the counted wait for chunk j
  // would also wait for the write acknowledgements of chunk j-3 (measured: 62 instead of ~30 cycles per MFMA).
  STAMP(5);
  f32x4 acc[NCH * HT][2];
  __builtin_amdgcn_sched_barrier(0);  // the accumulators take over the registers of the raw rows: no early zeroing
#pragma unroll
  for (int t = 0; t < NCH * HT; ++t) {
    acc[t][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    // chunk j has landed in this wave's view; the pieces of chunk j+1 may stay in flight (the refill below runs in
    // every iteration -- past the end it re-copies the last chunk into a dead slot -- so the count is always the same)
    wait_vmcnt<(NBUF - 2) * PIECES>();
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");  // no LDS read of this iteration may move above the barrier
    STAMP(6 + 2 * j);
    // every wave is past its reads of the slot used one iteration ago: it is refilled piece by piece below
    const unsigned bo = (unsigned)(j % NBUF) * CHUNK_BYTES;
    const unsigned prev = (unsigned)((j + NBUF - 1) % NBUF) * CHUNK_BYTES;
    const int next_chunk = (j + NBUF - 1 < NCH) ? j + NBUF - 1 : NCH - 1;
    const char* cb = smem + bo + lane * 16;
    bf16x8 fr[AHEAD + 1];
#pragma unroll
    for (int p = 0; p < AHEAD; ++p) fr[p] = *reinterpret_cast<const bf16x8*>(cb + p * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < FPC; ++p) {
      const bf16x8 w = fr[p % (AHEAD + 1)];
      const int ks = p / HT, t = j * HT + p % HT;  // image order: k-step major, output tile minor
      acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xa[0][ks], acc[t][0], 0, 0, 0);
      acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xa[1][ks], acc[t][1], 0, 0, 0);
      if (p + AHEAD < FPC) fr[(p + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const bf16x8*>(cb + (p + AHEAD) * 1024);
      if (p % (FPC / PIECES) == 1) STAGE_PIECE(next_chunk, prev, p / (FPC / PIECES));
      __builtin_amdgcn_sched_barrier(0);
    }
    STAMP(7 + 2 * j);
  }
  wait_vmcnt<0>();  // the trailing refills must have landed before the workgroup's LDS is released
  STAMP(60);
  // accumulator row 4*lg + r of tile t = output feature 16*t + 4*lg + r of token l15: 64 contiguous bytes per token and
  // store instruction
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int tok = tok0 + mt * 16 + l15;
    if (tok >= a.T4) continue;
#pragma unroll
    for (int t = 0; t < NCH * HT; ++t)
      *reinterpret_cast<f32x4*>(a.out + (size_t)tok * N + 16 * t + 4 * lg) = acc[t][mt];
  }
  STAMP(61);
#undef STAGE_PIECE
}

template <int CS, int NC>
int launch_one(const MergeArgs& a, hipStream_t s) {
  constexpr size_t lds = (size_t)3 * (NC * 4 * CS * 2) + (size_t)8 * CS * sizeof(float);
  static std::atomic<uint64_t> lds_ok{0};  // dynamic LDS above 64 KiB needs an explicit opt-in, per device
  if (int e = ensure_dyn_lds(lds_ok, (const void*)k_merge_linear<CS, NC, 3>, (int)lds)) return e;
  hipLaunchKernelGGL((k_merge_linear<CS, NC, 3>), dim3(cdiv(a.T4, 128)), dim3(256), lds, s, a);
  return 0;
}

}  // namespace

bool merge_fused_supported(int C, int res) { return (C == 96 || C == 192) && res % 2 == 0; }

int launch_merge_fused(int C, int B, int res, const float* x, const bf16* img, const float* g, const float* b, float* out,
                       hipStream_t s) {
  if (!merge_fused_supported(C, res)) return SWG_E_UNSUPPORTED;
  if (!x || !img || !g || !b || !out) return SWG_E_NULL;
  const int T4 = B * (res / 2) * (res / 2);
  if (T4 <= 0) return 0;
  MergeArgs a = {x, img, g, b, out, T4, res};
  SwgProfScope ps(KID_MERGE_FUSED, 2.0 * T4 * 8.0 * C * C, 4.0 * T4 * 4.0 * C + 4.0 * T4 * 2.0 * C + 16.0 * C * C, s);
  int rc = (C == 96) ? launch_one<96, 32>(a, s) : launch_one<192, 16>(a, s);
  if (rc) return rc;
  SWG_LAUNCH_CHECK();
  return 0;
}
