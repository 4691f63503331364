// Fused MLP half of a Swin block for the WIDE stages (C = 384): model/swin_gnn/swin_gnn.py:275 with Mlp :19-25,
//     x_out = x + fc2( GELU( fc1( LayerNorm2(x) ) ) )
// PROJ form (C = 384, where the attention kernel cannot hold the projection accumulators): the attention half's tail
// is folded in front, :137 / :272 / :238-240,
//     x = silu(shift + x_in (1 + scale)) + proj_b + o Wproj^T      (o: the attention output, bf16, slot order)
// as six more chunks of the same weight stream whose MFMAs accumulate straight into the fc2 accumulators; LayerNorm2
// then reads those accumulators.  One kernel and one pass over the token rows replace the projection GEMM (read o,
// read x_in, write x) and the MLP kernel's own read of x.
// Same register-resident scheme as mlp_fused.hip (tokens on lanes, the fc1 accumulators ARE the fc2 operand), but
// built for ONE wave per SIMD with 512 registers, where nothing but the instruction order hides latency:
//  * hidden chunks of HC features stream through an NBUF-deep LDS ring by LDS-DMA; the ring is waited for with a
//    COUNTED s_waitcnt so NBUF-2 chunks stay in flight across the per-chunk barrier;
//  * the hidden loop is a 3-stage software pipeline: iteration j multiplies fc1 of chunk j and fc2 of chunk j-2
//    while the GELU of chunk j-1 runs on the VALU in the shadow of those MFMAs.  Image chunk j therefore carries the
//    fc1 rows of hidden chunk j next to the fc2 columns of hidden chunk j-2 (zero blocks at both ends keep the body
//    branch-free);
//  * the body is issued as "MFMA pairs": {2 MFMAs sharing one weight fragment, the ds_read of the fragment AHEAD
//    pairs later, one slice of the GELU, every few pairs one LDS-DMA piece}, pinned with sched_barrier because the
//    compiler's own order (read, wait, 2 MFMAs; all VALU in one block) exposes every LDS latency;
//  * the image is in FRAGMENT order (1 KiB per MFMA operand fragment, lane-linear), so every ds_read_b128 is
//    conflict-free with one base register + immediate offsets, and staging is a linear copy.
#include "common.h"

namespace {

struct MlpArgs {
  const float* x;
  float* out;
  const bf16* img;   // weight image, see engine.py mlp_stream_image()
  const float* b1;   // [4C]
  const float* b2;   // [C]
  const float* g;    // LN2 weight
  const float* b;    // LN2 bias
  int T;
  // PROJ form only
  const bf16* o;     // attention output [T, ldo], channel slots as k_wattn_ring stores them
  int ldo;
  const float* pb;   // proj bias [C]
  const float* cond; // conditioning rows [B, cond_stride]; this block's (scale | shift) at cond_off
  int cond_stride, cond_off, L;
};

typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
  const f32x2 f = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f, bf16x2));
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int C, int HC, int NBUF, int LB, bool PROJ>
__global__ __launch_bounds__(256, LB) void k_mlp_stream(MlpArgs a) {
  constexpr int KS = C / 32;        // k-steps of fc1
  constexpr int NT = C / 16;        // output feature tiles of fc2
  constexpr int HT = HC / 16;       // hidden tiles per chunk (fc1 output tiles)
  constexpr int S2 = HC / 32;       // k-steps of fc2 per chunk
  constexpr int NCH = 4 * C / HC;   // hidden chunks; the image has NCH + 2 chunks (pipeline skew)
  constexpr int W1_BYTES = HC * C * 2;
  constexpr int CHUNK_BYTES = 2 * W1_BYTES;
  constexpr int PIECES = CHUNK_BYTES / 4096;  // 1-KiB LDS-DMA pieces per wave per chunk
  constexpr int FPC = CHUNK_BYTES / 1024;     // fragments per chunk
  constexpr int NP = PROJ ? (C * C * 2) / CHUNK_BYTES : 0;  // projection chunks in front of the MLP stream
  static_assert(!PROJ || (C * C * 2) % CHUNK_BYTES == 0, "the projection weight is a whole number of chunks");
  static_assert(CHUNK_BYTES % 4096 == 0 && PIECES * (NBUF - 1) < 64, "vmcnt is 6 bits");
  static_assert(NBUF == 2 || NBUF == 3, "ring depth");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NBUF x CHUNK_BYTES | b1 | 2 HC zeros

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int tok0 = blockIdx.x * 128 + wave * 32;
  const unsigned smem0 = lds_addr(smem);
#ifdef SWG_STAMPS
  const bool dbg = (blockIdx.x == 3 || blockIdx.x == gridDim.x / 2) && (wave == 0 || wave == 2);
  const int dbgslot = (blockIdx.x == 3 ? 0 : 2) + (wave == 0 ? 0 : 1);
#endif
  STAMP(0);

  // (A wave-uniform source base + scalar address math per piece -- common.h lds_dma16_u, wave index through
  // v_readfirstlane -- removes 38 VALU instructions per chunk from this loop, yet the C = 384 PROJ instance ran 12 %
  // SLOWER with the wave index made uniform, whichever DMA form followed (1410 -> 1584 us per pass; C = 192 +1 %): the
  // register allocation of its prologue changes.  Measured round 3, not kept.)
  const char* img_lane = reinterpret_cast<const char*>(a.img) + lane * 16;
  // LDS-DMA piece i (of PIECES per wave) of image chunk `chunk` into the ring slot at byte offset buf_off
  // C = 192: the LDS destination of a piece is computed on the scalar unit (one v_readfirstlane per kernel instead of
  // a v_add + v_readfirstlane per piece: 710 -> 694 us per pass).  The C = 384 PROJ instance runs 10 % SLOWER with
  // exactly this change (1434 -> 1584 us, reproducibly, also with a wave-uniform source base): left as it was.
  constexpr bool DST_SCALAR = (C == 192);
  const unsigned dst_wave = DST_SCALAR ? __builtin_amdgcn_readfirstlane(smem0 + wave * 1024) : 0u;
#define STAGE_PIECE(chunk, buf_off, i)                                                        \
  lds_dma16(img_lane + (size_t)(chunk) * CHUNK_BYTES + ((i) * 4 + wave) * 1024,               \
            DST_SCALAR ? dst_wave + (buf_off) + (i) * 4096 : smem0 + (buf_off) + ((i) * 4 + wave) * 1024)
  // the token rows are requested BEFORE the ring is primed: loads return in order, so rows issued behind two chunks
  // of LDS-DMA would wait for them (C = 384: 192 registers of raw rows do not fit beside the LayerNorm temporaries)
  constexpr bool ROWS_FIRST = C <= 192 && !PROJ;
  f32x4 xr[ROWS_FIRST ? 2 : 1][ROWS_FIRST ? KS : 1][2];
  if (ROWS_FIRST) {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const float* row = a.x + (size_t)min(tok0 + mt * 16 + l15, a.T - 1) * C + 8 * lg;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        xr[ROWS_FIRST ? mt : 0][ROWS_FIRST ? ks : 0][0] = *reinterpret_cast<const f32x4*>(row + 32 * ks);
        xr[ROWS_FIRST ? mt : 0][ROWS_FIRST ? ks : 0][1] = *reinterpret_cast<const f32x4*>(row + 32 * ks + 4);
      }
    }
  }
  float* sb1 = reinterpret_cast<float*>(smem + NBUF * CHUNK_BYTES);  // fc1 bias [4C] + 2 HC zeros (skew tail)
  for (int i = tid; i < 4 * C; i += 256) sb1[i] = a.b1[i];
  for (int i = tid; i < 2 * HC; i += 256) sb1[4 * C + i] = 0.f;
#pragma unroll
  for (int i = 0; i < PIECES; ++i) STAGE_PIECE(0, 0, i);
  if (NBUF == 3) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) STAGE_PIECE(1, CHUNK_BYTES, i);
  }

  // ---- the wave's 32 token rows in operand layout, LayerNorm2 in registers.  The fc2 accumulators START as
  // residual + fc2 bias: accumulator row 4*lg+r of out tile nt = 2*ks+hh is channel 32ks + 8lg + 4hh + r (rho order of
  // the image rows), the channels this lane loads for k-step ks, so the epilogue is a plain store ----
  bf16x8 xa[2][KS];
  f32x4 acc[NT][2];
  unsigned bo = 0;  // byte offset of the ring slot holding the current chunk
  if (PROJ) {
    // ---- attention tail: acc = shortcut + proj bias, then acc += Wproj o^T over NP chunks of the stream ----
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int tok = min(tok0 + mt * 16 + l15, a.T - 1);
      const float* row = a.x + (size_t)tok * C + 8 * lg;
      const float* crow = a.cond + (size_t)(tok / a.L) * a.cond_stride + a.cond_off + 8 * lg;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int c = 32 * ks + 4 * hh;
          const f32x4 xv = *reinterpret_cast<const f32x4*>(row + c);
          const f32x4 sc = *reinterpret_cast<const f32x4*>(crow + c);
          const f32x4 sh = *reinterpret_cast<const f32x4*>(crow + C + c);
          const f32x4 pb = *reinterpret_cast<const f32x4*>(a.pb + 8 * lg + c);
          f32x4 r;
#pragma unroll
          for (int e = 0; e < 4; ++e) r[e] = silu_f(sh[e] + xv[e] * (sc[e] + 1.f)) + pb[e];
          acc[2 * ks + hh][mt] = r;
        }
        // register peak: loads are hoisted at most across four k-steps (one exposed latency per group, not per step)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    STAMP(1);
    bf16x8 oa[2][KS];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const bf16* orow = a.o + (size_t)min(tok0 + mt * 16 + l15, a.T - 1) * a.ldo + 8 * lg;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) oa[mt][ks] = *reinterpret_cast<const bf16x8*>(orow + 32 * ks);
    }
    __syncthreads();  // sb1 visible (same role as the barrier in front of the hidden loop)
#pragma unroll
    for (int jp = 0; jp < NP; ++jp) {
      wait_vmcnt<(NBUF - 2) * PIECES>();
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const unsigned prev = (bo == 0) ? (NBUF - 1) * CHUNK_BYTES : bo - CHUNK_BYTES;
      const char* cb = smem + bo + lane * 16;
      bf16x8 fr[7];
#pragma unroll
      for (int p = 0; p < 6; ++p) fr[p] = *reinterpret_cast<const bf16x8*>(cb + p * 1024);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int p = 0; p < FPC; ++p) {
        const bf16x8 w = fr[p % 7];
        const int f = jp * FPC + p, ks = f / NT, nt = f % NT;  // stream order: k-step major, rho-ordered out tile
        acc[nt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, oa[0][ks], acc[nt][0], 0, 0, 0);
        acc[nt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, oa[1][ks], acc[nt][1], 0, 0, 0);
        if (p + 6 < FPC) fr[(p + 6) % 7] = *reinterpret_cast<const bf16x8*>(cb + (p + 6) * 1024);
        if (p % (FPC / PIECES) == 1) STAGE_PIECE(jp + NBUF - 1, prev, p / (FPC / PIECES));
        __builtin_amdgcn_sched_barrier(0);
      }
      bo = (bo == (NBUF - 1) * CHUNK_BYTES) ? 0u : bo + CHUNK_BYTES;
    }
  }
  STAMP(2);
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int tok = min(tok0 + mt * 16 + l15, a.T - 1);
    const float* row = a.x + (size_t)tok * C;
    float v[KS][8];
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      f32x4 p0, p1;
      if (PROJ) {
        p0 = acc[2 * ks][mt];
        p1 = acc[2 * ks + 1][mt];
      } else if (ROWS_FIRST) {
        p0 = xr[ROWS_FIRST ? mt : 0][ROWS_FIRST ? ks : 0][0];
        p1 = xr[ROWS_FIRST ? mt : 0][ROWS_FIRST ? ks : 0][1];
      } else {
        p0 = *reinterpret_cast<const f32x4*>(row + 32 * ks + 8 * lg);
        p1 = *reinterpret_cast<const f32x4*>(row + 32 * ks + 8 * lg + 4);
      }
      v[ks][0] = p0[0]; v[ks][1] = p0[1]; v[ks][2] = p0[2]; v[ks][3] = p0[3];
      v[ks][4] = p1[0]; v[ks][5] = p1[1]; v[ks][6] = p1[2]; v[ks][7] = p1[3];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[ks][j];
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(a.b2 + 32 * ks + 8 * lg);
      const f32x4 c1 = *reinterpret_cast<const f32x4*>(a.b2 + 32 * ks + 8 * lg + 4);
      acc[2 * ks][mt] = p0 + c0;
      acc[2 * ks + 1][mt] = p1 + c1;
      if (PROJ) __builtin_amdgcn_sched_barrier(0);  // register peak: no hoisting of the later k-steps' loads
    }
    s = rows_sum(s);
    const float mean = s * (1.0f / C);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[ks][j] - mean; q += d * d; }
    q = rows_sum(q);
    const float rstd = ln_rstd(q * (1.0f / C));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float4 g0 = *reinterpret_cast<const float4*>(a.g + 32 * ks + 8 * lg);
      const float4 g1 = *reinterpret_cast<const float4*>(a.g + 32 * ks + 8 * lg + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(a.b + 32 * ks + 8 * lg);
      const float4 b1 = *reinterpret_cast<const float4*>(a.b + 32 * ks + 8 * lg + 4);
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      bf16x8 f;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (bf16)((v[ks][j] - mean) * rstd * gg[j] + bb[j]);
      xa[mt][ks] = f;
      if (PROJ) __builtin_amdgcn_sched_barrier(0);
    }
  }

  constexpr int NV = HT * 2 * 4;           // GELU evaluations per lane per chunk
  constexpr int PAIRS = KS * HT + S2 * NT;  // weight fragments == MFMA pairs per iteration
  constexpr int NVEC = HT * 2;              // ... as f32x4 accumulator vectors
  constexpr int NSL = 11 * NVEC;            // GELU slices (eleven levels per vector)
  constexpr int AHEAD = 6;                  // fragment reads in flight ahead of their MFMAs
  constexpr int DMA_EVERY = PAIRS / PIECES;
  constexpr int MAXS = (NSL + PAIRS - 1) / PAIRS;  // GELU slices per pair (VALU-bound geometries take several)
  static_assert(PAIRS % PIECES == 0 && AHEAD < PAIRS, "schedule shape");
  f32x4 hcur[HT][2];   // fc1 output (+bias) of chunk j-1
  u32x4 hb[2][S2];     // GELU output of chunk j-2 as fc2 operand (packed bf16)
#pragma unroll
  for (int ht = 0; ht < HT; ++ht)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) hcur[ht][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int s2 = 0; s2 < S2; ++s2) hb[mt][s2] = (u32x4){0u, 0u, 0u, 0u};

  if (!PROJ) __syncthreads();  // sb1 visible (the LDS-DMA above is inline asm: no vmcnt(0) is emitted for it here)
  STAMP(3);
#pragma nounroll
  for (int j = 0; j < NCH + 2; ++j) {
#ifdef SWG_STAMPS
    if (j < 50 && (j < 4 || j % 8 == 0)) STAMP(4 + j);
#endif
    // chunk j has landed in every wave's view; NBUF-2 younger chunks may stay in flight
    // (the refill below runs in every iteration -- past the end it re-copies the last chunk into a dead slot --
    // so the count is the same every time)
#ifndef MLP_DIAG_NODMA
    wait_vmcnt<(NBUF - 2) * PIECES>();
#endif
#ifndef MLP_DIAG_NOBARRIER
    __builtin_amdgcn_s_barrier();
#endif
    asm volatile("" ::: "memory");  // no LDS read of this iteration may move above the barrier
    // every wave is now past its reads of the slot used one iteration ago: it is refilled piece by piece below
    const unsigned prev = (bo == 0) ? (NBUF - 1) * CHUNK_BYTES : bo - CHUNK_BYTES;
    const int next_chunk = NP + min(j + NBUF - 1, NCH + 1);
    const char* cb = smem + bo + lane * 16;
    const float* bj = sb1 + j * HC + 4 * lg;

    f32x4 hn[HT][2];       // fc1 accumulators of chunk j, initial value = bias
    u32x4 hbn[2][S2];      // GELU output of chunk j-1 (next iteration's fc2 operand)
    float gs[NV], gq[NV];  // GELU pipeline state
    bf16x8 fr[AHEAD + 1];  // fragment ring
#define FRAG_PTR(P) (cb + ((P) < KS * HT ? (P) * 1024 : W1_BYTES + ((P) - KS * HT) * 1024))
#pragma unroll
    for (int ht = 0; ht < HT; ++ht) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(bj + ht * 16);
      hn[ht][0] = bb;
      hn[ht][1] = bb;
    }
#pragma unroll
    for (int p = 0; p < AHEAD; ++p) fr[p] = *reinterpret_cast<const bf16x8*>(FRAG_PTR(p));
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < PAIRS; ++p) {
      // -- two MFMAs on fragment p: fc1 fragment (k-step, hidden tile) or fc2 fragment (k-step, out tile)
      const bf16x8 w = fr[p % (AHEAD + 1)];
      if (p < KS * HT) {
        const int ks = p / HT, ht = p % HT;
        hn[ht][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xa[0][ks], hn[ht][0], 0, 0, 0);
        hn[ht][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xa[1][ks], hn[ht][1], 0, 0, 0);
      } else {
        const int q = p - KS * HT, s2 = q / NT, nt = q % NT;
        acc[nt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, __builtin_bit_cast(bf16x8, hb[0][s2]), acc[nt][0], 0, 0, 0);
        acc[nt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, __builtin_bit_cast(bf16x8, hb[1][s2]), acc[nt][1], 0, 0, 0);
      }
      // -- fragment read AHEAD pairs ahead
      if (p + AHEAD < PAIRS) fr[(p + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const bf16x8*>(FRAG_PTR(p + AHEAD));
      // -- one LDS-DMA piece of chunk j+NBUF-1 every DMA_EVERY pairs
#ifndef MLP_DIAG_NODMA    // diagnostic only: the chunk loop without refilling the ring
      if (p % DMA_EVERY == 1) STAGE_PIECE(next_chunk, prev, p / DMA_EVERY);
#endif
      // -- GELU of chunk j-1 (gelu_f of common.h, spelled out): vector t = ht*2+mt (4 independent values) advances
      //    one LEVEL per slice, the vectors take turns, so dependent VALU ops are several pairs apart
#pragma unroll
      for (int k = 0; k < MAXS; ++k) {
        const int sl = p * NSL / PAIRS + k;
        if (sl >= (p + 1) * NSL / PAIRS) continue;
        const int t = sl % NVEC, level = sl / NVEC, ht = t >> 1, mt = t & 1;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int v = t * 4 + r;
          const float x = hcur[ht][mt][r];
#ifdef MLP_DIAG_NOGELU   // diagnostic only (wrong results): the chunk loop without the GELU arithmetic
          if (level == 9) gq[v] = x;
          else if (level == 10 && (r & 1)) hbn[mt][ht >> 1][(ht & 1) * 2 + (r >> 1)] = pack_bf16x2(gq[v - 1], gq[v]);
          continue;
#endif
          if (level == 0) gs[v] = x * x;
          else if (level == 1) gq[v] = fmaf(-3.228973355362541e-06f, gs[v], 8.823775715427473e-05f);
          else if (level == 2) gq[v] = fmaf(gq[v], gs[v], 0.00036027710302732885f);
          else if (level == 3) gq[v] = fmaf(gq[v], gs[v], -0.10522669553756714f);
          else if (level == 4) gq[v] = fmaf(gq[v], gs[v], -2.3020453453063965f);
          else if (level == 5) gq[v] = x * gq[v];
          else if (level == 6) gq[v] = __builtin_amdgcn_exp2f(gq[v]);
          else if (level == 7) gq[v] = 1.0f + gq[v];
          else if (level == 8) gq[v] = __builtin_amdgcn_rcpf(gq[v]);
          else if (level == 9) gq[v] = x * gq[v];
          else if (r & 1) hbn[mt][ht >> 1][(ht & 1) * 2 + (r >> 1)] = pack_bf16x2(gq[v - 1], gq[v]);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef FRAG_PTR
#pragma unroll
    for (int ht = 0; ht < HT; ++ht)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) hcur[ht][mt] = hn[ht][mt];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int s2 = 0; s2 < S2; ++s2) hb[mt][s2] = hbn[mt][s2];
    bo = (bo == (NBUF - 1) * CHUNK_BYTES) ? 0u : bo + CHUNK_BYTES;
  }
#undef STAGE_PIECE
  STAMP(60);
  wait_vmcnt<0>();  // the trailing refills must have landed before the workgroup's LDS is released

  // ---- epilogue: the accumulators hold x + b2 + fc2(...): a lane stores 32 contiguous bytes per k-step ----
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int tok = tok0 + mt * 16 + l15;
    if (tok >= a.T) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      *reinterpret_cast<f32x4*>(a.out + (size_t)tok * C + 32 * (nt >> 1) + 8 * lg + 4 * (nt & 1)) = acc[nt][mt];
  }
  STAMP(61);
}

template <int C, int HC, int NBUF, int LB, bool PROJ>
int launch_one(const MlpArgs& a, hipStream_t s) {
  constexpr size_t lds = (size_t)NBUF * (4 * HC * C) + (size_t)(4 * C + 2 * HC) * sizeof(float);
  static std::atomic<uint64_t> lds_ok{0};  // dynamic LDS above 64 KiB needs an explicit opt-in, per device
  if (int e = ensure_dyn_lds(lds_ok, (const void*)k_mlp_stream<C, HC, NBUF, LB, PROJ>, (int)((int)lds))) return e;
  hipLaunchKernelGGL((k_mlp_stream<C, HC, NBUF, LB, PROJ>), dim3(cdiv(a.T, 128)), dim3(256), lds, s, a);
  return 0;
}

}  // namespace

bool mlp_stream_supported(int C) { return C == 384 || C == 192 || C == 96; }

int launch_mlp_stream(int C, int T, const float* x, float* out, const bf16* img, const float* b1, const float* b2,
                      const float* g, const float* b, hipStream_t s) {
  if (!mlp_stream_supported(C)) return SWG_E_UNSUPPORTED;
  if (T <= 0) return 0;
  MlpArgs a = {x, out, img, b1, b2, g, b, T, nullptr, 0, nullptr, nullptr, 0, 0, 1};
  SwgProfScope ps(C == 384 ? KID_MLP_STREAM_384 : KID_MLP_STREAM, 16.0 * T * (double)C * C, 8.0 * T * (double)C + 16.0 * C * C, s);
  int rc = (C == 384) ? launch_one<384, 32, 3, 1, false>(a, s) : launch_one<192, 32, 3, 1, false>(a, s);
  if (rc) return rc;
  SWG_LAUNCH_CHECK();
  return 0;
}

// attention tail + MLP half in one kernel (PROJ form above).  x_in: the BLOCK input (the shortcut is recomputed from
// it), o: attention output [T, ldo] bf16, img: projection fragments followed by the mlp_stream image.  x_in == out is
// allowed (a wave reads only the rows it writes).
bool mlp_stream_proj_supported(int C) { return C == 384 || C == 192; }
int launch_mlp_stream_proj(int C, int T, int L, const float* x_in, const bf16* o, int ldo, const float* cond, int cond_stride,
                           int cond_off, const float* proj_b, float* out, const bf16* img, const float* b1,
                           const float* b2, const float* g, const float* b, hipStream_t s) {
  if (!mlp_stream_proj_supported(C)) return SWG_E_UNSUPPORTED;
  if (T <= 0) return 0;
  if (L <= 0 || T % L || ldo < C || ldo % 8) return SWG_E_SHAPE;  // the conditioning row is looked up per token
  if (!x_in || !o || !cond || !proj_b || !out || !img) return SWG_E_NULL;
  MlpArgs a = {x_in, out, img, b1, b2, g, b, T, o, ldo, proj_b, cond, cond_stride, cond_off, L};
  SwgProfScope ps(C == 384 ? KID_MLP_PROJ_384 : KID_MLP_PROJ_192, 18.0 * T * (double)C * C,
                  10.0 * T * (double)C + 18.0 * C * C, s);
  int rc = (C == 384) ? launch_one<384, 32, 3, 1, true>(a, s) : launch_one<192, 32, 3, 1, true>(a, s);
  if (rc) return rc;
  SWG_LAUNCH_CHECK();
  return 0;
}
