// Fused PatchBreakup for the last two decoder levels (Cin = 384 -> Co = 96, Cin = 768 -> Co = 192):
// model/swin_gnn/swin_gnn.py:743 + :374-403,
//     y   = LayerNorm_{Cin}( cat([x, skip], -1) Wpre^T )
//     out[b, 2y+dy, 2x+dx, :] = LayerNorm_{Co}( y[b, y, x, k Co : (k+1) Co] ) Wpost^T,   (dy, dx) = (k & 1, k >> 1)
// in ONE pass over the two fp32 inputs.  Before: cast + concat pass (-> bf16), pre_linear GEMM (-> fp32 y), and the
// breakup tail kernel (y -> out): y and the bf16 operand made two round trips through HBM (16 B per input element on
// top of the 8 B that are necessary).
// Skeleton of merge_fused.hip: 32 tokens per wave as bf16 MFMA fragments (a plain cast here: no norm in front of
// pre_linear), Wpre [Cin, Cin] streams through a 3-deep LDS-DMA ring in 24-KiB chunks of 32 output features, all Cin
// outputs of the 32 tokens stay in accumulators (192 registers).  Then, on the accumulators:
//   * LayerNorm(Cin) per token (a token's features sit in one lane column: in-lane sums + two row swaps);
//   * per quarter k: LayerNorm(Co) over accumulator tiles 6k .. 6k+5, whose pairs (2s, 2s+1) ARE the B operand of the
//     post_linear k-step s once Wpost's input columns are put in the accumulator-pair slot order (pi of engine.py);
//     Wpost (18 fragments, 18 KiB) is resident in LDS;
//   * 16-byte stores of the [4 x 32 tokens, Co] result rows.
#include "common.h"

namespace {

struct ExpandArgs {
  const float* x;     // [T, Cin/2]
  const float* skip;  // [T, Cin/2]
  const bf16* img_pre;   // engine.py merge_image(pre_linear.weight, 32)
  const bf16* img_post;  // engine.py expand_post_image(post_linear.weight)
  const float* g1; const float* b1;  // LayerNorm(Cin)
  const float* g2; const float* b2;  // LayerNorm(Co)
  float* out;         // [4T, Co]
  int T, res;
};

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// MTW: 16-token MFMA tiles per wave (2 at Co = 96; 1 at Co = 192, where the Cin accumulators of 16 tokens are already
// 192 registers); NC: output features per 24-KiB chunk of Wpre.
template <int CO, int NC, int MTW, int NBUF>
__global__ __launch_bounds__(256, 1) void k_expand_fused(ExpandArgs a) {
  constexpr int CIN = 4 * CO, CH = 2 * CO;
  constexpr int KS = CIN / 32;  // k-steps of pre_linear
  constexpr int HT = NC / 16, NCH = CIN / NC;
  constexpr int NT = CIN / 16;  // accumulator tiles per token tile
  constexpr int CHUNK_BYTES = NC * CIN * 2;
  constexpr int PIECES = CHUNK_BYTES / 4096, FPC = CHUNK_BYTES / 1024;
  constexpr int AHEAD = 6;
  constexpr int QT = CO / 16;   // accumulator tiles per quarter
  constexpr int S2 = CO / 32;   // k-steps of post_linear
  constexpr int PF = S2 * QT;   // Wpost fragments
  static_assert(FPC == KS * HT && FPC % PIECES == 0 && CHUNK_BYTES % 4096 == 0, "chunk shape");
  static_assert(NBUF == 3 && NCH >= 3 && (PIECES * (NBUF - 1) + (PF + 3) / 4) < 64, "vmcnt is 6 bits");
  static_assert(CH % 32 == 0 && QT % 2 == 0, "halves and quarters are whole k-steps");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // NBUF x CHUNK_BYTES | Wpost PF KiB | g1 b1 [CIN] | g2 b2 [CO]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int tok0 = blockIdx.x * (64 * MTW) + wave * (16 * MTW);
  const unsigned smem0 = lds_addr(smem);
  const char* img_lane = reinterpret_cast<const char*>(a.img_pre) + lane * 16;
#define STAGE_PIECE(chunk, buf_off, i)                                                        \
  lds_dma16(img_lane + (size_t)(chunk) * CHUNK_BYTES + ((i) * 4 + wave) * 1024,               \
            smem0 + (buf_off) + ((i) * 4 + wave) * 1024)
  constexpr int POST_OFF = NBUF * CHUNK_BYTES;
  float* sg = reinterpret_cast<float*>(smem + POST_OFF + PF * 1024);  // g1 [CIN] b1 [CIN] g2 [CO] b2 [CO]

  // LayerNorm constants: loads issued first (see merge_fused.hip), stored to LDS below
  constexpr int NCONST = 2 * CIN + 2 * CO, GB = (NCONST + 255) / 256;
  float cv[GB];
#pragma unroll
  for (int i = 0; i < GB; ++i) {
    const int c = min(tid + 256 * i, NCONST - 1);
    cv[i] = c < CIN ? a.g1[c] : c < 2 * CIN ? a.b1[c - CIN] : c < 2 * CIN + CO ? a.g2[c - 2 * CIN] : a.b2[c - 2 * CIN - CO];
  }
  // token rows: k-steps [0, KS/2) come from x, [KS/2, KS) from skip (torch.cat([x, skip], -1), :743)
  bf16x8 xa[MTW][KS];
  {
    f32x4 xr[MTW][KS][2];
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) {
      const size_t tok = (size_t)min(tok0 + mt * 16 + l15, a.T - 1);
      const float* px = a.x + tok * CH + 8 * lg;
      const float* ps = a.skip + tok * CH + 8 * lg;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float* p = (ks < KS / 2) ? px + 32 * ks : ps + 32 * (ks - KS / 2);
        xr[mt][ks][0] = *reinterpret_cast<const f32x4*>(p);
        xr[mt][ks][1] = *reinterpret_cast<const f32x4*>(p + 4);
      }
    }
    // Wpost into LDS and the ring priming queue behind the rows (vector-memory operations return in order)
#pragma unroll
    for (int i = 0; i < (PF + 3) / 4; ++i)
      if (i * 4 + wave < PF)
        lds_dma16(reinterpret_cast<const char*>(a.img_post) + lane * 16 + (size_t)(i * 4 + wave) * 1024,
                  smem0 + POST_OFF + (i * 4 + wave) * 1024);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) STAGE_PIECE(0, 0, i);
#pragma unroll
    for (int i = 0; i < PIECES; ++i) STAGE_PIECE(1, CHUNK_BYTES, i);
#pragma unroll
    for (int i = 0; i < GB; ++i) {
      const int c = tid + 256 * i;
      if (c < NCONST) sg[c] = cv[i];
    }
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt)
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        bf16x8 f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f[e] = (bf16)xr[mt][ks][0][e];
          f[4 + e] = (bf16)xr[mt][ks][1][e];
        }
        xa[mt][ks] = f;
      }
  }

  __builtin_amdgcn_sched_barrier(0);  // the accumulators take over the registers of the raw rows: no early zeroing
  f32x4 acc[NT][MTW];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mt = 0; mt < MTW; ++mt) acc[t][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < NCH; ++j) {
    // chunk j (and, older than it, Wpost) has landed in this wave's view; the pieces of chunk j+1 may stay in flight
    wait_vmcnt<(NBUF - 2) * PIECES>();
    if (j == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's constants are in LDS
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");  // no LDS read of this iteration may move above the barrier
    const unsigned bo = (unsigned)(j % NBUF) * CHUNK_BYTES;
    const unsigned prev = (unsigned)((j + NBUF - 1) % NBUF) * CHUNK_BYTES;
    const int next_chunk = (j + NBUF - 1 < NCH) ? j + NBUF - 1 : NCH - 1;  // past the end: a dead slot is re-filled
    const char* cb = smem + bo + lane * 16;
    bf16x8 fr[AHEAD + 1];
#pragma unroll
    for (int p = 0; p < AHEAD; ++p) fr[p] = *reinterpret_cast<const bf16x8*>(cb + p * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < FPC; ++p) {
      const bf16x8 w = fr[p % (AHEAD + 1)];
      const int ks = p / HT, t = j * HT + p % HT;  // image order: k-step major, output tile minor
#pragma unroll
      for (int mt = 0; mt < MTW; ++mt)
        acc[t][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, xa[mt][ks], acc[t][mt], 0, 0, 0);
      if (p + AHEAD < FPC) fr[(p + AHEAD) % (AHEAD + 1)] = *reinterpret_cast<const bf16x8*>(cb + (p + AHEAD) * 1024);
      if (p % (FPC / PIECES) == 1) STAGE_PIECE(next_chunk, prev, p / (FPC / PIECES));
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef STAGE_PIECE
  wait_vmcnt<0>();  // the trailing refills must have landed before the workgroup's LDS is released

  // ---- tail on the accumulators.  Row 4*lg + r of tile t is feature 16*t + 4*lg + r of token l15 ----
  const float* g1 = sg + 4 * lg; const float* b1 = sg + CIN + 4 * lg;
  const float* g2 = sg + 2 * CIN + 4 * lg; const float* b2 = sg + 2 * CIN + CO + 4 * lg;
  const char* wp = smem + POST_OFF + lane * 16;
  const int L = a.res * a.res;
#pragma unroll
  for (int mt = 0; mt < MTW; ++mt) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t) s += (acc[t][mt][0] + acc[t][mt][1]) + (acc[t][mt][2] + acc[t][mt][3]);
    const float mean = rows_sum(s) * (1.0f / CIN);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) { const float d = acc[t][mt][r] - mean; q += d * d; }
    const float rstd = ln_rstd(rows_sum(q) * (1.0f / CIN));
    const int tok = tok0 + mt * 16 + l15;
    const int tc = min(tok, a.T - 1);
    const int bi = tc / L, rem = tc - bi * L;
    const int yy = rem / a.res, xx = rem - yy * a.res;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      // norm over Cin (its affine parameters are indexed by the position in the Cin row), then the quarter's statistics
      f32x4 z[QT];
      float s2 = 0.f;
#pragma unroll
      for (int t = 0; t < QT; ++t) {
        const f32x4 ga = *reinterpret_cast<const f32x4*>(g1 + 16 * (k * QT + t));
        const f32x4 ba = *reinterpret_cast<const f32x4*>(b1 + 16 * (k * QT + t));
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          z[t][r] = (acc[k * QT + t][mt][r] - mean) * rstd * ga[r] + ba[r];
          s2 += z[t][r];
        }
      }
      const float mean2 = rows_sum(s2) * (1.0f / CO);
      float q2 = 0.f;
#pragma unroll
      for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { const float d = z[t][r] - mean2; q2 += d * d; }
      const float rstd2 = ln_rstd(rows_sum(q2) * (1.0f / CO));
      bf16x8 zf[S2];
#pragma unroll
      for (int s2i = 0; s2i < S2; ++s2i) {
        const f32x4 ga = *reinterpret_cast<const f32x4*>(g2 + 32 * s2i);
        const f32x4 gb = *reinterpret_cast<const f32x4*>(g2 + 32 * s2i + 16);
        const f32x4 ba = *reinterpret_cast<const f32x4*>(b2 + 32 * s2i);
        const f32x4 bb = *reinterpret_cast<const f32x4*>(b2 + 32 * s2i + 16);
        bf16x8 f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f[r] = (bf16)((z[2 * s2i][r] - mean2) * rstd2 * ga[r] + ba[r]);
          f[4 + r] = (bf16)((z[2 * s2i + 1][r] - mean2) * rstd2 * gb[r] + bb[r]);
        }
        zf[s2i] = f;
      }
      f32x4 o[QT];
#pragma unroll
      for (int nt = 0; nt < QT; ++nt) o[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s2i = 0; s2i < S2; ++s2i)
#pragma unroll
        for (int nt = 0; nt < QT; ++nt) {
          const bf16x8 w = *reinterpret_cast<const bf16x8*>(wp + (s2i * QT + nt) * 1024);
          o[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, zf[s2i], o[nt], 0, 0, 0);
        }
      if (tok < a.T) {
        // quarter k of token (b, y, x) is sub-token (2y + dy, 2x + dx), (dy, dx) = (k & 1, k >> 1): swin_gnn.py:389-397
        const size_t drow = (size_t)bi * 4 * L + (size_t)(2 * yy + (k & 1)) * 2 * a.res + (2 * xx + (k >> 1));
        float* orow = a.out + drow * CO + 4 * lg;
#pragma unroll
        for (int nt = 0; nt < QT; ++nt) *reinterpret_cast<f32x4*>(orow + 16 * nt) = o[nt];
      }
    }
  }
}

template <int CO, int NC, int MTW>
int launch_one(const ExpandArgs& a, hipStream_t s) {
  constexpr size_t lds = (size_t)3 * (NC * 4 * CO * 2) + (size_t)(CO / 32) * (CO / 16) * 1024 + (size_t)(10 * CO) * sizeof(float);
  static_assert(lds <= 160 * 1024, "LDS per CU");
  static std::atomic<uint64_t> lds_ok{0};  // dynamic LDS above 64 KiB needs an explicit opt-in, per device
  if (int e = ensure_dyn_lds(lds_ok, (const void*)k_expand_fused<CO, NC, MTW, 3>, (int)lds)) return e;
  hipLaunchKernelGGL((k_expand_fused<CO, NC, MTW, 3>), dim3(cdiv(a.T, 64 * MTW)), dim3(256), lds, s, a);
  return 0;
}

}  // namespace

bool expand_fused_supported(int Cin) { return Cin == 384 || Cin == 768; }

int launch_expand_fused(int Cin, int B, int res, const float* x, const float* skip, const bf16* img_pre,
                        const bf16* img_post, const float* g1, const float* b1, const float* g2, const float* b2,
                        float* out, hipStream_t s) {
  if (!expand_fused_supported(Cin)) return SWG_E_UNSUPPORTED;
  if (!x || !skip || !img_pre || !img_post || !g1 || !b1 || !g2 || !b2 || !out) return SWG_E_NULL;
  const int T = B * res * res;
  if (T <= 0) return 0;
  ExpandArgs a = {x, skip, img_pre, img_post, g1, b1, g2, b2, out, T, res};
  SwgProfScope ps(KID_EXPAND_FUSED, 2.0 * T * (double)Cin * Cin + 2.0 * T * (double)Cin * (Cin / 4),
                  4.0 * T * (double)Cin + 4.0 * T * (double)Cin + 2.0 * Cin * Cin, s);
  int rc = (Cin == 384) ? launch_one<96, 32, 2>(a, s) : launch_one<192, 16, 1>(a, s);
  if (rc) return rc;
  SWG_LAUNCH_CHECK();
  return 0;
}
