// Front half of the MLP backward of a Swin block (model/swin_gnn/swin_gnn.py:275 with Mlp :19-25) in ONE pass over the
// token rows, for C = 96, 192, 384 (every hidden-sized tensor of the unfused chain crosses HBM several times):
//     xn = LN2(x)            h  = xn W1^T + b1          a  = GELU(h)
//     da = dy W2             dh = da * GELU'(h)
// Outputs are the four bf16 GEMM operands the rest of the backward needs -- xn and dy (C wide), a and dh (4C wide);
// h and da never leave the register file.  This replaces LayerNorm, the fc1-recompute GEMM (writes h), the dy cast, the
// da GEMM (writes da) and the GELU pass (reads h, da; writes a, dh): 64 T C bytes of traffic become 28 T C.
// Same mapping as mlp_fused.hip: a wave owns 32 tokens (token on the lane), hidden chunks of 64 stream through a
// double-buffered LDS image by LDS-DMA; per chunk both products have the SAME shape ([64 hidden] x [32 tokens], K = C),
// so the image holds W1 rows and W2^T rows in one layout.  GELU and its derivative share x*x and the sigmoid:
// Phi(x) = 1 / (1 + 2^(x P(x^2))) (gelu_f of common.h), phi(x) = 2^(-x^2 / (2 ln 2)) / sqrt(2 pi).
#include "common.h"

namespace {


struct FrontArgs {
  const float* x;
  const float* dy;
  const bf16* img;  // engine.py mlp_bwd_front_image()
  const float* b1;  // fc1 bias [4C]
  const float* g;   // LN2 weight / bias
  const float* b;
  bf16* xn;         // [T, ldc]
  bf16* dyb;        // [T, ldc]
  bf16* act;        // [T, ldh]
  bf16* dh;         // [T, ldh]
  float* part;      // [workgroups, 5C]: per-workgroup column sums of dh (4C, -> fc1 bias gradient) and of dy (C, -> fc2's)
  int T, ldc, ldh;
};

__device__ __forceinline__ int slot_swz(int row, int ch) { return ch ^ ((-(row >> 2)) & 3); }

template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// HC: hidden features per chunk (64 at C = 96, 32 at C = 192, 16 at C = 384: two resident workgroups per CU each way)
// MT: 16-token row tiles per wave (2: 128 tokens per workgroup; 3 at C = 384: 192 tokens, one workgroup per CU with the
// whole register file -- 46080 tokens are then ONE round of 240 workgroups instead of 1.4 rounds of 360)
template <int C, int HC, int WGS, int MT>
__global__ __launch_bounds__(256, WGS) void k_mlp_bwd_front(FrontArgs a) {
  constexpr int KS = C / 32;
  constexpr int HT = HC / 16;
  constexpr int NCH = 4 * C / HC;
  constexpr int CHUNK_BYTES = 2 * HC * C * 2;  // W1 rows (HC x C) + W2^T rows (HC x C)
  constexpr int DMA_PER_WAVE = CHUNK_BYTES / 4096;
  static_assert(CHUNK_BYTES % 4096 == 0, "chunk must be a whole number of 1-KiB pieces per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x CHUNK_BYTES | b1 [4C] | column sums [5C]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int tok0 = blockIdx.x * (64 * MT) + wave * (16 * MT);

  auto stage = [&](int hc, int buf) {
    const char* src = reinterpret_cast<const char*>(a.img) + (size_t)hc * CHUNK_BYTES;
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
      const int piece = i * 4 + wave;
      lds_dma16(src + piece * 1024 + lane * 16, lds_addr(smem) + buf * CHUNK_BYTES + piece * 1024);
    }
  };
  stage(0, 0);
  float* sb1 = reinterpret_cast<float*>(smem + 2 * CHUNK_BYTES);
  float* ssum = sb1 + 4 * C;  // [4C] sum_t dh | [C] sum_t dy
  for (int i = tid; i < 4 * C; i += 256) sb1[i] = a.b1[i];
  for (int i = tid; i < 5 * C; i += 256) ssum[i] = 0.f;

  // ---- the wave's 32 rows of x (LayerNorm2 -> operand fragments, also stored) and of dy (cast, also stored) ----
  bf16x8 xa[MT][KS], da[MT][KS];
  __syncthreads();  // ssum is zero before the first column sums go in
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    // rows past the end are clamped to the last row: such lanes compute and STORE that row's values again (identical
    // bytes), so every wave issues the same number of stores -- the counted wait in the chunk loop relies on it
    const int tok = tok0 + mt * 16 + l15;
    const int tc = min(tok, a.T - 1);
    const float live = tok < a.T ? 1.0f : 0.0f;
    const float* row = a.x + (size_t)tc * C + 8 * lg;
    const float* drow = a.dy + (size_t)tc * C + 8 * lg;
    float v[KS][8];
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const f32x4 p0 = *reinterpret_cast<const f32x4*>(row + 32 * ks);
      const f32x4 p1 = *reinterpret_cast<const f32x4*>(row + 32 * ks + 4);
      const f32x4 d0 = *reinterpret_cast<const f32x4*>(drow + 32 * ks);
      const f32x4 d1 = *reinterpret_cast<const f32x4*>(drow + 32 * ks + 4);
      bf16x8 f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[ks][e] = p0[e]; v[ks][4 + e] = p1[e];
        s += p0[e] + p1[e];
        f[e] = (bf16)d0[e]; f[4 + e] = (bf16)d1[e];
      }
      // column sums of dy (fc2 bias gradient) over the 16 tokens of this lane group, straight into the LDS sums (a
      // register array over both row tiles would cost KS * 8 registers for the whole kernel)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float t0 = row16_sum(live * d0[e]), t1 = row16_sum(live * d1[e]);
        if (l15 == 0) {
          atomicAdd(&ssum[4 * C + 32 * ks + 8 * lg + e], t0);
          atomicAdd(&ssum[4 * C + 32 * ks + 8 * lg + 4 + e], t1);
        }
      }
      da[mt][ks] = f;
      *reinterpret_cast<bf16x8*>(a.dyb + (size_t)tc * a.ldc + 32 * ks + 8 * lg) = f;
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
      const f32x4 g0 = *reinterpret_cast<const f32x4*>(a.g + 32 * ks + 8 * lg);
      const f32x4 g1 = *reinterpret_cast<const f32x4*>(a.g + 32 * ks + 8 * lg + 4);
      const f32x4 b0 = *reinterpret_cast<const f32x4*>(a.b + 32 * ks + 8 * lg);
      const f32x4 b1 = *reinterpret_cast<const f32x4*>(a.b + 32 * ks + 8 * lg + 4);
      bf16x8 f;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        f[e] = (bf16)((v[ks][e] - mean) * rstd * g0[e] + b0[e]);
        f[4 + e] = (bf16)((v[ks][4 + e] - mean) * rstd * g1[e] + b1[e]);
      }
      xa[mt][ks] = f;
      *reinterpret_cast<bf16x8*>(a.xn + (size_t)tc * a.ldc + 32 * ks + 8 * lg) = f;
    }
  }
  float live2[MT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) live2[mt] = (tok0 + mt * 16 + l15 < a.T) ? 1.0f : 0.0f;

  int w_off[HT];  // row nt*16 + l15 of an HC-row part, k-step 0; k-step s adds HC*64 bytes; the W2^T part adds HC*C*2
#pragma unroll
  for (int nt = 0; nt < HT; ++nt) {
    const int r = nt * 16 + l15;
    w_off[nt] = (r * 4 + slot_swz(r, lg)) * 16;
  }

  for (int hc = 0; hc < NCH; ++hc) {
    const int buf = hc & 1;
    // chunk hc has landed: its LDS-DMA pieces are older than the 2 MT HT stores of the previous iteration, which may stay
    // in flight (a vmcnt(0) here would expose a store latency per chunk)
    if (hc == 0) wait_vm<0>();
    else wait_vm<2 * MT * HT>();
    __syncthreads();
    if (hc + 1 < NCH) stage(hc + 1, buf ^ 1);
    const char* cb = smem + buf * CHUNK_BYTES;

    f32x4 h[HT][MT], d[HT][MT];
#pragma unroll
    for (int nt = 0; nt < HT; ++nt)
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) h[nt][mt] = d[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 w1f[2][HT], w2f[2][HT];
#pragma unroll
    for (int nt = 0; nt < HT; ++nt) {
      w1f[0][nt] = *reinterpret_cast<const bf16x8*>(cb + w_off[nt]);
      w2f[0][nt] = *reinterpret_cast<const bf16x8*>(cb + HC * C * 2 + w_off[nt]);
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) {
#pragma unroll
        for (int nt = 0; nt < HT; ++nt) {
          w1f[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(cb + w_off[nt] + (ks + 1) * (HC * 64));
          w2f[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(cb + HC * C * 2 + w_off[nt] + (ks + 1) * (HC * 64));
        }
      }
#pragma unroll
      for (int nt = 0; nt < HT; ++nt)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          h[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[ks & 1][nt], xa[mt][ks], h[nt][mt], 0, 0, 0);
          d[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[ks & 1][nt], da[mt][ks], d[nt][mt], 0, 0, 0);
        }
    }
    // ---- bias, GELU and its derivative; accumulator row 4*lg + r of tile nt is hidden hc*64 + nt*16 + 4*lg + r ----
    float hs[HT][4];
#pragma unroll
    for (int nt = 0; nt < HT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) hs[nt][r] = 0.f;
#pragma unroll
    for (int nt = 0; nt < HT; ++nt) {
      const f32x4 bb = *reinterpret_cast<const f32x4*>(sb1 + hc * HC + nt * 16 + 4 * lg);
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
        float av[4], gv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float x = h[nt][mt][r] + bb[r];
          const float s = x * x;
          float q = fmaf(-3.228973355362541e-06f, s, 8.823775715427473e-05f);
          q = fmaf(q, s, 0.00036027710302732885f);
          q = fmaf(q, s, -0.10522669553756714f);
          q = fmaf(q, s, -2.3020453453063965f);
          const float cdf = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * q));
          const float pdf = 0.3989422804014327f * __builtin_amdgcn_exp2f(s * -0.72134752044448170368f);
          av[r] = x * cdf;
          gv[r] = d[nt][mt][r] * fmaf(x, pdf, cdf);
        }
        const int tc = min(tok0 + mt * 16 + l15, a.T - 1);
        const size_t o = (size_t)tc * a.ldh + hc * HC + nt * 16 + 4 * lg;
        *reinterpret_cast<uint2*>(a.act + o) = pack_bf16x4(av[0], av[1], av[2], av[3]);
        const uint2 gp = pack_bf16x4(gv[0], gv[1], gv[2], gv[3]);
        *reinterpret_cast<uint2*>(a.dh + o) = gp;
        // bias gradient from the ROUNDED values (what the weight-gradient GEMM sees), rows past the end excluded
        const bf16x4 gr = __builtin_bit_cast(bf16x4, gp);
#pragma unroll
        for (int r = 0; r < 4; ++r) hs[nt][r] += live2[mt] * (float)gr[r];
      }
    }
    // column sums of this chunk over the wave's 32 tokens (the 16 lanes of a lane group hold 16 tokens each)
#pragma unroll
    for (int nt = 0; nt < HT; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float t = row16_sum(hs[nt][r]);
        if (l15 == 0) atomicAdd(&ssum[hc * HC + nt * 16 + 4 * lg + r], t);
      }
  }
  __syncthreads();
  for (int i = tid; i < 5 * C; i += 256) a.part[(size_t)blockIdx.x * (5 * C) + i] = ssum[i];
}

template <int C, int HC, int WGS, int MT>
int launch_c(const FrontArgs& a, hipStream_t s) {
  constexpr size_t lds = (size_t)2 * (2 * HC * C * 2) + (size_t)9 * C * sizeof(float);
  static std::atomic<uint64_t> lds_ok{0};
  if (int e = ensure_dyn_lds(lds_ok, (const void*)k_mlp_bwd_front<C, HC, WGS, MT>, (int)lds)) return e;
  hipLaunchKernelGGL((k_mlp_bwd_front<C, HC, WGS, MT>), dim3(cdiv(a.T, 64 * MT)), dim3(256), lds, s, a);
  return 0;
}

}  // namespace

bool mlp_bwd_front_supported(int C) { return C == 96 || C == 192 || C == 384; }
// floats of the per-workgroup column-sum buffer `part` ([workgroups, 5C]: sums of dh, then of dy)
// tokens per workgroup (= rows of `part` per 5C floats)
int mlp_bwd_front_tokens_per_wg(int C) { return C == 384 ? 192 : 128; }
size_t mlp_bwd_front_part_floats(int C, int T) { return (size_t)cdiv(T, mlp_bwd_front_tokens_per_wg(C)) * 5 * C; }

int launch_mlp_bwd_front(int C, int T, const float* x, const float* dy, const bf16* img, const float* b1, const float* g,
                         const float* b, bf16* xn, bf16* dyb, int ldc, bf16* act, bf16* dh, int ldh, float* part,
                         hipStream_t s) {
  if (!mlp_bwd_front_supported(C)) return SWG_E_UNSUPPORTED;
  if (T <= 0) return 0;
  if (!x || !dy || !img || !b1 || !g || !b || !xn || !dyb || !act || !dh || !part) return SWG_E_NULL;
  if (ldc < C || ldc % 8 || ldh < 4 * C || ldh % 4) return SWG_E_SHAPE;
  FrontArgs a = {x, dy, img, b1, g, b, xn, dyb, act, dh, part, T, ldc, ldh};
  SwgProfScope ps(KID_BWD_ROWS, 8.0 * T * (double)C * C * 2.0, 8.0 * T * (double)C + 4.0 * T * (double)C + 16.0 * T * (double)C, s);
  int rc = (C == 96) ? launch_c<96, 64, 2, 2>(a, s) : (C == 192) ? launch_c<192, 32, 2, 2>(a, s) : launch_c<384, 32, 1, 3>(a, s);
  if (rc) return rc;
  SWG_LAUNCH_CHECK();
  return 0;
}
