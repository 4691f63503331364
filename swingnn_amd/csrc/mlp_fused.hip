// Fused MLP half of a Swin block (model/swin_gnn/swin_gnn.py:275 with Mlp :19-25):
//     x_out = x + fc2( GELU( fc1( LayerNorm2(x) ) ) )
// in ONE pass over the residual stream: each fp32 token row is read once and written once; the 4C-wide hidden
// activation never leaves the register file.
//
// gfx950 mapping
//  * A wavefront owns 32 tokens (two 16-token MFMA column tiles).  It loads their rows from HBM directly in
//    MFMA operand layout (lane = token l&15, 8 consecutive channels per 32-deep k-step), computes the
//    LayerNorm statistics with two xor-shuffles, and keeps the normalised bf16 fragments in VGPRs.
//  * The hidden dimension is processed in chunks of 64: H^T = W1c . A^T (features on accumulator rows, tokens
//    on lanes) -> +bias, GELU -> the accumulator registers ARE the B operand of out^T += W2c . H^T, because the
//    k-slot permutation they imply is baked into the host-prepared W2 image.  No LDS round trip for activations.
//  * Weights stream through LDS: the host lays W1/W2 out as a per-chunk image already in swizzled fragment order,
//    so staging is a linear LDS-DMA copy (global_load_lds, 16 B per lane), double buffered, one barrier per chunk.
#include "common.h"

namespace {

constexpr int HC = 64;  // hidden features per chunk

struct MlpArgs {
  const float* x;
  float* out;
  const bf16* img;   // weight image, see engine.py mlp_image()
  const float* b1;   // [4C]
  const float* b2;   // [C]
  const float* g;    // LN2 weight
  const float* b;    // LN2 bias
  int T;
  // PROJ form only (head_dim-20 blocks: attention tail folded in front of the MLP)
  const bf16* o;      // attention output [T, ldo], 32-slot head groups as k_wattn_ring<hd20> stores them
  int ldo;
  const bf16* pimg;   // projection weight as fragments (engine.py proj_frag_image)
  const float* pb;    // proj bias [CT]
  const float* cond;  // conditioning rows [B, cond_stride]; this block's (scale | shift) at cond_off
  int cond_stride, cond_off, L;
};

// 16-byte slot swizzle of a 64-byte LDS row (4 slots): conflict-free for ds_read_b128 fragment reads
__device__ __forceinline__ int slot_swz(int row, int ch) { return ch ^ ((-(row >> 2)) & 3); }

// CT < C (the dim-60 networks: CT = 60 / 120 true channels in C = 64 / 128 operand slots): rows, LayerNorm constants and
// biases hold CT entries; the slots CT..C-1 are loaded as zero, stay out of the variance, and are not stored.  The
// weight image is zero-padded to [4C, C] / [C, 4C] (engine.py), so padded hidden features are gelu(0) = 0.
// PROJ (C = 64 / 128 with CT = 60 / 120, head_dim 20): the kernel starts from the block INPUT x and the attention output
// o: the fc2 accumulators start as shortcut + proj bias (the shortcut silu(shift + x (1 + scale)) recomputed, :238-242),
// KO = CT / 20 k-steps of Wproj fragments read straight from global memory (12 / 48 KiB, L2-resident) accumulate
// o Wproj^T into them, LayerNorm2 reads the accumulators, and they go on as residual + fc2 bias: the projection GEMM
// launch and the round trip of x through memory go.
template <int C, bool KEEPX, int CT = C, bool PROJ = false>
__global__ __launch_bounds__(256, 2) void k_mlp_fused(MlpArgs a) {
  static_assert(CT % 4 == 0 && CT <= C && C - CT < 32, "padding lives in the last k-step, in whole float4 groups");
  static_assert(!PROJ || (CT % 20 == 0 && !KEEPX), "PROJ: head_dim 20, residual in the accumulators");
  constexpr bool PAD = CT != C;
  constexpr int KO = PROJ ? CT / 20 : 1;   // 32-slot head groups of o
  constexpr int KS = C / 32;       // k-steps of fc1
  constexpr int NT = C / 16;       // output feature tiles of fc2
  constexpr int NCH = 4 * C / HC;  // hidden chunks
  constexpr int CHUNK_ELEMS = 2 * HC * C;          // W1 part (HC x C) + W2 part (C x HC)
  constexpr int CHUNK_BYTES = CHUNK_ELEMS * 2;
  constexpr int DMA_PER_WAVE = CHUNK_BYTES / (4 * 1024);  // 1 KiB per wave-instruction, 4 waves
  static_assert(CHUNK_BYTES % 4096 == 0, "chunk must be a whole number of 1-KiB pieces per wave");
  extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 x CHUNK_BYTES

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  const int tok0 = blockIdx.x * 128 + wave * 32;

  // ---- stage chunk `hc` into LDS buffer `buf` by LDS-DMA (wave-uniform LDS base + lane*16) ----
  auto stage = [&](int hc, int buf) {
    const char* src = reinterpret_cast<const char*>(a.img) + (size_t)hc * CHUNK_BYTES;
#pragma unroll
    for (int i = 0; i < DMA_PER_WAVE; ++i) {
      const int piece = i * 4 + wave;  // 1-KiB piece index inside the chunk
      lds_dma16(src + piece * 1024 + lane * 16, lds_addr(smem) + buf * CHUNK_BYTES + piece * 1024);
    }
  };
  stage(0, 0);
  float* sb1 = reinterpret_cast<float*>(smem + 2 * CHUNK_BYTES);  // fc1 bias [4C] (read every chunk)
  for (int i = tid; i < 4 * C; i += 256) sb1[i] = (!PAD || i < 4 * CT) ? a.b1[i] : 0.f;

  // ---- load the wave's 32 token rows in operand layout, LayerNorm2 in registers ----
  bf16x8 xa[2][KS];
  float xr[KEEPX ? 2 : 1][KEEPX ? KS : 1][8];  // fp32 rows kept for the residual add (KEEPX)
  f32x4 acc[NT][2];
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  // float4 group hh of k-step ks holds channels 32ks + 8lg + 4hh ..+3: real below CT (only the last k-step pads)
  auto real = [&](int ks, int hh) { return !PAD || ks < KS - 1 || 32 * ks + 8 * lg + 4 * hh < CT; };
  if (PROJ) {
    bf16x8 oa[2][KO];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int tok = min(tok0 + mt * 16 + l15, a.T - 1);
      const float* row = a.x + (size_t)tok * CT;
      const float* crow = a.cond + (size_t)(tok / a.L) * a.cond_stride + a.cond_off;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
          const int c = 32 * ks + 8 * lg + 4 * hh;
          f32x4 r = {0.f, 0.f, 0.f, 0.f};
          if (real(ks, hh)) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(row + c);
            const f32x4 sc = *reinterpret_cast<const f32x4*>(crow + c);
            const f32x4 sh = *reinterpret_cast<const f32x4*>(crow + CT + c);
            const f32x4 pb = *reinterpret_cast<const f32x4*>(a.pb + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) r[e] = silu_f(sh[e] + xv[e] * (sc[e] + 1.f)) + pb[e];
          }
          acc[2 * ks + hh][mt] = r;
          if (hh) __builtin_amdgcn_sched_barrier(0);   // register peak: one k-step's loads hoisted at most
        }
      const bf16* orow = a.o + (size_t)tok * a.ldo + 8 * lg;
#pragma unroll
      for (int ko = 0; ko < KO; ++ko) oa[mt][PROJ ? ko : 0] = *reinterpret_cast<const bf16x8*>(orow + 32 * ko);
    }
    // acc^T += Wproj o^T: fragment (ko, nt), k-step major, rows in rho order, columns in o's slot order
    const bf16* pf = a.pimg + lane * 8;
#pragma unroll
    for (int f = 0; f < KO * NT; ++f) {
      const bf16x8 w = *reinterpret_cast<const bf16x8*>(pf + f * 512);
      const int ko = f / NT, nt = f % NT;
      acc[nt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, oa[0][PROJ ? ko : 0], acc[nt][0], 0, 0, 0);
      acc[nt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, oa[1][PROJ ? ko : 0], acc[nt][1], 0, 0, 0);
      if ((f & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // register peak: at most eight fragment loads in flight
    }
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int tok = min(tok0 + mt * 16 + l15, a.T - 1);
    const float* row = a.x + (size_t)tok * CT;
    float v[KS][8];
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float4 p0, p1;
      if (PROJ) {   // the attention half's output: accumulator rows = the channels this lane holds for k-step ks
        const f32x4 q0 = acc[2 * ks][mt], q1 = acc[2 * ks + 1][mt];
        p0 = make_float4(q0[0], q0[1], q0[2], q0[3]);
        p1 = make_float4(q1[0], q1[1], q1[2], q1[3]);
        // ... and they go on as residual + fc2 bias
        const f32x4 c0 = real(ks, 0) ? *reinterpret_cast<const f32x4*>(a.b2 + 32 * ks + 8 * lg) : (f32x4){0.f, 0.f, 0.f, 0.f};
        const f32x4 c1 = real(ks, 1) ? *reinterpret_cast<const f32x4*>(a.b2 + 32 * ks + 8 * lg + 4) : (f32x4){0.f, 0.f, 0.f, 0.f};
        acc[2 * ks][mt] = q0 + c0;
        acc[2 * ks + 1][mt] = q1 + c1;
      } else {
        p0 = real(ks, 0) ? *reinterpret_cast<const float4*>(row + 32 * ks + 8 * lg) : z4;
        p1 = real(ks, 1) ? *reinterpret_cast<const float4*>(row + 32 * ks + 8 * lg + 4) : z4;
      }
      v[ks][0] = p0.x; v[ks][1] = p0.y; v[ks][2] = p0.z; v[ks][3] = p0.w;
      v[ks][4] = p1.x; v[ks][5] = p1.y; v[ks][6] = p1.z; v[ks][7] = p1.w;
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[ks][j];
    }
    s = rows_sum(s);
    const float mean = s * (1.0f / CT);
    float q = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float d = real(ks, j >> 2) ? v[ks][j] - mean : 0.f;
        q += d * d;
      }
    q = rows_sum(q);
    const float rstd = ln_rstd(q * (1.0f / CT));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      // padded slots: g = b = 0, so the operand is 0
      const float4 g0 = real(ks, 0) ? *reinterpret_cast<const float4*>(a.g + 32 * ks + 8 * lg) : z4;
      const float4 g1 = real(ks, 1) ? *reinterpret_cast<const float4*>(a.g + 32 * ks + 8 * lg + 4) : z4;
      const float4 b0 = real(ks, 0) ? *reinterpret_cast<const float4*>(a.b + 32 * ks + 8 * lg) : z4;
      const float4 b1 = real(ks, 1) ? *reinterpret_cast<const float4*>(a.b + 32 * ks + 8 * lg + 4) : z4;
      const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
      const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      bf16x8 f;
#pragma unroll
      for (int j = 0; j < 8; ++j) f[j] = (bf16)((v[ks][j] - mean) * rstd * gg[j] + bb[j]);
      xa[mt][ks] = f;
      if (KEEPX) {
#pragma unroll
        for (int j = 0; j < 8; ++j) xr[mt][ks][j] = v[ks][j];
      }
    }
  }

  if (!PROJ) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) acc[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // fragment byte offsets inside a chunk image (this lane)
  int w1_off[4];   // row nt*16+l15 of the W1 part, k-step 0; k-step s adds s*HC*64 bytes
  int w2_off[NT];  // row nt*16+l15 of the W2 part, k-step 0; k-step s adds s*C*64 bytes
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int r = nt * 16 + l15;
    w1_off[nt] = (r * 4 + slot_swz(r, lg)) * 16;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int r = nt * 16 + l15;
    w2_off[nt] = HC * C * 2 + (r * 4 + slot_swz(r, lg)) * 16;
  }

  for (int hc = 0; hc < NCH; ++hc) {
    const int buf = hc & 1;
    // chunk hc has landed (LDS-DMA is only ordered by this wave's vmcnt, the compiler does not wait for it),
    // and after the barrier every wave is done reading the other buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (hc + 1 < NCH) stage(hc + 1, buf ^ 1);
    const char* cb = smem + buf * CHUNK_BYTES;

    // ---- fc1: H^T[HC x 32 tokens] = W1c . A^T ; fragment reads run one k-step ahead of the MFMAs ----
    f32x4 h[4][2];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) h[nt][mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    bf16x8 w1f[2][4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) w1f[0][nt] = *reinterpret_cast<const bf16x8*>(cb + w1_off[nt]);
    bf16x8 w2f[2][NT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
          w1f[(ks + 1) & 1][nt] = *reinterpret_cast<const bf16x8*>(cb + w1_off[nt] + (ks + 1) * (HC * 64));
      } else {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) w2f[0][nt] = *reinterpret_cast<const bf16x8*>(cb + w2_off[nt]);
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
          h[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w1f[ks & 1][nt], xa[mt][ks], h[nt][mt], 0, 0, 0);
    }
    // ---- bias + GELU; the accumulators become the B operand of fc2 (k-slot order matches the W2 image) ----
    bf16x8 hb[2][2];  // [mt][k-step of fc2]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      const float4 bb = *reinterpret_cast<const float4*>(sb1 + hc * HC + nt * 16 + 4 * lg);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int s2 = nt >> 1, o = (nt & 1) * 4;
        hb[mt][s2][o + 0] = (bf16)gelu_f(h[nt][mt][0] + bb.x);
        hb[mt][s2][o + 1] = (bf16)gelu_f(h[nt][mt][1] + bb.y);
        hb[mt][s2][o + 2] = (bf16)gelu_f(h[nt][mt][2] + bb.z);
        hb[mt][s2][o + 3] = (bf16)gelu_f(h[nt][mt][3] + bb.w);
      }
    }
    // ---- fc2: out^T[C x 32 tokens] += W2c . H^T ----
    {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) w2f[1][nt] = *reinterpret_cast<const bf16x8*>(cb + w2_off[nt] + C * 64);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int mt = 0; mt < 2; ++mt)
            acc[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w2f[s2][nt], hb[mt][s2], acc[nt][mt], 0, 0, 0);
    }
  }

  // ---- epilogue: + b2 + residual, fp32 store.  The fc2 rows of the image are ordered so that accumulator
  // row 4*lg+r of feature tile nt = 2*ks+hh is channel 32*ks + 8*lg + 4*hh + r: exactly the channels this lane
  // loaded at k-step ks, so the residual comes from registers (KEEPX) and a lane stores 32 contiguous bytes.
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int tok = tok0 + mt * 16 + l15;
    if (tok >= a.T) continue;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int ks = nt >> 1, hh = nt & 1;
      const int c = 32 * ks + 8 * lg + 4 * hh;
      if (PAD && ks == KS - 1 && c >= CT) continue;
      if (PROJ) {   // residual and bias are in the accumulators already
        *reinterpret_cast<f32x4*>(a.out + (size_t)tok * CT + c) = acc[nt][mt];
        continue;
      }
      const float4 bb = *reinterpret_cast<const float4*>(a.b2 + c);
      float4 r4;
      if (KEEPX) r4 = make_float4(xr[mt][ks][4 * hh], xr[mt][ks][4 * hh + 1], xr[mt][ks][4 * hh + 2], xr[mt][ks][4 * hh + 3]);
      else r4 = *reinterpret_cast<const float4*>(a.x + (size_t)tok * CT + c);
      float4 o;
      o.x = acc[nt][mt][0] + bb.x + r4.x;
      o.y = acc[nt][mt][1] + bb.y + r4.y;
      o.z = acc[nt][mt][2] + bb.z + r4.z;
      o.w = acc[nt][mt][3] + bb.w + r4.w;
      *reinterpret_cast<float4*>(a.out + (size_t)tok * CT + c) = o;
    }
  }
}

}  // namespace

// C = 96; C = 60 / 120 (head_dim-20 networks) on zero-padded 64 / 128-slot operands
bool mlp_fused_supported(int C) { return C == 96 || C == 60 || C == 120; }
// PROJ form: the head_dim-20 blocks whose attention core is k_wattn_ring<hd20>
bool mlp_fused_proj_supported(int C) { return C == 60 || C == 120; }

int launch_mlp_fused_proj(int C, int T, int L, const float* x_in, const bf16* o, int ldo, const float* cond, int cond_stride,
                          int cond_off, const float* proj_b, float* out, const bf16* pimg, const bf16* img, const float* b1,
                          const float* b2, const float* g, const float* b, hipStream_t s) {
  if (!mlp_fused_proj_supported(C)) return SWG_E_UNSUPPORTED;
  if (!x_in || !o || !cond || !proj_b || !out || !pimg || !img) return SWG_E_NULL;
  if (L <= 0 || T % L || ldo < (C / 20) * 32 || ldo % 8) return SWG_E_SHAPE;
  if (T <= 0) return 0;
  MlpArgs a = {x_in, out, img, b1, b2, g, b, T, o, ldo, pimg, proj_b, cond, cond_stride, cond_off, L};
  const int Cs = round_up(C, 32);
  const size_t lds = (size_t)2 * (2 * HC * Cs) * 2 + (size_t)4 * Cs * sizeof(float);
  SwgProfScope ps(KID_FUSED_MLP, 18.0 * T * (double)C * C, 10.0 * T * (double)C + 18.0 * C * C, s);
  dim3 grid(cdiv(T, 128)), block(256);
  static std::atomic<uint64_t> lds_ok{0};
  if (int e = ensure_dyn_lds(lds_ok, (const void*)k_mlp_fused<128, false, 120, true>, (int)(2 * (2 * HC * 128) * 2 + 4 * 128 * 4))) return e;
  if (C == 60) hipLaunchKernelGGL((k_mlp_fused<64, false, 60, true>), grid, block, lds, s, a);
  else hipLaunchKernelGGL((k_mlp_fused<128, false, 120, true>), grid, block, lds, s, a);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_mlp_fused(int C, int T, const float* x, float* out, const bf16* img, const float* b1, const float* b2,
                     const float* g, const float* b, hipStream_t s) {
  if (!mlp_fused_supported(C)) return SWG_E_UNSUPPORTED;
  if (T <= 0) return 0;
  MlpArgs a = {x, out, img, b1, b2, g, b, T, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 1};
  const int Cs = round_up(C, 32);   // operand slots
  const size_t lds = (size_t)2 * (2 * HC * Cs) * 2 + (size_t)4 * Cs * sizeof(float);
  SwgProfScope ps(KID_FUSED_MLP, 16.0 * T * (double)C * C, 8.0 * T * (double)C + 16.0 * C * C, s);
  dim3 grid(cdiv(T, 128)), block(256);
  static std::atomic<uint64_t> lds_ok{0};  // dynamic LDS above 64 KiB needs an explicit opt-in, per device
  if (int e = ensure_dyn_lds(lds_ok, (const void*)k_mlp_fused<192, false>, (int)(2 * (2 * HC * 192) * 2 + 4 * 192 * 4))) return e;
  if (C == 96) hipLaunchKernelGGL((k_mlp_fused<96, true>), grid, block, lds, s, a);
  else if (C == 60) hipLaunchKernelGGL((k_mlp_fused<64, true, 60>), grid, block, lds, s, a);
  else if (C == 120) hipLaunchKernelGGL((k_mlp_fused<128, false, 120>), grid, block, lds, s, a);
  else hipLaunchKernelGGL((k_mlp_fused<192, false>), grid, block, lds, s, a);
  SWG_LAUNCH_CHECK();
  return 0;
}
