// Memory-bound kernels of the SwinGNN denoiser path (gfx950): conditioning + LayerNorm prologues,
// patch merge / breakup re-layouts, patch embedding, noise embedding, read-out + symmetrise +
// preconditioning epilogue, and the EDM Heun updates.  All of them are HBM-bound: rows are read
// once as float4 by a lane group, statistics are reduced with wavefront shuffles, bf16 is only a
// storage format for the following MFMA GEMMs.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// Generic row kernel: LANES lanes own one output row, V float4 chunks per lane.
// ------------------------------------------------------------------------------------------------
enum RowMode { ROW_COND_LN = 0, ROW_MERGE_LN = 1, ROW_CONCAT = 2, ROW_LN_F32 = 3, ROW_BREAKUP_LN = 4 };

struct RowArgs {
  const float* x;   // primary source
  const float* x2;  // second source (concat)
  int rows;         // output rows
  int C;            // logical row width (floats) over which LN runs
  int L;            // tokens per graph (cond row lookup)
  int res;          // spatial side (merge: input res; breakup: input res)
  int Csrc;         // merge: source token width; concat: width of each half; breakup: Co
  const float* cond;
  int cond_stride, cond_off;
  const float* g;
  const float* b;
  float* out_f32;   // optional fp32 output (row stride C)
  bf16* out_bf16;   // optional bf16 output (row stride ldo, zero padded)
  int ldo;
};

template <int LANES, int V, int MODE>
__global__ __launch_bounds__(256) void k_row(RowArgs a) {
  constexpr int RPB = 256 / LANES;
  const int row = blockIdx.x * RPB + (int)threadIdx.x / LANES;
  const int g = threadIdx.x % LANES;
  if (row >= a.rows) return;  // whole lane group leaves together
  const int nch = a.C >> 2;

  // ---- source addressing ----
  float4 v[V];
  int src_tok = row;  // MERGE recomputes per chunk
  int b_img = 0, Y = 0, X = 0;
  if (MODE == ROW_MERGE_LN) {
    const int r2 = a.res >> 1;
    b_img = row / (r2 * r2);
    const int rem = row - b_img * r2 * r2;
    Y = rem / r2;
    X = rem - Y * r2;
  }
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = g + i * LANES;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nch) {
      const float* p;
      if (MODE == ROW_MERGE_LN) {
        const int cs = a.Csrc >> 2;
        const int k = c / cs, cc = c - k * cs;
        const int dy = k & 1, dx = k >> 1;  // (0,0),(1,0),(0,1),(1,1): swin_gnn.py:325-329
        src_tok = b_img * a.res * a.res + (2 * Y + dy) * a.res + (2 * X + dx);
        p = a.x + (size_t)src_tok * a.Csrc + 4 * cc;
      } else if (MODE == ROW_CONCAT) {
        const int cs = a.Csrc >> 2;
        p = (c < cs) ? a.x + (size_t)row * a.Csrc + 4 * c : a.x2 + (size_t)row * a.Csrc + 4 * (c - cs);
      } else {
        p = a.x + (size_t)row * a.C + 4 * c;  // BREAKUP_LN: input viewed as [4T, Co] contiguous rows
      }
      v[i] = *reinterpret_cast<const float4*>(p);
    }
  }

  // ---- noise conditioning: x <- silu(shift + x*(scale+1)), swin_gnn.py:238-240 ----
  if (MODE == ROW_COND_LN && a.cond != nullptr) {
    const float* crow = a.cond + (size_t)(row / a.L) * a.cond_stride + a.cond_off;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = g + i * LANES;
      if (c < nch) {
        const float4 sc = *reinterpret_cast<const float4*>(crow + 4 * c);
        const float4 sh = *reinterpret_cast<const float4*>(crow + a.C + 4 * c);
        v[i].x = silu_f(sh.x + v[i].x * (sc.x + 1.f));
        v[i].y = silu_f(sh.y + v[i].y * (sc.y + 1.f));
        v[i].z = silu_f(sh.z + v[i].z * (sc.z + 1.f));
        v[i].w = silu_f(sh.w + v[i].w * (sc.w + 1.f));
        if (a.out_f32) *reinterpret_cast<float4*>(a.out_f32 + (size_t)row * a.C + 4 * c) = v[i];
      }
    }
  }

  // ---- LayerNorm statistics (biased variance, eps 1e-5) ----
  if (MODE != ROW_CONCAT) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = group_sum<LANES>(s) / (float)a.C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = g + i * LANES;
      if (c < nch) {
        const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
        q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
      }
    }
    const float rstd = ln_rstd(group_sum<LANES>(q) / (float)a.C);
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = g + i * LANES;
      if (c < nch) {
        const float4 gg = *reinterpret_cast<const float4*>(a.g + 4 * c);
        const float4 bb = *reinterpret_cast<const float4*>(a.b + 4 * c);
        v[i].x = (v[i].x - mean) * rstd * gg.x + bb.x;
        v[i].y = (v[i].y - mean) * rstd * gg.y + bb.y;
        v[i].z = (v[i].z - mean) * rstd * gg.z + bb.z;
        v[i].w = (v[i].w - mean) * rstd * gg.w + bb.w;
      }
    }
  }

  // ---- destination ----
  int drow = row;
  if (MODE == ROW_BREAKUP_LN) {
    // input row = 4*t + k, t = (b, y, x) on the res grid; chunk k goes to (2y+dy, 2x+dx): swin_gnn.py:389-397
    const int t = row >> 2, k = row & 3;
    const int bi = t / (a.res * a.res);
    const int rem = t - bi * a.res * a.res;
    const int y = rem / a.res, x = rem - y * a.res;
    const int dy = k & 1, dx = k >> 1;
    drow = bi * 4 * a.res * a.res + (2 * y + dy) * 2 * a.res + (2 * x + dx);
  }
  if (MODE == ROW_LN_F32) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = g + i * LANES;
      if (c < nch) *reinterpret_cast<float4*>(a.out_f32 + (size_t)drow * a.C + 4 * c) = v[i];
    }
  } else {
    const int nst = a.ldo >> 2;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = g + i * LANES;
      if (c < nst)  // chunks in [nch, nst) are the zero K-padding of the following GEMM
        *reinterpret_cast<uint2*>(a.out_bf16 + (size_t)drow * a.ldo + 4 * c) = pack_bf16x4(v[i].x, v[i].y, v[i].z, v[i].w);
    }
  }
}

template <int MODE>
static int launch_row(const RowArgs& a, int chunks, hipStream_t s) {
  if (a.rows <= 0) return 0;
  if (a.C % 4) return SWG_E_SHAPE;
  RowGeom geo = row_geom(chunks);
  static const int kid_of[5] = {KID_ROW_COND_LN, KID_ROW_MERGE_LN, KID_ROW_CONCAT, KID_ROW_LN_F32, KID_ROW_BREAKUP_LN};
  const double el = (double)a.rows * a.C;
  double bytes = 4.0 * el + (a.out_bf16 ? 2.0 * el : 0.0) + ((a.out_f32 && MODE != ROW_COND_LN) ? 4.0 * el : 0.0);
  if (MODE == ROW_COND_LN && a.out_f32 && a.cond) bytes += 4.0 * el;
  SwgProfScope ps(kid_of[MODE], 0.0, bytes, s);
#define CALL(LN, VV)                                                                            \
  hipLaunchKernelGGL((k_row<LN, VV, MODE>), dim3(cdiv(a.rows, 256 / LN)), dim3(256), 0, s, a)
  SWG_ROW_DISPATCH(geo, CALL);
#undef CALL
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_cond_ln(const float* x, int rows, int C, int L, const float* cond, int cond_stride, int cond_off,
                   const float* g, const float* b, float* x1, bf16* xn, int ldn, hipStream_t s) {
  RowArgs a = {};
  a.x = x; a.rows = rows; a.C = C; a.L = L; a.cond = cond; a.cond_stride = cond_stride; a.cond_off = cond_off;
  a.g = g; a.b = b; a.out_f32 = x1; a.out_bf16 = xn; a.ldo = ldn;
  return launch_row<ROW_COND_LN>(a, ldn / 4, s);
}

int launch_merge_ln(const float* x, int B, int res, int C, const float* g, const float* b, bf16* xn, int ldn,
                    hipStream_t s) {
  RowArgs a = {};
  a.x = x; a.rows = B * (res / 2) * (res / 2); a.C = 4 * C; a.res = res; a.Csrc = C; a.g = g; a.b = b;
  a.out_bf16 = xn; a.ldo = ldn;
  return launch_row<ROW_MERGE_LN>(a, ldn / 4, s);
}

int launch_concat_cast(const float* xa, const float* xb, int rows, int Ch, bf16* out, int ldo, hipStream_t s) {
  RowArgs a = {};
  a.x = xa; a.x2 = xb; a.rows = rows; a.C = 2 * Ch; a.Csrc = Ch; a.out_bf16 = out; a.ldo = ldo;
  return launch_row<ROW_CONCAT>(a, ldo / 4, s);
}

int launch_ln_f32(const float* x, int rows, int C, const float* g, const float* b, float* y, hipStream_t s) {
  RowArgs a = {};
  a.x = x; a.rows = rows; a.C = C; a.g = g; a.b = b; a.out_f32 = y;
  return launch_row<ROW_LN_F32>(a, C / 4, s);
}

int launch_breakup_ln(const float* y, int B, int res, int Co, const float* g, const float* b, bf16* out, int ldo,
                      hipStream_t s) {
  RowArgs a = {};
  a.x = y; a.rows = 4 * B * res * res; a.C = Co; a.res = res; a.g = g; a.b = b; a.out_bf16 = out; a.ldo = ldo;
  return launch_row<ROW_BREAKUP_LN>(a, ldo / 4, s);
}

// ------------------------------------------------------------------------------------------------
// PatchBreakup middle (swin_gnn.py:385-399) in one pass: y row [Cin] -> LayerNorm(Cin) -> split into 4 sub-vectors of
// Co = Cin/4 -> scatter to the 2x2 sub-tokens -> post_norm LayerNorm(Co) -> bf16 rows for the post_linear GEMM.
// LANES lanes own one source row; lane g serves quarter g / (LANES/4), so the row statistics are a LANES-wide and
// the sub-vector statistics a LANES/4-wide shuffle reduction, and y is read once (no in-place LN pass).
// ------------------------------------------------------------------------------------------------
struct Breakup2Args {
  const float* y; const float* g1; const float* b1; const float* g2; const float* b2;
  bf16* out;
  int rows, res, Cin, Co, ldo;
  float* z;  // optional: the scattered rows BEFORE the second LayerNorm, fp32 [4 rows, Co] (the backward pass needs them)
};

template <int LANES, int V>
__global__ __launch_bounds__(256) void k_breakup_ln2(Breakup2Args a) {
  constexpr int RPB = 256 / LANES, Q = LANES / 4;
  const int row = blockIdx.x * RPB + (int)threadIdx.x / LANES;
  const int g = threadIdx.x % LANES;
  if (row >= a.rows) return;
  const int k = g / Q, jq = g % Q;  // quarter, lane inside the quarter
  const int cq = a.Co >> 2;         // float4 chunks per quarter
  const float* src = a.y + (size_t)row * a.Cin + k * a.Co;
  float4 v[V];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int j = jq + i * Q;
    v[i] = (j < cq) ? *reinterpret_cast<const float4*>(src + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = group_sum<LANES>(s) / (float)a.Cin;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i)
    if (jq + i * Q < cq) {
      const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  const float rstd = ln_rstd(group_sum<LANES>(q) / (float)a.Cin);
  float s2 = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int j = jq + i * Q;
    if (j < cq) {
      const float4 gg = *reinterpret_cast<const float4*>(a.g1 + k * a.Co + 4 * j);
      const float4 bb = *reinterpret_cast<const float4*>(a.b1 + k * a.Co + 4 * j);
      v[i].x = (v[i].x - mean) * rstd * gg.x + bb.x;
      v[i].y = (v[i].y - mean) * rstd * gg.y + bb.y;
      v[i].z = (v[i].z - mean) * rstd * gg.z + bb.z;
      v[i].w = (v[i].w - mean) * rstd * gg.w + bb.w;
      s2 += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean2 = group_sum<Q>(s2) / (float)a.Co;
  float q2 = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i)
    if (jq + i * Q < cq) {
      const float dx = v[i].x - mean2, dy = v[i].y - mean2, dz = v[i].z - mean2, dw = v[i].w - mean2;
      q2 += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  const float rstd2 = ln_rstd(group_sum<Q>(q2) / (float)a.Co);
  // source token t = (b, y, x) on the res grid; quarter k goes to (2y+dy, 2x+dx): swin_gnn.py:389-397
  const int bi = row / (a.res * a.res);
  const int rem = row - bi * a.res * a.res;
  const int yy = rem / a.res, xx = rem - yy * a.res;
  const int dy = k & 1, dx = k >> 1;
  const size_t drow = (size_t)bi * 4 * a.res * a.res + (size_t)(2 * yy + dy) * 2 * a.res + (2 * xx + dx);
  const int nst = a.ldo >> 2;
  if (a.z) {
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int j = jq + i * Q;
      if (j < cq) *reinterpret_cast<float4*>(a.z + drow * a.Co + 4 * j) = v[i];
    }
  }
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int j = jq + i * Q;
    if (j < nst) {
      uint2 o = make_uint2(0u, 0u);  // chunks in [cq, nst) are the zero K-padding of the following GEMM
      if (j < cq) {
        const float4 gg = *reinterpret_cast<const float4*>(a.g2 + 4 * j);
        const float4 bb = *reinterpret_cast<const float4*>(a.b2 + 4 * j);
        o = pack_bf16x4((v[i].x - mean2) * rstd2 * gg.x + bb.x, (v[i].y - mean2) * rstd2 * gg.y + bb.y,
                        (v[i].z - mean2) * rstd2 * gg.z + bb.z, (v[i].w - mean2) * rstd2 * gg.w + bb.w);
      }
      *reinterpret_cast<uint2*>(a.out + drow * a.ldo + 4 * j) = o;
    }
  }
}

int launch_breakup_ln2(const float* y, int B, int res, int Cin, const float* g1, const float* b1, const float* g2,
                       const float* b2, bf16* out, int ldo, hipStream_t s, float* z_out) {
  if (Cin % 16 || ldo % 4) return SWG_E_SHAPE;
  Breakup2Args a = {y, g1, b1, g2, b2, out, B * res * res, res, Cin, Cin / 4, ldo, z_out};
  if (a.rows <= 0) return 0;
  const int cq = ldo / 4;  // chunks to write per quarter
  int lanes = 16;
  while (lanes < 64 && cdiv(cq, lanes / 4) > 3) lanes *= 2;
  const int v = cdiv(cq, lanes / 4);
  SwgProfScope ps(KID_ROW_BREAKUP_LN, 0.0, 4.0 * a.rows * (double)Cin + 2.0 * a.rows * 4.0 * ldo, s);
#define CALL(LN, VV) hipLaunchKernelGGL((k_breakup_ln2<LN, VV>), dim3(cdiv(a.rows, 256 / LN)), dim3(256), 0, s, a)
  if (lanes == 16 && v <= 3) { if (v == 1) CALL(16, 1); else if (v == 2) CALL(16, 2); else CALL(16, 3); }
  else if (lanes == 32 && v <= 3) { if (v == 1) CALL(32, 1); else if (v == 2) CALL(32, 2); else CALL(32, 3); }
  else if (v <= 3) { if (v == 1) CALL(64, 1); else if (v == 2) CALL(64, 2); else CALL(64, 3); }
  else if (v <= 6) CALL(64, 6);
  else if (v <= 12) CALL(64, 12);
  else return SWG_E_UNSUPPORTED;
#undef CALL
  SWG_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Noise embedding (swin_gnn.py:507-513, 757-760) and the stacked affine layers: three small GEMVs.
// One wavefront per output feature; the weight row stays in registers while the batch is swept.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gemv_rows(const float* __restrict__ W, const float* __restrict__ bias,
                                                   const float* __restrict__ x, int Bx, int N, int K, int act,
                                                   float* __restrict__ out) {
  const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (n >= N) return;
  const int kch = K >> 2;
  float4 w0 = make_float4(0.f, 0.f, 0.f, 0.f), w1 = w0;
  if (lane < kch) w0 = *reinterpret_cast<const float4*>(W + (size_t)n * K + 4 * lane);
  if (lane + 64 < kch) w1 = *reinterpret_cast<const float4*>(W + (size_t)n * K + 4 * (lane + 64));
  const float bn = bias ? bias[n] : 0.f;
  for (int b = 0; b < Bx; ++b) {
    float acc = 0.f;
    if (lane < kch) {
      const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)b * K + 4 * lane);
      acc += (w0.x * xv.x + w0.y * xv.y) + (w0.z * xv.z + w0.w * xv.w);
    }
    if (lane + 64 < kch) {
      const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)b * K + 4 * (lane + 64));
      acc += (w1.x * xv.x + w1.y * xv.y) + (w1.z * xv.z + w1.w * xv.w);
    }
    acc = group_sum<64>(acc) + bn;
    if (act == 1) acc = silu_f(acc);
    if (lane == 0) out[(size_t)b * N + n] = acc;
  }
}

// The same product for a whole training batch (per-sample sigmas: Bx = B rows): fp32 LDS-tiled GEMM, 64 (rows) x 64
// (features) per workgroup, BK = 16, 4 x 4 outputs per thread.  Exact fp32 like the GEMV (the conditioning vectors
// scale every block's residual stream, so they do not go through bf16).
__global__ __launch_bounds__(256) void k_linear_f32_tiled(const float* __restrict__ W, const float* __restrict__ bias,
                                                          const float* __restrict__ x, int Bx, int N, int K, int act,
                                                          float* __restrict__ out) {
  __shared__ __attribute__((aligned(16))) float sX[16][68];
  __shared__ __attribute__((aligned(16))) float sW[16][68];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int n0 = blockIdx.x * 64, b0 = blockIdx.y * 64;
  const int lrow = tid >> 2, lk = (tid & 3) * 4;
  const float* xr = x + (size_t)min(b0 + lrow, Bx - 1) * K + lk;
  const float* wr = W + (size_t)min(n0 + lrow, N - 1) * K + lk;
  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;
  for (int k0 = 0; k0 < K; k0 += 16) {
    const float4 xv = *reinterpret_cast<const float4*>(xr + k0);
    const float4 wv = *reinterpret_cast<const float4*>(wr + k0);
    sX[lk + 0][lrow] = xv.x; sX[lk + 1][lrow] = xv.y; sX[lk + 2][lrow] = xv.z; sX[lk + 3][lrow] = xv.w;
    sW[lk + 0][lrow] = wv.x; sW[lk + 1][lrow] = wv.y; sW[lk + 2][lrow] = wv.z; sW[lk + 3][lrow] = wv.w;
    __syncthreads();
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) {
      const float4 a = *reinterpret_cast<const float4*>(&sX[kk][4 * ty]);
      const float4 w = *reinterpret_cast<const float4*>(&sW[kk][4 * tx]);
      const float av[4] = {a.x, a.y, a.z, a.w}, wv4[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(av[i], wv4[j], acc[i][j]);
    }
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int b = b0 + 4 * ty + i;
    if (b >= Bx) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + 4 * tx + j;
      if (n >= N) continue;
      float v = acc[i][j] + (bias ? bias[n] : 0.f);
      if (act == 1) v = silu_f(v);
      out[(size_t)b * N + n] = v;
    }
  }
}

int launch_gemv_rows(const float* W, const float* bias, const float* x, int Bx, int N, int K, int act, float* out,
                     hipStream_t s) {
  if (K % 4 || K > 512) return SWG_E_UNSUPPORTED;
  SwgProfScope ps(KID_GEMV, 2.0 * Bx * (double)N * K, 4.0 * N * (double)K + 4.0 * Bx * (K + N), s);
  if (Bx >= 16 && K % 16 == 0) {
    hipLaunchKernelGGL(k_linear_f32_tiled, dim3(cdiv(N, 64), cdiv(Bx, 64)), dim3(256), 0, s, W, bias, x, Bx, N, K, act, out);
    SWG_LAUNCH_CHECK();
    return 0;
  }
  hipLaunchKernelGGL(k_gemv_rows, dim3(cdiv(N, 4)), dim3(256), 0, s, W, bias, x, Bx, N, K, act, out);
  SWG_LAUNCH_CHECK();
  return 0;
}

__global__ void k_pos_embed(const float* __restrict__ labels, int Bc, const float* __restrict__ freqs, int nf,
                            float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Bc * nf) return;
  const int b = i / nf, k = i - b * nf;
  const float ang = labels[b] * freqs[k];
  out[(size_t)b * 2 * nf + k] = cosf(ang);
  out[(size_t)b * 2 * nf + nf + k] = sinf(ang);
}

int launch_pos_embed(const float* labels, int Bc, const float* freqs, int nf, float* out, hipStream_t s) {
  SwgProfScope ps(KID_POS_EMBED, 0.0, 8.0 * Bc * nf, s);
  hipLaunchKernelGGL(k_pos_embed, dim3(cdiv(Bc * nf, 256)), dim3(256), 0, s, labels, Bc, freqs, nf, out);
  SWG_LAUNCH_CHECK();
  return 0;
}

// EDM preconditioning coefficients (runner/objectives/edm.py:122-126), sigma_data = 0.5.
__global__ void k_edm_coef(const float* __restrict__ labels, int B, int uniform, int mode, float* __restrict__ coef,
                           float* __restrict__ c_noise) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float s = labels[uniform ? 0 : b];
  float c_skip = 0.f, c_out = 1.f, c_in = 1.f, c_n = s;
  if (mode == 1) {
    const float d = s * s + 0.25f;
    c_skip = 0.25f / d;
    c_out = s * 0.5f / sqrtf(d);
    c_in = 1.0f / sqrtf(d);
    c_n = logf(s) * 0.25f;
  }
  coef[4 * b + 0] = c_skip;
  coef[4 * b + 1] = c_out;
  coef[4 * b + 2] = c_in;
  coef[4 * b + 3] = c_n;
  if (!uniform || b == 0) c_noise[b] = c_n;
}

int launch_edm_coef(const float* labels, int B, int uniform, int mode, float* coef, float* c_noise, hipStream_t s) {
  SwgProfScope ps(KID_EDM_COEF, 0.0, 24.0 * B, s);
  hipLaunchKernelGGL(k_edm_coef, dim3(cdiv(B, 64)), dim3(64), 0, s, labels, B, uniform, mode, coef, c_noise);
  SWG_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// PatchEmbed (swin_gnn.py:562-577) with the self-conditioning concat (:769-771) and the c_in scaling
// of Precond (precond.py:47) folded into the gather: conv(k=s=p) -> LN -> silu(shift + y(1+scale)).
// ------------------------------------------------------------------------------------------------
struct EmbedArgs {
  const float* x; const float* sc; const float* c_in; int c_in_stride;
  const float* cond; int cond_stride, cond_off;
  const float* w; const float* bias; const float* g; const float* b;
  float* tok;
  int B, N, p, in_ch, C, self_cond;
};

template <int LANES, int V>
__global__ __launch_bounds__(256) void k_patch_embed(EmbedArgs a) {
  // LDS: conv weight transposed to [kin][C] (a lane reads its 4 channels as one float4 per tap) and the
  // RPB x kin input taps of the block's tokens (broadcast reads inside a lane group).
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RPB = 256 / LANES;
  const int hp = a.N / a.p;
  const int L = hp * hp;
  const int pp = a.p * a.p;
  const int kin = a.in_ch * pp;
  float* sW = smem;                 // [kin][C]
  float* sIn = smem + kin * a.C;    // [RPB][kin]
  const int rows = a.B * L;
  const int row0 = blockIdx.x * RPB;
  for (int i = threadIdx.x; i < kin * a.C; i += 256) {
    const int c = i / kin, k = i - c * kin;  // global layout [C][kin]
    sW[k * a.C + c] = a.w[i];
  }
  for (int i = threadIdx.x; i < RPB * kin; i += 256) {
    const int t = i / kin, k = i - t * kin;
    const int row = min(row0 + t, rows - 1);
    const int bi = row / L;
    const int rem = row - bi * L;
    const int Y = rem / hp, X = rem - Y * hp;
    const int ci = k / pp, ij = k - ci * pp;
    const int ii = ij / a.p, jj = ij - ii * a.p;
    const size_t off = ((size_t)bi * a.N + (Y * a.p + ii)) * a.N + (X * a.p + jj);
    const bool is_x = (!a.self_cond) || ci == a.in_ch - 1;  // cat([self_cond, x]) : last channel is x
    float v = 0.f;
    if (is_x) v = (a.c_in ? a.c_in[(size_t)bi * a.c_in_stride] : 1.0f) * a.x[off];
    else if (a.sc) v = a.sc[off];
    sIn[i] = v;
  }
  __syncthreads();
  const int tl = (int)threadIdx.x / LANES;
  const int row = row0 + tl;
  const int g = threadIdx.x % LANES;
  if (row >= rows) return;
  const int bi = row / L;
  const int nch = a.C >> 2;

  float4 v[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = g + i * LANES;
    v[i] = (c < nch) ? *reinterpret_cast<const float4*>(a.bias + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float* in = sIn + tl * kin;
  for (int k = 0; k < kin; ++k) {
    const float xin = in[k];
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = g + i * LANES;
      if (c < nch) {
        const float4 w = *reinterpret_cast<const float4*>(sW + k * a.C + 4 * c);
        v[i].x = fmaf(xin, w.x, v[i].x); v[i].y = fmaf(xin, w.y, v[i].y);
        v[i].z = fmaf(xin, w.z, v[i].z); v[i].w = fmaf(xin, w.w, v[i].w);
      }
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  const float mean = group_sum<LANES>(s) / (float)a.C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i)
    if (g + i * LANES < nch) {
      const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  const float rstd = ln_rstd(group_sum<LANES>(q) / (float)a.C);
  const float* crow = a.cond + (size_t)bi * a.cond_stride + a.cond_off;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = g + i * LANES;
    if (c < nch) {
      const float4 gg = *reinterpret_cast<const float4*>(a.g + 4 * c);
      const float4 bb = *reinterpret_cast<const float4*>(a.b + 4 * c);
      const float4 sc = *reinterpret_cast<const float4*>(crow + 4 * c);
      const float4 sh = *reinterpret_cast<const float4*>(crow + a.C + 4 * c);
      float4 y;
      y.x = silu_f(sh.x + ((v[i].x - mean) * rstd * gg.x + bb.x) * (sc.x + 1.f));
      y.y = silu_f(sh.y + ((v[i].y - mean) * rstd * gg.y + bb.y) * (sc.y + 1.f));
      y.z = silu_f(sh.z + ((v[i].z - mean) * rstd * gg.z + bb.z) * (sc.z + 1.f));
      y.w = silu_f(sh.w + ((v[i].w - mean) * rstd * gg.w + bb.w) * (sc.w + 1.f));
      *reinterpret_cast<float4*>(a.tok + (size_t)row * a.C + 4 * c) = y;
    }
  }
}

int launch_patch_embed(const swg_net* net, int B, const float* x, const float* sc, const float* c_in, int c_in_stride,
                       const float* cond, int cond_stride, float* tok, hipStream_t s) {
  if (embed_mfma_supported(net))
    return launch_patch_embed_mfma(net, B, x, sc, c_in, c_in_stride, cond, cond_stride, tok, s);
  const int kin = net->in_ch * net->p * net->p;
  if (kin > 32 || net->N % net->p || net->C0 % 4) return SWG_E_UNSUPPORTED;
  if (net->self_cond ? net->in_ch != 2 : net->in_ch != 1) return SWG_E_UNSUPPORTED;
  EmbedArgs a = {x, sc, c_in, c_in_stride, cond, cond_stride, net->pe_cond_off, net->pe_w, net->pe_b,
                 net->pe_ln.g, net->pe_ln.b, tok, B, net->N, net->p, net->in_ch, net->C0, net->self_cond};
  const int hp = net->N / net->p;
  const int rows = B * hp * hp;
  RowGeom geo = row_geom(net->C0 / 4);
  SwgProfScope ps(KID_PATCH_EMBED, 2.0 * rows * (double)net->C0 * kin,
                  4.0 * B * (double)net->N * net->N * (sc ? 2 : 1) + 4.0 * rows * (double)net->C0, s);
#define CALL(LN, VV)                                                                                   \
  hipLaunchKernelGGL((k_patch_embed<LN, VV>), dim3(cdiv(rows, 256 / LN)), dim3(256),                   \
                     (size_t)(kin * net->C0 + (256 / LN) * kin) * sizeof(float), s, a)
  SWG_ROW_DISPATCH(geo, CALL);
#undef CALL
  SWG_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// Final LayerNorm (swin_gnn.py:748) + collapsed read-out (:697-701): per token C -> p*p values.
// ------------------------------------------------------------------------------------------------
struct ReadoutArgs {
  const float* tok; const float* g; const float* b; const float* w; float bias; float* F;
  int B, N, p, C;
};

template <int LANES, int V>
__global__ __launch_bounds__(256) void k_final_readout(ReadoutArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];  // collapsed read-out weight [C][16] (pp padded to 16)
  constexpr int RPB = 256 / LANES;
  const int hp = a.N / a.p;
  const int L = hp * hp;
  const int pp = a.p * a.p;
  for (int i = threadIdx.x; i < a.C * 16; i += 256) {
    const int c = i >> 4, o = i & 15;
    smem[i] = (o < pp) ? a.w[c * pp + o] : 0.f;
  }
  __syncthreads();
  const int row = blockIdx.x * RPB + (int)threadIdx.x / LANES;
  const int g = threadIdx.x % LANES;
  if (row >= a.B * L) return;
  const int nch = a.C >> 2;
  float4 v[V];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = g + i * LANES;
    v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < nch) v[i] = *reinterpret_cast<const float4*>(a.tok + (size_t)row * a.C + 4 * c);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = group_sum<LANES>(s) / (float)a.C;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i)
    if (g + i * LANES < nch) {
      const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
      q += (dx * dx + dy * dy) + (dz * dz + dw * dw);
    }
  const float rstd = ln_rstd(group_sum<LANES>(q) / (float)a.C);
  float o[16];
#pragma unroll
  for (int k = 0; k < 16; ++k) o[k] = 0.f;
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = g + i * LANES;
    if (c < nch) {
      const float4 gg = *reinterpret_cast<const float4*>(a.g + 4 * c);
      const float4 bb = *reinterpret_cast<const float4*>(a.b + 4 * c);
      const float y[4] = {(v[i].x - mean) * rstd * gg.x + bb.x, (v[i].y - mean) * rstd * gg.y + bb.y,
                          (v[i].z - mean) * rstd * gg.z + bb.z, (v[i].w - mean) * rstd * gg.w + bb.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float4* wr = reinterpret_cast<const float4*>(smem + (4 * c + e) * 16);
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
          const float4 w = wr[k4];
          o[4 * k4 + 0] = fmaf(y[e], w.x, o[4 * k4 + 0]); o[4 * k4 + 1] = fmaf(y[e], w.y, o[4 * k4 + 1]);
          o[4 * k4 + 2] = fmaf(y[e], w.z, o[4 * k4 + 2]); o[4 * k4 + 3] = fmaf(y[e], w.w, o[4 * k4 + 3]);
        }
      }
    }
  }
  const int bi = row / L;
  const int rem = row - bi * L;
  const int Y = rem / hp, X = rem - Y * hp;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const float acc = group_sum<LANES>(o[k]) + a.bias;
    if (k < pp && g == (k % LANES)) {
      const int i2 = k / a.p, j2 = k - i2 * a.p;
      a.F[((size_t)bi * a.N + (Y * a.p + i2)) * a.N + (X * a.p + j2)] = acc;
    }
  }
}

int launch_final_readout(const swg_net* net, int B, const float* tok, float* F, hipStream_t s) {
  if (embed_mfma_supported(net)) return launch_final_readout_mfma(net, B, tok, F, s);
  if (net->p * net->p > 16) return SWG_E_UNSUPPORTED;
  ReadoutArgs a = {tok, net->norm.g, net->norm.b, net->ro_w, net->ro_b, F, B, net->N, net->p, net->C0};
  const int hp = net->N / net->p;
  const int rows = B * hp * hp;
  RowGeom geo = row_geom(net->C0 / 4);
  SwgProfScope ps(KID_READOUT, 2.0 * rows * (double)net->C0 * net->p * net->p,
                  4.0 * rows * (double)net->C0 + 4.0 * B * (double)net->N * net->N, s);
#define CALL(LN, VV)                                                                                   \
  hipLaunchKernelGGL((k_final_readout<LN, VV>), dim3(cdiv(rows, 256 / LN)), dim3(256),                 \
                     (size_t)net->C0 * 16 * sizeof(float), s, a)
  SWG_ROW_DISPATCH(geo, CALL);
#undef CALL
  SWG_LAUNCH_CHECK();
  return 0;
}

// ------------------------------------------------------------------------------------------------
// N x N adjacency kernels, tiled 32 x 32 through LDS so both A[r][c] and A[c][r] are read coalesced.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_256(float v, float* red) {
  v = group_sum<64>(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// out = mask(c_skip*x + c_out*0.5*(Fm + Fm^T)), Fm = mask(F): swin_gnn.py:777-778, precond.py:49-51
// Ca: adjacency channels per graph (planes [B*Ca, N, N]; flags / coef are per graph) -- 1 for SwinGNN
__global__ __launch_bounds__(256) void k_sym_precond(int N, int Ca, const float* __restrict__ F,
                                                     const float* __restrict__ x, const float* __restrict__ flags,
                                                     const float* __restrict__ coef, float* __restrict__ out,
                                                     float* __restrict__ sym_err) {
  __shared__ float tF[32][33];
  __shared__ float tX[32][33];
  __shared__ float red[4];
  const int bi = blockIdx.z / Ca;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t base = (size_t)blockIdx.z * N * N;
  const float c_skip = coef ? coef[4 * bi + 0] : 0.f;
  const float c_out = coef ? coef[4 * bi + 1] : 1.f;
  // transposed tile: rows c0.., cols r0..
  for (int k = ty; k < 32; k += 8) {
    const int rr = c0 + k, cc = r0 + tx;
    const bool ok = rr < N && cc < N;
    tF[k][tx] = ok ? F[base + (size_t)rr * N + cc] : 0.f;
    tX[k][tx] = (ok && sym_err && x) ? x[base + (size_t)rr * N + cc] : 0.f;
  }
  __syncthreads();
  float err = 0.f;
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    if (r < N && c < N) {
      const bool m = flags[(size_t)bi * N + r] != 0.f && flags[(size_t)bi * N + c] != 0.f;
      float v = 0.f;
      if (m) {
        const float sym = 0.5f * (F[base + (size_t)r * N + c] + tF[tx][k]);
        const float xv = x ? x[base + (size_t)r * N + c] : 0.f;
        // explicit fma on both sides: D[r][c] and D[c][r] must round identically for the check to be exact
        const float cs = __fmul_rn(c_out, sym);
        v = __fmaf_rn(c_skip, xv, cs);
        if (sym_err) err += fabsf(v - __fmaf_rn(c_skip, tX[tx][k], cs));
      }
      out[base + (size_t)r * N + c] = v;
    }
  }
  if (sym_err) {
    err = block_sum_256(err, red);
    if (threadIdx.x == 0 && err != 0.f) atomicAdd(sym_err, err);
  }
}

// Same result for N % 4 == 0, one workgroup per PAIR of mirrored 32x32 tiles (i, j), (j, i), i <= j: both tiles of F and
// x are read once (float4 rows) and both output tiles are written, instead of every tile fetching its mirror again
// (20 -> 12 B per entry; the single-tile kernel ran at 1.55x its algorithmic traffic).  The symmetry error of a
// mirrored pair is the same number seen from both sides, so an off-diagonal pair adds it twice.
__global__ __launch_bounds__(256) void k_sym_precond_pair(int N, int Ca, int nt, const float* __restrict__ F,
                                                          const float* __restrict__ x, const float* __restrict__ flags,
                                                          const float* __restrict__ coef, float* __restrict__ out,
                                                          float* __restrict__ sym_err) {
  __shared__ float tF[2][32][33];
  __shared__ float tX[2][32][33];
  __shared__ float red[4];
  int ti = 0, p = blockIdx.x;
  while (p >= nt - ti) { p -= nt - ti; ++ti; }  // pair index -> (ti, tj = ti + p), row-major over the upper triangle
  const int tj = ti + p;
  const int bi = blockIdx.z / Ca;
  const size_t base = (size_t)blockIdx.z * N * N;
  const float c_skip = coef ? coef[4 * bi + 0] : 0.f;
  const float c_out = coef ? coef[4 * bi + 1] : 1.f;
  const int row = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
  const int o0[2][2] = {{ti * 32, tj * 32}, {tj * 32, ti * 32}};  // (first row, first column) of tile a / tile b
  const int ntile = (ti == tj) ? 1 : 2;
  for (int t = 0; t < ntile; ++t) {
    const int r = o0[t][0] + row, c = o0[t][1] + c4;
    float4 f = make_float4(0.f, 0.f, 0.f, 0.f), xv = f;
    if (r < N && c < N) {
      f = *reinterpret_cast<const float4*>(F + base + (size_t)r * N + c);
      if (x) xv = *reinterpret_cast<const float4*>(x + base + (size_t)r * N + c);
    }
    tF[t][row][c4] = f.x; tF[t][row][c4 + 1] = f.y; tF[t][row][c4 + 2] = f.z; tF[t][row][c4 + 3] = f.w;
    tX[t][row][c4] = xv.x; tX[t][row][c4 + 1] = xv.y; tX[t][row][c4 + 2] = xv.z; tX[t][row][c4 + 3] = xv.w;
  }
  __syncthreads();
  float err = 0.f;
  for (int t = 0; t < ntile; ++t) {
    const int u = (ntile == 1) ? 0 : 1 - t;  // the mirrored tile
    const int r = o0[t][0] + row, c = o0[t][1] + c4;
    if (r < N && c < N) {
      const bool mr = flags[(size_t)bi * N + r] != 0.f;
      float v[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = 0.f;
        if (mr && flags[(size_t)bi * N + c + e] != 0.f) {
          const float sym = 0.5f * (tF[t][row][c4 + e] + tF[u][c4 + e][row]);
          // explicit fma on both sides: D[r][c] and D[c][r] must round identically for the check to be exact
          const float cs = __fmul_rn(c_out, sym);
          v[e] = __fmaf_rn(c_skip, tX[t][row][c4 + e], cs);
          if (sym_err && x && t == 0) err += fabsf(v[e] - __fmaf_rn(c_skip, tX[u][c4 + e][row], cs));
        }
      }
      *reinterpret_cast<float4*>(out + base + (size_t)r * N + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
  }
  if (sym_err) {
    err = block_sum_256(ntile == 2 ? 2.f * err : err, red);
    if (threadIdx.x == 0 && err != 0.f) atomicAdd(sym_err, err);
  }
}

int launch_sym_precond_ch(int B, int Ca, int N, const float* F, const float* x, const float* flags, const float* coef,
                          float* out, float* sym_err, hipStream_t s) {
  SwgProfScope ps(KID_SYM_PRECOND, 0.0, 4.0 * B * Ca * (double)N * N * (x ? 3 : 2), s);
  const int nt = cdiv(N, 32);
  // the mirrored-tile form reads / writes float4: every operand must be 16-byte aligned (public entry points hand in
  // arbitrary float* views)
  if (N % 4 == 0 && aligned16(F, x, flags, out)) {
    hipLaunchKernelGGL(k_sym_precond_pair, dim3(nt * (nt + 1) / 2, 1, B * Ca), dim3(256), 0, s, N, Ca, nt, F, x, flags,
                       coef, out, sym_err);
  } else {
    hipLaunchKernelGGL(k_sym_precond, dim3(nt, nt, B * Ca), dim3(256), 0, s, N, Ca, F, x, flags, coef, out, sym_err);
  }
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_sym_precond(int B, int N, const float* F, const float* x, const float* flags, const float* coef, float* out,
                       float* sym_err, hipStream_t s) {
  return launch_sym_precond_ch(B, 1, N, F, x, flags, coef, out, sym_err, s);
}

// x_hat = mask(x + coef * sym(raw)); sym(raw)[r][c] = raw[min][max] off the diagonal, 0 on it.
__global__ __launch_bounds__(256) void k_heun_churn(int N, const float* __restrict__ x, const float* __restrict__ raw,
                                                    const float* __restrict__ flags, float coef,
                                                    float* __restrict__ xh) {
  __shared__ float tR[32][33];
  const int bi = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t base = (size_t)bi * N * N;
  for (int k = ty; k < 32; k += 8) {
    const int rr = c0 + k, cc = r0 + tx;
    tR[k][tx] = (rr < N && cc < N) ? raw[base + (size_t)rr * N + cc] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    if (r < N && c < N) {
      const bool m = flags[(size_t)bi * N + r] != 0.f && flags[(size_t)bi * N + c] != 0.f;
      float v = 0.f;
      if (m) {
        const float e = (r < c) ? raw[base + (size_t)r * N + c] : ((r > c) ? tR[tx][k] : 0.f);
        v = x[base + (size_t)r * N + c] + coef * e;
      }
      xh[base + (size_t)r * N + c] = v;
    }
  }
}

// d = mask((1/t) x_hat - (1/t) mask(D)); x' = x_hat + h d   (mcmc_sampler/edm.py:188-193, 197)
__global__ __launch_bounds__(256) void k_heun_euler(int N, size_t total, const float* __restrict__ xh,
                                                    const float* __restrict__ den, const float* __restrict__ flags,
                                                    float inv_t, float h, int mask_out, float* __restrict__ d_cur,
                                                    float* __restrict__ xp) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t nn = (size_t)N * N;
  const int bi = (int)(i / nn);
  const int rem = (int)(i - (size_t)bi * nn);
  const int r = rem / N, c = rem - r * N;
  const bool m = flags[(size_t)bi * N + r] != 0.f && flags[(size_t)bi * N + c] != 0.f;
  const float xv = xh[i];
  const float d = m ? (inv_t * xv - inv_t * den[i]) : 0.f;
  if (d_cur) d_cur[i] = d;
  const float o = xv + h * d;
  xp[i] = (mask_out && !m) ? 0.f : o;
}

// d' = (1/t') x' - (1/t') D2 ; x+ = mask(x_hat + h (0.5 d + 0.5 d'))   (mcmc_sampler/edm.py:211-215)
__global__ __launch_bounds__(256) void k_heun_correct(int N, size_t total, const float* __restrict__ xh,
                                                      const float* __restrict__ xp, const float* __restrict__ d_cur,
                                                      const float* __restrict__ den2, const float* __restrict__ flags,
                                                      float inv_tp, float h, float* __restrict__ xn) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t nn = (size_t)N * N;
  const int bi = (int)(i / nn);
  const int rem = (int)(i - (size_t)bi * nn);
  const int r = rem / N, c = rem - r * N;
  const bool m = flags[(size_t)bi * N + r] != 0.f && flags[(size_t)bi * N + c] != 0.f;
  float o = 0.f;
  if (m) {
    const float dp = inv_tp * xp[i] - inv_tp * den2[i];
    o = xh[i] + h * (0.5f * d_cur[i] + 0.5f * dp);
  }
  xn[i] = o;
}

// float4 forms of the two updates for N % 4 == 0 (four entries of one row per thread: one row / column decode and four
// 16-byte accesses instead of four of each; the per-entry arithmetic is the scalar kernels' own, results bit-identical)
__global__ __launch_bounds__(256) void k_heun_euler4(int N, size_t total4, const float* __restrict__ xh,
                                                     const float* __restrict__ den, const float* __restrict__ flags,
                                                     float inv_t, float h, int mask_out, float* __restrict__ d_cur,
                                                     float* __restrict__ xp) {
  const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 >= total4) return;
  const size_t i = i4 * 4, nn = (size_t)N * N;
  const int bi = (int)(i / nn);
  const int rem = (int)(i - (size_t)bi * nn);
  const int r = rem / N, c = rem - r * N;
  const bool mr = flags[(size_t)bi * N + r] != 0.f;
  const float4 fc = *reinterpret_cast<const float4*>(flags + (size_t)bi * N + c);
  const float4 xv = *reinterpret_cast<const float4*>(xh + i);
  const float4 dv = *reinterpret_cast<const float4*>(den + i);
  const float x4[4] = {xv.x, xv.y, xv.z, xv.w}, d4[4] = {dv.x, dv.y, dv.z, dv.w}, f4[4] = {fc.x, fc.y, fc.z, fc.w};
  float dd[4], oo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const bool m = mr && f4[e] != 0.f;
    dd[e] = m ? (inv_t * x4[e] - inv_t * d4[e]) : 0.f;
    const float o = x4[e] + h * dd[e];
    oo[e] = (mask_out && !m) ? 0.f : o;
  }
  if (d_cur) *reinterpret_cast<float4*>(d_cur + i) = make_float4(dd[0], dd[1], dd[2], dd[3]);
  *reinterpret_cast<float4*>(xp + i) = make_float4(oo[0], oo[1], oo[2], oo[3]);
}

__global__ __launch_bounds__(256) void k_heun_correct4(int N, size_t total4, const float* __restrict__ xh,
                                                       const float* __restrict__ xp, const float* __restrict__ d_cur,
                                                       const float* __restrict__ den2, const float* __restrict__ flags,
                                                       float inv_tp, float h, float* __restrict__ xn) {
  const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i4 >= total4) return;
  const size_t i = i4 * 4, nn = (size_t)N * N;
  const int bi = (int)(i / nn);
  const int rem = (int)(i - (size_t)bi * nn);
  const int r = rem / N, c = rem - r * N;
  const bool mr = flags[(size_t)bi * N + r] != 0.f;
  const float4 fc = *reinterpret_cast<const float4*>(flags + (size_t)bi * N + c);
  const float4 a = *reinterpret_cast<const float4*>(xh + i), b = *reinterpret_cast<const float4*>(xp + i);
  const float4 dc = *reinterpret_cast<const float4*>(d_cur + i), d2 = *reinterpret_cast<const float4*>(den2 + i);
  const float xh4[4] = {a.x, a.y, a.z, a.w}, xp4[4] = {b.x, b.y, b.z, b.w}, dc4[4] = {dc.x, dc.y, dc.z, dc.w},
              d24[4] = {d2.x, d2.y, d2.z, d2.w}, f4[4] = {fc.x, fc.y, fc.z, fc.w};
  float oo[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    oo[e] = 0.f;
    if (mr && f4[e] != 0.f) {
      const float dp = inv_tp * xp4[e] - inv_tp * d24[e];
      oo[e] = xh4[e] + h * (0.5f * dc4[e] + 0.5f * dp);
    }
  }
  *reinterpret_cast<float4*>(xn + i) = make_float4(oo[0], oo[1], oo[2], oo[3]);
}

__global__ __launch_bounds__(256) void k_mask_adjs(int N, size_t total, float* __restrict__ a,
                                                   const float* __restrict__ flags) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const size_t nn = (size_t)N * N;
  const int bi = (int)(i / nn);
  const int rem = (int)(i - (size_t)bi * nn);
  const int r = rem / N, c = rem - r * N;
  if (flags[(size_t)bi * N + r] == 0.f || flags[(size_t)bi * N + c] == 0.f) a[i] = 0.f;
}

__global__ __launch_bounds__(256) void k_sym_check(int N, const float* __restrict__ a, float* __restrict__ acc) {
  __shared__ float tA[32][33];
  __shared__ float red[4];
  const int bi = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t base = (size_t)bi * N * N;
  for (int k = ty; k < 32; k += 8) {
    const int rr = c0 + k, cc = r0 + tx;
    tA[k][tx] = (rr < N && cc < N) ? a[base + (size_t)rr * N + cc] : 0.f;
  }
  __syncthreads();
  float err = 0.f;
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    if (r < N && c < N) err += fabsf(a[base + (size_t)r * N + c] - tA[tx][k]);
  }
  err = block_sum_256(err, red);
  if (threadIdx.x == 0 && err != 0.f) atomicAdd(acc, err);
}

// ------------------------------------------------------------------------------------------------
// Training objective and loss of the EDM path (forward only): runner/objectives/edm.py:160-212,
// loss/rainbow_loss.py:57-111.
// ------------------------------------------------------------------------------------------------
// sigma = exp(P_std r + P_mean), w = (sigma^2 + sd^2) / (sigma sd)^2 (:174-176); preconditioning scalars (:122-126)
__global__ void k_edm_train_params(int B, const float* __restrict__ rnd, float* __restrict__ sigmas,
                                   float* __restrict__ weights, float* __restrict__ c_skip, float* __restrict__ c_out,
                                   float* __restrict__ c_in, float* __restrict__ c_noise) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float sd = 0.5f;
  const float sg = expf(rnd[b] * 1.2f - 1.2f);
  sigmas[b] = sg;
  const float ss = sg * sd;
  weights[b] = (sg * sg + sd * sd) / (ss * ss);
  if (c_skip) c_skip[b] = sd * sd / (sg * sg + sd * sd);
  if (c_out) c_out[b] = sg * sd / sqrtf(sg * sg + sd * sd);
  if (c_in) c_in[b] = 1.0f / sqrtf(sd * sd + sg * sg);
  if (c_noise) c_noise[b] = logf(sg) * 0.25f;
}

// noisy = mask(clean + sigma_b * sym(raw)), noise = mask(sigma_b * sym(raw))  (:181-186, graph_utils.py:122-152)
__global__ __launch_bounds__(256) void k_edm_noisy(int N, const float* __restrict__ clean, const float* __restrict__ raw,
                                                   const float* __restrict__ flags, const float* __restrict__ sigmas,
                                                   float* __restrict__ noisy, float* __restrict__ noise) {
  __shared__ float tR[32][33];
  const int bi = blockIdx.z;
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t base = (size_t)bi * N * N;
  const float sg = sigmas[bi];
  for (int k = ty; k < 32; k += 8) {
    const int rr = c0 + k, cc = r0 + tx;
    tR[k][tx] = (rr < N && cc < N) ? raw[base + (size_t)rr * N + cc] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    if (r < N && c < N) {
      const bool m = flags[(size_t)bi * N + r] != 0.f && flags[(size_t)bi * N + c] != 0.f;
      float v = 0.f, e = 0.f;
      if (m) {
        e = sg * ((r < c) ? raw[base + (size_t)r * N + c] : ((r > c) ? tR[tx][k] : 0.f));
        v = clean[base + (size_t)r * N + c] + e;
      }
      noisy[base + (size_t)r * N + c] = v;
      if (noise) noise[base + (size_t)r * N + c] = e;
    }
  }
}

// per graph: S_b = w_b * sum_{ij} mask_ij (pred - target)^2 and n_b = sum flags_b (one workgroup of 16 waves per
// graph, fixed summation order: deterministic).  VEC: N % 4 == 0 and 16-byte aligned planes -> the plane is walked as
// one flat run of float4 (four loads in flight per lane); the masks of a float4 come from the cached flag row.
template <bool VEC>
__global__ __launch_bounds__(1024) void k_edm_loss_sums(int N, const float* __restrict__ pred,
                                                        const float* __restrict__ target,
                                                        const float* __restrict__ flags,
                                                        const float* __restrict__ weights, float* __restrict__ S,
                                                        float* __restrict__ nn) {
  __shared__ float red[16];
  const int bi = blockIdx.x;
  const size_t base = (size_t)bi * N * N;
  const float* fl = flags + (size_t)bi * N;
  float acc = 0.f;
  if (VEC) {
    const float4* p4 = reinterpret_cast<const float4*>(pred + base);
    const float4* t4 = reinterpret_cast<const float4*>(target + base);
    const int n4 = N * (N / 4);
    for (int i0 = threadIdx.x; i0 < n4; i0 += 4096) {
      float4 a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 1024 * u;
        if (i < n4) { a[u] = p4[i]; b[u] = t4[i]; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + 1024 * u;
        if (i >= n4) continue;
        const int r = (4 * i) / N, c = 4 * i - r * N;
        if (fl[r] == 0.f) continue;
        const float dx = a[u].x - b[u].x, dy = a[u].y - b[u].y, dz = a[u].z - b[u].z, dw = a[u].w - b[u].w;
        if (fl[c] != 0.f) acc = fmaf(dx, dx, acc);
        if (fl[c + 1] != 0.f) acc = fmaf(dy, dy, acc);
        if (fl[c + 2] != 0.f) acc = fmaf(dz, dz, acc);
        if (fl[c + 3] != 0.f) acc = fmaf(dw, dw, acc);
      }
    }
  } else {
    for (int r = threadIdx.x >> 6; r < N; r += 16) {
      if (fl[r] == 0.f) continue;
      for (int c = threadIdx.x & 63; c < N; c += 64) {
        if (fl[c] == 0.f) continue;
        const float d = pred[base + (size_t)r * N + c] - target[base + (size_t)r * N + c];
        acc = fmaf(d, d, acc);
      }
    }
  }
  float cnt = 0.f;
  for (int r = threadIdx.x; r < N; r += 1024) cnt += fl[r];
  acc = group_sum<64>(acc);
  cnt = group_sum<64>(cnt);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  float tot = 0.f;
  if (threadIdx.x == 0)
    for (int w = 0; w < 16; ++w) tot += red[w];
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) {
    float n = 0.f;
    for (int w = 0; w < 16; ++w) n += red[w];
    S[bi] = tot * (weights ? weights[bi] : 1.0f);
    nn[bi] = n;
  }
}

static void launch_edm_loss_sums(int B, int N, const float* pred, const float* target, const float* node_flags,
                                 const float* weights, float* S, float* nn, hipStream_t s) {
  const bool vec = N % 4 == 0 && aligned16(pred, target);
  if (vec) hipLaunchKernelGGL(k_edm_loss_sums<true>, dim3(B), dim3(1024), 0, s, N, pred, target, node_flags, weights, S, nn);
  else hipLaunchKernelGGL(k_edm_loss_sums<false>, dim3(B), dim3(1024), 0, s, N, pred, target, node_flags, weights, S, nn);
}

// reduction 'none': loss_b = S_b / n_b^2 * rlw ; reduction 'mean' (rainbow_loss.py:94-96): (sum_b S_b) / n_b^2 * rlw
__global__ void k_edm_loss_finish(int B, const float* __restrict__ S, const float* __restrict__ nn, float rlw, int mean,
                                  float* __restrict__ loss) {
  __shared__ float s_tot;
  if (mean) {   // one thread sums in index order (the order is part of the result's bits), the others wait
    if (threadIdx.x == 0) {
      float tot = 0.f;
      for (int b = 0; b < B; ++b) tot += S[b];
      s_tot = tot;
    }
    __syncthreads();
  }
  const float tot = mean ? s_tot : 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) loss[b] = (mean ? tot : S[b]) / (nn[b] * nn[b]) * rlw;
}

extern "C" {

int swg_heun_churn(int32_t B, int32_t N, const float* x, const float* raw_noise, const float* node_flags,
                   float noise_coef, float* x_hat, void* stream) {
  if (!x || !raw_noise || !node_flags || !x_hat) return SWG_E_NULL;
  if (B <= 0 || N <= 0) return SWG_E_SHAPE;
  SwgProfScope ps(KID_HEUN, 0.0, 12.0 * B * (double)N * N, (hipStream_t)stream);
  hipLaunchKernelGGL(k_heun_churn, dim3(cdiv(N, 32), cdiv(N, 32), B), dim3(256), 0, (hipStream_t)stream, N, x,
                     raw_noise, node_flags, noise_coef, x_hat);
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_heun_euler(int32_t B, int32_t N, const float* x_hat, const float* denoised, const float* node_flags,
                   float t_hat, float h, int32_t mask_out, float* d_cur, float* x_prime, void* stream) {
  if (!x_hat || !denoised || !node_flags || !x_prime) return SWG_E_NULL;
  if (B <= 0 || N <= 0 || t_hat == 0.f) return SWG_E_SHAPE;
  const size_t total = (size_t)B * N * N;
  SwgProfScope ps(KID_HEUN, 0.0, 16.0 * total, (hipStream_t)stream);
  if (N % 4 == 0 && aligned16(x_hat, denoised, node_flags, d_cur, x_prime))
    hipLaunchKernelGGL(k_heun_euler4, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N,
                       total / 4, x_hat, denoised, node_flags, 1.0f / t_hat, h, mask_out, d_cur, x_prime);
  else
    hipLaunchKernelGGL(k_heun_euler, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, total,
                       x_hat, denoised, node_flags, 1.0f / t_hat, h, mask_out, d_cur, x_prime);
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_heun_correct(int32_t B, int32_t N, const float* x_hat, const float* x_prime, const float* d_cur,
                     const float* denoised2, const float* node_flags, float t_prime, float h, float* x_next,
                     void* stream) {
  if (!x_hat || !x_prime || !d_cur || !denoised2 || !node_flags || !x_next) return SWG_E_NULL;
  if (B <= 0 || N <= 0 || t_prime == 0.f) return SWG_E_SHAPE;
  const size_t total = (size_t)B * N * N;
  SwgProfScope ps(KID_HEUN, 0.0, 20.0 * total, (hipStream_t)stream);
  if (N % 4 == 0 && aligned16(x_hat, x_prime, d_cur, denoised2, node_flags, x_next))
    hipLaunchKernelGGL(k_heun_correct4, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N,
                       total / 4, x_hat, x_prime, d_cur, denoised2, node_flags, 1.0f / t_prime, h, x_next);
  else
    hipLaunchKernelGGL(k_heun_correct, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N,
                       total, x_hat, x_prime, d_cur, denoised2, node_flags, 1.0f / t_prime, h, x_next);
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_mask_adjs(int32_t B, int32_t N, float* a, const float* node_flags, void* stream) {
  if (!a || !node_flags) return SWG_E_NULL;
  if (B <= 0 || N <= 0) return SWG_E_SHAPE;
  const size_t total = (size_t)B * N * N;
  hipLaunchKernelGGL(k_mask_adjs, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, N, total,
                     a, node_flags);
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_sym_check(int32_t B, int32_t N, const float* a, float* acc, void* stream) {
  if (!a || !acc) return SWG_E_NULL;
  if (B <= 0 || N <= 0) return SWG_E_SHAPE;
  hipLaunchKernelGGL(k_sym_check, dim3(cdiv(N, 32), cdiv(N, 32), B), dim3(256), 0, (hipStream_t)stream, N, a, acc);
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_edm_objective(int32_t B, int32_t N, const float* clean, const float* node_flags, const float* rnd_normal,
                      const float* raw_noise, float* noisy, float* noise_added, float* sigmas, float* weights,
                      float* c_skip, float* c_out, float* c_in, float* c_noise, void* stream) {
  if (!clean || !node_flags || !rnd_normal || !raw_noise || !noisy || !sigmas || !weights) return SWG_E_NULL;
  if (B <= 0 || N <= 0) return SWG_E_SHAPE;
  hipLaunchKernelGGL(k_edm_train_params, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, B, rnd_normal, sigmas,
                     weights, c_skip, c_out, c_in, c_noise);
  hipLaunchKernelGGL(k_edm_noisy, dim3(cdiv(N, 32), cdiv(N, 32), B), dim3(256), 0, (hipStream_t)stream, N, clean,
                     raw_noise, node_flags, sigmas, noisy, noise_added);
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_edm_loss(int32_t B, int32_t N, const float* pred, const float* target, const float* node_flags,
                 const float* weights, float regression_loss_weight, int32_t reduction_mean, float* loss,
                 float* scratch2B, void* stream) {
  if (!pred || !target || !node_flags || !loss || !scratch2B) return SWG_E_NULL;
  if (B <= 0 || N <= 0) return SWG_E_SHAPE;
  launch_edm_loss_sums(B, N, pred, target, node_flags, weights, scratch2B, scratch2B + B, (hipStream_t)stream);
  hipLaunchKernelGGL(k_edm_loss_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, B, scratch2B, scratch2B + B,
                     regression_loss_weight, reduction_mean, loss);
  SWG_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"

// ---- swg_edm_loss_noalloc: the loss without a caller workspace (for callers that cannot allocate) ------------
// The per-graph partial sums (S_b, n_b) of k_edm_loss_sums live in a file-scope device array instead of the
// caller's scratch2B.  `scratch` below is the HOST-side stand-in of that array: host code only ever uses the
// per-device address obtained once through hipGetSymbolAddress.  That first call per device may load the code object
// lazily, so it must happen OUTSIDE a stream capture: swg_edm_loss_noalloc_warmup() (or any earlier call of the entry
// point on that device) resolves it.
#define SWG_LOSS_NOALLOC_MAX_B 65536
static __device__ __align__(256) float scratch[2 * SWG_LOSS_NOALLOC_MAX_B];  // 512 KiB per GPU the library runs on

#include <mutex>
namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
float* g_scratch_ptr[kMaxDevices] = {};  // device address of `scratch` on each GPU, filled on first use there

// device address of this GPU's copy of `scratch` (the current device must be the one `stream` belongs to)
int loss_scratch_ptr(float** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= kMaxDevices) return SWG_E_UNSUPPORTED;
  std::lock_guard<std::mutex> lk(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    float* ptr = nullptr;
    e = hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return (int)e;
    if (!ptr) return SWG_E_NULL;
    g_scratch_ptr[dev] = ptr;
  }
  *out = g_scratch_ptr[dev];
  return 0;
}
}  // namespace

extern "C" int swg_edm_loss_noalloc_warmup(void) {
  float* ptr = nullptr;
  return loss_scratch_ptr(&ptr);
}

extern "C" int swg_edm_loss_noalloc(int32_t B, int32_t N, const float* pred, const float* target,
                                    const float* node_flags, const float* weights, float regression_loss_weight,
                                    int32_t reduction_mean, float* loss, void* stream) {
  if (!pred || !target || !node_flags || !loss) return SWG_E_NULL;
  if (B <= 0 || N <= 0) return SWG_E_SHAPE;
  if (B > SWG_LOSS_NOALLOC_MAX_B) return SWG_E_WORKSPACE;  // beyond the static array: use swg_edm_loss
  float* ptr = nullptr;
  const int rc = loss_scratch_ptr(&ptr);
  if (rc != 0) return rc;
  // every slot b < B of both halves is overwritten by k_edm_loss_sums before k_edm_loss_finish reads it, so the
  // previous call's contents need no clearing
  launch_edm_loss_sums(B, N, pred, target, node_flags, weights, ptr, ptr + SWG_LOSS_NOALLOC_MAX_B, (hipStream_t)stream);
  hipLaunchKernelGGL(k_edm_loss_finish, dim3(1), dim3(64), 0, (hipStream_t)stream, B, ptr, ptr + SWG_LOSS_NOALLOC_MAX_B,
                     regression_loss_weight, reduction_mean, loss);
  SWG_LAUNCH_CHECK();
  return 0;
}
