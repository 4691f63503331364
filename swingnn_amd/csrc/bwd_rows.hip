// Memory-bound kernels of the training backward pass: LayerNorm / GELU / conditioning
// derivatives, the 2x2 patch permutation of PatchMerging / PatchBreakup, bf16 casts and transposes that feed the
// MFMA GEMMs (dX = dY W, dW = dY^T X), column sums for the bias gradients, and the embedding / read-out / loss
// ends of the chain.  Activations and gradients on the residual stream are fp32; bf16 exists only as GEMM operands.
// Parameter gradients are ACCUMULATED (+=) into fp32 buffers with vector atomics (float add order is not fixed).
#include "common.h"

namespace {

constexpr float kInvSqrt2 = 0.70710678118654752440f;
constexpr float kInvSqrt2Pi = 0.39894228040143267794f;

__device__ __forceinline__ float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * kInvSqrt2)); }
__device__ __forceinline__ float gelu_grad(float x) {
  return 0.5f * (1.0f + erff(x * kInvSqrt2)) + x * kInvSqrt2Pi * __expf(-0.5f * x * x);
}

// ---- fp32 [rows, C] (row stride ldx) -> bf16 [rows, ldo], columns [C, ldo) zero ----
__global__ __launch_bounds__(256) void k_cast_pad(const float* __restrict__ x, int rows, int C, int ldx,
                                                  bf16* __restrict__ out, int ldo) {
  const int nst = ldo >> 2;
  const size_t total = (size_t)rows * nst;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int r = (int)(i / nst), c = (int)(i - (size_t)r * nst) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c < C) v = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + c);
    *reinterpret_cast<uint2*>(out + (size_t)r * ldo + c) = pack_bf16x4(v.x, v.y, v.z, v.w);
  }
}

// ---- src [M, ld] (N columns used) -> dst bf16 [N, ldd] = src^T, columns [M, ldd) zero ----
template <typename T>
__global__ __launch_bounds__(256) void k_transpose_bf16(const T* __restrict__ src, int M, int N, int ld,
                                                        bf16* __restrict__ dst, int ldd) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;
  for (int k = ty; k < 32; k += 8) {
    const int m = m0 + k, n = n0 + tx;
    tile[k][tx] = (m < M && n < N) ? (float)src[(size_t)m * ld + n] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int n = n0 + k, m = m0 + tx;
    if (n < N && m < ldd) dst[(size_t)n * ldd + m] = (bf16)tile[tx][k];
  }
}

// ---- out[n] += sum_m src[m][n] ----
// 64 columns x 4 row phases per workgroup (narrow matrices keep all 256 threads busy), rpb rows per workgroup,
// one atomic per column and workgroup
template <typename T>
__global__ __launch_bounds__(256) void k_colsum(const T* __restrict__ src, int M, int N, int ld, int rpb,
                                                float* __restrict__ out) {
  __shared__ float part[4][64];
  const int cx = threadIdx.x & 63, ry = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + cx;
  const int r0 = blockIdx.y * rpb, r1 = min(M, r0 + rpb);
  float s0 = 0.f, s1 = 0.f;
  if (n < N) {
    int m = r0 + ry;
    for (; m + 4 < r1; m += 8) {
      s0 += (float)src[(size_t)m * ld + n];
      s1 += (float)src[(size_t)(m + 4) * ld + n];
    }
    if (m < r1) s0 += (float)src[(size_t)m * ld + n];
  }
  part[ry][cx] = s0 + s1;
  __syncthreads();
  if (ry == 0 && n < N) atomicAdd(out + n, (part[0][cx] + part[1][cx]) + (part[2][cx] + part[3][cx]));
}

// ---- GELU (exact erf form, swin_gnn.py:10,15) on the bf16 hidden buffer, forward and derivative ----
__global__ __launch_bounds__(256) void k_gelu_fwd(const bf16* __restrict__ h, bf16* __restrict__ a, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(h + 8 * i);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)gelu_exact((float)v[j]);
    *reinterpret_cast<bf16x8*>(a + 8 * i) = o;
  }
}
__global__ __launch_bounds__(256) void k_gelu_bwd(const bf16* __restrict__ h, const bf16* __restrict__ da,
                                                  bf16* __restrict__ dh, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const bf16x8 v = *reinterpret_cast<const bf16x8*>(h + 8 * i);
    const bf16x8 g = *reinterpret_cast<const bf16x8*>(da + 8 * i);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)g[j] * gelu_grad((float)v[j]));
    *reinterpret_cast<bf16x8*>(dh + 8 * i) = o;
  }
}

// ---- fused forms: the bf16 GEMM operand AND the bias gradient from one read of the gradient ----
// workgroup = 16 column chunks x 16 row phases over `rpb` rows; column sums folded through LDS, one atomic per column
// cast: fp32 dy [rows, C] -> bf16 [rows, ldo] (columns [C, ldo) zero), db[c] += sum_rows dy
__global__ __launch_bounds__(256) void k_cast_pad_colsum(const float* __restrict__ x, int rows, int C, int ldx,
                                                         bf16* __restrict__ out, int ldo, int rpb, float* __restrict__ db) {
  __shared__ float part[16][65];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = (blockIdx.x * 16 + cx) * 4;
  const int r0 = blockIdx.y * rpb, r1 = min(rows, r0 + rpb);
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < ldo) {
    for (int r = r0 + ry; r < r1; r += 16) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < C) v = *reinterpret_cast<const float4*>(x + (size_t)r * ldx + c);
      *reinterpret_cast<uint2*>(out + (size_t)r * ldo + c) = pack_bf16x4(v.x, v.y, v.z, v.w);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  part[ry][4 * cx] = acc.x; part[ry][4 * cx + 1] = acc.y; part[ry][4 * cx + 2] = acc.z; part[ry][4 * cx + 3] = acc.w;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int n = blockIdx.x * 64 + threadIdx.x;
    if (n < C) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += part[k][threadIdx.x];
      atomicAdd(db + n, t);
    }
  }
}
// GELU backward: dh = da gelu'(h) (bf16 [rows, ld], in place allowed), db[c] += sum_rows dh for c < N
// a_out (optional): the activation a = gelu(h) itself, for the fc2 weight gradient -- one read of h serves both
__global__ __launch_bounds__(256) void k_gelu_bwd_colsum(const bf16* __restrict__ h, const bf16* __restrict__ da,
                                                         bf16* __restrict__ dh, bf16* __restrict__ a_out, int rows, int N,
                                                         int ld, int rpb, float* __restrict__ db) {
  __shared__ float part[16][129];
  const int cx = threadIdx.x & 15, ry = threadIdx.x >> 4;
  const int c = (blockIdx.x * 16 + cx) * 8;
  const int r0 = blockIdx.y * rpb, r1 = min(rows, r0 + rpb);
  float acc[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) acc[j] = 0.f;
  if (c < ld) {
    for (int r = r0 + ry; r < r1; r += 16) {
      const size_t o = (size_t)r * ld + c;
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(h + o);
      const bf16x8 g = *reinterpret_cast<const bf16x8*>(da + o);
      bf16x8 w, av;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = (float)v[j];
        const float cdf = 0.5f * (1.0f + erff(x * kInvSqrt2));  // gelu = x cdf, gelu' = cdf + x pdf
        av[j] = (bf16)(x * cdf);
        w[j] = (bf16)((float)g[j] * (cdf + x * kInvSqrt2Pi * __expf(-0.5f * x * x)));
        acc[j] += (float)w[j];
      }
      if (a_out) *reinterpret_cast<bf16x8*>(a_out + o) = av;
      *reinterpret_cast<bf16x8*>(dh + o) = w;
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) part[ry][8 * cx + j] = acc[j];
  __syncthreads();
  if (threadIdx.x < 128) {
    const int n = blockIdx.x * 128 + threadIdx.x;
    if (n < N) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < 16; ++k) t += part[k][threadIdx.x];
      atomicAdd(db + n, t);
    }
  }
}

// ---- LayerNorm backward: one wavefront per row (grid-stride), V float4 chunks per lane ----
//   xhat = (x - mean) rstd, dxhat = dy g, dx = rstd (dxhat - mean(dxhat) - xhat mean(dxhat xhat)) [+ dres]
//   dg += sum_rows dy xhat, db += sum_rows dy
// Optional 2x2 patch permutation folded into the addressing (PatchMerging / PatchBreakup, k_perm22 below): with
// perm_res > 0 a row is the PACKED row (b, Y, X) of 4 tokens x Cs channels and the operands named in perm_mask
// (1: x, 2: dy, 4: dx) live in TOKEN layout [B, perm_res^2, Cs]; chunk c of the row is channel chunk c % (Cs/4) of token
// (2Y + (k&1), 2X + (k>>1)), k = c / (Cs/4).
// LANES lanes per row (64 / LANES rows per wavefront at a time): narrow rows keep every lane busy (C = 96: 8 lanes x 3
// float4, 8 rows per wave instead of one row on 24 of 64 lanes).
template <int LANES, int V>
__global__ __launch_bounds__(256) void k_ln_bwd(const float* __restrict__ x, const float* __restrict__ dy,
                                                const float* __restrict__ g, const float* __restrict__ dres, int rows,
                                                int C, float* __restrict__ dx, float* __restrict__ dg,
                                                float* __restrict__ db, int perm_res, int perm_mask,
                                                bf16* __restrict__ dxb) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [2][C]
  constexpr int RPW = 64 / LANES;  // rows per wavefront
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gl = lane % LANES, grp = lane / LANES;
  const int nch = C >> 2;
  float4 ag[V], ab[V], gg[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    ag[i] = ab[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = gl + LANES * i;
    gg[i] = (c < nch) ? *reinterpret_cast<const float4*>(g + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int c = threadIdx.x; c < 2 * C; c += 256) red[c] = 0.f;
  const float invC = 1.0f / (float)C;
  const int pcs = nch >> 2, r2 = perm_res >> 1;  // chunks per token; packed grid side
  for (int row0 = (blockIdx.x * 4 + wave) * RPW; row0 < rows; row0 += gridDim.x * 4 * RPW) {
    const int row = row0 + grp;
    const bool live = row < rows;
    float4 xv[V], dv[V];
    float s = 0.f;
    size_t ptok = 0;  // token (2Y, 2X) of the packed row
    if (perm_res && live) {
      const int X = row % r2, t = row / r2, Y = t % r2, b = t / r2;
      ptok = ((size_t)b * perm_res + 2 * Y) * perm_res + 2 * X;
    }
    size_t po[V];  // element offset of chunk i in token layout
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = gl + LANES * i;
      po[i] = 0;
      if (perm_res && c < nch) {
        const int k = c / pcs, cc = c - k * pcs;
        po[i] = (ptok + (size_t)(k & 1) * perm_res + (k >> 1)) * (size_t)(4 * pcs) + 4 * cc;
      }
    }
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = gl + LANES * i;
      xv[i] = dv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live && c < nch) {
        xv[i] = *reinterpret_cast<const float4*>(x + ((perm_mask & 1) ? po[i] : (size_t)row * C + 4 * c));
        dv[i] = *reinterpret_cast<const float4*>(dy + ((perm_mask & 2) ? po[i] : (size_t)row * C + 4 * c));
      }
      s += (xv[i].x + xv[i].y) + (xv[i].z + xv[i].w);
    }
    const float mean = group_sum<LANES>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i)
      if (gl + LANES * i < nch) {
        xv[i].x -= mean; xv[i].y -= mean; xv[i].z -= mean; xv[i].w -= mean;
        q += (xv[i].x * xv[i].x + xv[i].y * xv[i].y) + (xv[i].z * xv[i].z + xv[i].w * xv[i].w);
      }
    const float rstd = ln_rstd(group_sum<LANES>(q) * invC);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      xv[i].x *= rstd; xv[i].y *= rstd; xv[i].z *= rstd; xv[i].w *= rstd;  // xhat
      ag[i].x += dv[i].x * xv[i].x; ag[i].y += dv[i].y * xv[i].y; ag[i].z += dv[i].z * xv[i].z; ag[i].w += dv[i].w * xv[i].w;
      ab[i].x += dv[i].x; ab[i].y += dv[i].y; ab[i].z += dv[i].z; ab[i].w += dv[i].w;
      dv[i].x *= gg[i].x; dv[i].y *= gg[i].y; dv[i].z *= gg[i].z; dv[i].w *= gg[i].w;  // dxhat
      s1 += (dv[i].x + dv[i].y) + (dv[i].z + dv[i].w);
      s2 += (dv[i].x * xv[i].x + dv[i].y * xv[i].y) + (dv[i].z * xv[i].z + dv[i].w * xv[i].w);
    }
    const float m1 = group_sum<LANES>(s1) * invC, m2 = group_sum<LANES>(s2) * invC;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = gl + LANES * i;
      if (live && c < nch) {
        float4 o;
        o.x = rstd * (dv[i].x - m1 - xv[i].x * m2); o.y = rstd * (dv[i].y - m1 - xv[i].y * m2);
        o.z = rstd * (dv[i].z - m1 - xv[i].z * m2); o.w = rstd * (dv[i].w - m1 - xv[i].w * m2);
        if (dres) {
          const float4 r = *reinterpret_cast<const float4*>(dres + (size_t)row * C + 4 * c);
          o.x += r.x; o.y += r.y; o.z += r.z; o.w += r.w;
        }
        if (dxb) *reinterpret_cast<uint2*>(dxb + (size_t)row * C + 4 * c) = pack_bf16x4(o.x, o.y, o.z, o.w);
        else *reinterpret_cast<float4*>(dx + ((perm_mask & 4) ? po[i] : (size_t)row * C + 4 * c)) = o;
      }
    }
  }
  // fold the column partials of the 4 x RPW row groups through LDS (one group after the other: a fixed order inside
  // the workgroup), then one atomic per column per workgroup
  __syncthreads();
  for (int w = 0; w < 4 * RPW; ++w) {
    if (wave * RPW + grp == w) {
#pragma unroll
      for (int i = 0; i < V; ++i) {
        const int c = gl + LANES * i;
        if (c < nch) {
          float4* pg = reinterpret_cast<float4*>(red + 4 * c);
          float4* pb = reinterpret_cast<float4*>(red + C + 4 * c);
          float4 a = *pg, b = *pb;
          a.x += ag[i].x; a.y += ag[i].y; a.z += ag[i].z; a.w += ag[i].w;
          b.x += ab[i].x; b.y += ab[i].y; b.z += ab[i].z; b.w += ab[i].w;
          *pg = a; *pb = b;
        }
      }
    }
    __syncthreads();
  }
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dg + c, red[c]);
    atomicAdd(db + c, red[C + c]);
  }
}

// ---- conditioning backward: xc = silu(z), z = shift + x (1 + scale)  (swin_gnn.py:238-240, :571-574) ----
// dx = dxc silu'(z) (1 + scale); dscale[b,c] += sum_t dz x; dshift[b,c] += sum_t dz.  grid (B, token chunks).
// thread = (float4 column chunk, token phase): C / 4 chunks x (256 / chunks) phases, column partials folded through LDS
__global__ __launch_bounds__(256) void k_cond_bwd(const float* __restrict__ x, const float* __restrict__ dxc,
                                                  const float* __restrict__ cond, int cond_stride, int cond_off, int C,
                                                  int L, int tpb, float* __restrict__ dx, float* __restrict__ dcond,
                                                  int dcond_stride) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [2][C]
  const int b = blockIdx.x;
  const int t0 = blockIdx.y * tpb, t1 = min(L, t0 + tpb);
  const float* crow = cond + (size_t)b * cond_stride + cond_off;
  float* drow = dcond + (size_t)b * dcond_stride + cond_off;
  const int nch = C >> 2;
  const int nph = max(1, 256 / nch);
  for (int i = threadIdx.x; i < 2 * C; i += 256) red[i] = 0.f;
  __syncthreads();
  for (int cb = 0; cb < nch; cb += 256) {  // C > 1024 only: more than one pass over the chunks
    const int ch = cb + (int)threadIdx.x % min(nch, 256), ph = (int)threadIdx.x / min(nch, 256);
    if (ch < nch && ph < nph) {
      const float4 scv = *reinterpret_cast<const float4*>(crow + 4 * ch);
      const float4 shv = *reinterpret_cast<const float4*>(crow + C + 4 * ch);
      const float sc[4] = {scv.x + 1.f, scv.y + 1.f, scv.z + 1.f, scv.w + 1.f}, sh[4] = {shv.x, shv.y, shv.z, shv.w};
      float asc[4] = {0.f, 0.f, 0.f, 0.f}, ash[4] = {0.f, 0.f, 0.f, 0.f};
      for (int t = t0 + ph; t < t1; t += nph) {
        const size_t o = ((size_t)b * L + t) * C + 4 * ch;
        const float4 xq = *reinterpret_cast<const float4*>(x + o);
        const float4 dq = *reinterpret_cast<const float4*>(dxc + o);
        const float xv[4] = {xq.x, xq.y, xq.z, xq.w}, dv[4] = {dq.x, dq.y, dq.z, dq.w};
        float ov[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float z = sh[k] + xv[k] * sc[k];
          const float sg = 1.0f / (1.0f + __expf(-z));
          const float dz = dv[k] * sg * (1.0f + z * (1.0f - sg));
          ov[k] = dz * sc[k];
          asc[k] += dz * xv[k];
          ash[k] += dz;
        }
        if (dx) *reinterpret_cast<float4*>(dx + o) = make_float4(ov[0], ov[1], ov[2], ov[3]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        atomicAdd(&red[4 * ch + k], asc[k]);
        atomicAdd(&red[C + 4 * ch + k], ash[k]);
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(drow + c, red[c]);
    atomicAdd(drow + C + c, red[C + c]);
  }
}

// ---- LayerNorm1 backward + conditioning backward of a block's attention half in ONE pass over the rows
// (swin_gnn.py:238-244, :272):  z = shift + x (1 + scale), x1 = silu(z), xn = LN1(x1);  given dxn and the shortcut
// gradient dres:  dx1 = dres + LN1'(dxn),  dz = dx1 silu'(z),  dx = dz (1 + scale),  dscale += sum_t dz x,
// dshift += sum_t dz,  dg += sum dxn xhat,  db += sum dxn.  x1 is recomputed from the block input: neither x1 nor dx1
// ever goes to memory.  grid (token chunks, B); a group of LANES lanes per row, V float4 chunks per lane.
template <int LANES, int V>
__global__ __launch_bounds__(256) void k_ln_cond_bwd(const float* __restrict__ x, const float* __restrict__ dxn,
                                                     const float* __restrict__ dres, const float* __restrict__ g,
                                                     const float* __restrict__ cond, int cond_stride, int cond_off, int C,
                                                     int L, int tpb, float* __restrict__ dx, float* __restrict__ dg,
                                                     float* __restrict__ db, float* __restrict__ dcond, int dcond_stride) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // [4][C]: dg, db, dscale, dshift
  constexpr int GROUPS = 256 / LANES;
  const int li = threadIdx.x % LANES, gi = threadIdx.x / LANES;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * tpb, t1 = min(L, t0 + tpb);
  const float* crow = cond + (size_t)b * cond_stride + cond_off;
  const int nch = C >> 2;
  float4 gg[V], sc1[V], sh[V], ag[V], ab[V], asc[V], ash[V];
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = li + LANES * i;
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    gg[i] = sc1[i] = sh[i] = ag[i] = ab[i] = asc[i] = ash[i] = zero;
    if (c < nch) {
      gg[i] = *reinterpret_cast<const float4*>(g + 4 * c);
      const float4 s4 = *reinterpret_cast<const float4*>(crow + 4 * c);
      sc1[i] = make_float4(s4.x + 1.f, s4.y + 1.f, s4.z + 1.f, s4.w + 1.f);
      sh[i] = *reinterpret_cast<const float4*>(crow + C + 4 * c);
    }
  }
  for (int i = threadIdx.x; i < 4 * C; i += 256) red[i] = 0.f;
  const float invC = 1.0f / (float)C;
  for (int t = t0 + gi; t < t1; t += GROUPS) {
    const size_t ro = ((size_t)b * L + t) * C;
    float xv[V][4], zz[V][4], sg[V][4], xh[V][4], dn[V][4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = li + LANES * i;
      float4 xq = make_float4(0.f, 0.f, 0.f, 0.f), dq = xq;
      if (c < nch) {
        xq = *reinterpret_cast<const float4*>(x + ro + 4 * c);
        dq = *reinterpret_cast<const float4*>(dxn + ro + 4 * c);
      }
      const float xa[4] = {xq.x, xq.y, xq.z, xq.w}, da[4] = {dq.x, dq.y, dq.z, dq.w};
      const float sa[4] = {sc1[i].x, sc1[i].y, sc1[i].z, sc1[i].w}, ha[4] = {sh[i].x, sh[i].y, sh[i].z, sh[i].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xv[i][k] = xa[k];
        dn[i][k] = da[k];
        const float z = ha[k] + xa[k] * sa[k];
        const float sig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.44269504088896340736f));
        zz[i][k] = z;
        sg[i][k] = sig;
        xh[i][k] = (c < nch) ? z * sig : 0.f;  // x1 = silu(z)
        s += xh[i][k];
      }
    }
    const float mean = group_sum<LANES>(s) * invC;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i)
      if (li + LANES * i < nch) {
#pragma unroll
        for (int k = 0; k < 4; ++k) { xh[i][k] -= mean; q += xh[i][k] * xh[i][k]; }
      }
    const float rstd = ln_rstd(group_sum<LANES>(q) * invC);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const float ga[4] = {gg[i].x, gg[i].y, gg[i].z, gg[i].w};
      float* pag = reinterpret_cast<float*>(&ag[i]);
      float* pab = reinterpret_cast<float*>(&ab[i]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        xh[i][k] *= rstd;  // xhat
        pag[k] += dn[i][k] * xh[i][k];
        pab[k] += dn[i][k];
        dn[i][k] *= ga[k];  // dxhat
        s1 += dn[i][k];
        s2 += dn[i][k] * xh[i][k];
      }
    }
    const float m1 = group_sum<LANES>(s1) * invC, m2 = group_sum<LANES>(s2) * invC;
#pragma unroll
    for (int i = 0; i < V; ++i) {
      const int c = li + LANES * i;
      if (c < nch) {
        const float4 rq = *reinterpret_cast<const float4*>(dres + ro + 4 * c);
        const float ra[4] = {rq.x, rq.y, rq.z, rq.w};
        const float sa[4] = {sc1[i].x, sc1[i].y, sc1[i].z, sc1[i].w};
        float* pas = reinterpret_cast<float*>(&asc[i]);
        float* pah = reinterpret_cast<float*>(&ash[i]);
        float o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float dx1 = rstd * (dn[i][k] - m1 - xh[i][k] * m2) + ra[k];
          const float dz = dx1 * sg[i][k] * (1.0f + zz[i][k] * (1.0f - sg[i][k]));
          o[k] = dz * sa[k];
          pas[k] += dz * xv[i][k];
          pah[k] += dz;
        }
        *reinterpret_cast<float4*>(dx + ro + 4 * c) = make_float4(o[0], o[1], o[2], o[3]);
      }
    }
  }
  __syncthreads();  // red zeroed by everyone before the first LDS add
#pragma unroll
  for (int i = 0; i < V; ++i) {
    const int c = li + LANES * i;
    if (c < nch) {
      const float* pag = reinterpret_cast<const float*>(&ag[i]);
      const float* pab = reinterpret_cast<const float*>(&ab[i]);
      const float* pas = reinterpret_cast<const float*>(&asc[i]);
      const float* pah = reinterpret_cast<const float*>(&ash[i]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        atomicAdd(&red[4 * c + k], pag[k]);
        atomicAdd(&red[C + 4 * c + k], pab[k]);
        atomicAdd(&red[2 * C + 4 * c + k], pas[k]);
        atomicAdd(&red[3 * C + 4 * c + k], pah[k]);
      }
    }
  }
  __syncthreads();
  float* drow = dcond + (size_t)b * dcond_stride + cond_off;
  for (int c = threadIdx.x; c < C; c += 256) {
    atomicAdd(dg + c, red[c]);
    atomicAdd(db + c, red[C + c]);
    atomicAdd(drow + c, red[2 * C + c]);
    atomicAdd(drow + C + c, red[3 * C + c]);
  }
}

// ---- 2x2 patch permutation shared by PatchMerging (gather, swin_gnn.py:325-330) and PatchBreakup (scatter, :389-397):
// packed [B, (res/2)^2, 4 C] <-> tokens [B, res^2, C]; chunk k of a packed row is token (2Y + (k&1), 2X + (k>>1)).
// dir 0: packed <- tokens, dir 1: tokens <- packed.  res = side of the TOKEN grid. ----
__global__ __launch_bounds__(256) void k_perm22(const float* __restrict__ src, float* __restrict__ dst, int B, int res,
                                                int C, int dir) {
  const int r2 = res >> 1, cs = C >> 2;
  const size_t total = (size_t)B * r2 * r2 * 4 * cs;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int cc = (int)(i % cs);
    size_t r = i / cs;
    const int k = (int)(r & 3);
    r >>= 2;
    const int X = (int)(r % r2);
    r /= r2;
    const int Y = (int)(r % r2);
    const int b = (int)(r / r2);
    const size_t tok = (size_t)b * res * res + (size_t)(2 * Y + (k & 1)) * res + (2 * X + (k >> 1));
    const size_t po = i * 4, to = tok * C + 4 * cc;
    if (dir == 0) *reinterpret_cast<float4*>(dst + po) = *reinterpret_cast<const float4*>(src + to);
    else *reinterpret_cast<float4*>(dst + to) = *reinterpret_cast<const float4*>(src + po);
  }
}

// ---- PatchEmbed input taps as a GEMM operand: bf16 [T, Kp], column k = (channel, i, j) of the conv weight
// (swin_gnn.py:566, channel order cat([self_cond, x]) :769-771, x scaled by c_in precond.py:47) ----
__global__ __launch_bounds__(256) void k_embed_im2col(const float* __restrict__ x, const float* __restrict__ sc,
                                                      const float* __restrict__ c_in, int c_in_stride, int B, int N, int p,
                                                      int in_ch, int self_cond, bf16* __restrict__ out, int Kp) {
  const int hp = N / p, L = hp * hp, pp = p * p, kin = in_ch * pp;
  const size_t total = (size_t)B * L * Kp;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % Kp);
    const size_t row = i / Kp;
    float v = 0.f;
    if (k < kin) {
      const int bi = (int)(row / L), rem = (int)(row - (size_t)bi * L);
      const int Y = rem / hp, X = rem - Y * hp;
      const int ci = k / pp, ij = k - ci * pp;
      const int ii = ij / p, jj = ij - ii * p;
      const size_t off = ((size_t)bi * N + (Y * p + ii)) * N + (X * p + jj);
      const bool is_x = (!self_cond) || ci == in_ch - 1;
      if (is_x) v = (c_in ? c_in[(size_t)bi * c_in_stride] : 1.0f) * x[off];
      else if (sc) v = sc[off];
    }
    out[i] = (bf16)v;
  }
}

// ---- read-out end of the chain (swin_gnn.py:773-778, precond.py:49-51): D = mask(c_skip x + c_out sym(mask(R))) with
// R[b, Yp+i, Xp+j] = tokn[t] . A[:, ip+j] + bias.  dR = 0.5 c_out mask (dD + dD^T) as a bf16 [T, Kp] GEMM operand. ----
__global__ __launch_bounds__(256) void k_readout_dpix(const float* __restrict__ dD, const float* __restrict__ flags,
                                                      const float* __restrict__ coef, int B, int N, int p,
                                                      bf16* __restrict__ out, int Kp) {
  const int hp = N / p, L = hp * hp, pp = p * p;
  const size_t total = (size_t)B * L * Kp;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % Kp);
    const size_t row = i / Kp;
    float v = 0.f;
    if (k < pp) {
      const int bi = (int)(row / L), rem = (int)(row - (size_t)bi * L);
      const int Y = rem / hp, X = rem - Y * hp;
      const int ii = k / p, jj = k - ii * p;
      const int r = Y * p + ii, c = X * p + jj;
      if (flags[(size_t)bi * N + r] != 0.f && flags[(size_t)bi * N + c] != 0.f) {
        const size_t base = (size_t)bi * N * N;
        const float c_out = coef ? coef[4 * bi + 1] : 1.f;
        v = 0.5f * c_out * (dD[base + (size_t)r * N + c] + dD[base + (size_t)c * N + r]);
      }
    }
    out[i] = (bf16)v;
  }
}

// ---- grad[N, K] += tmp[N, ldt][:, :K] (weight gradients whose K is not a multiple of 4) ----
__global__ __launch_bounds__(256) void k_add_slice(const float* __restrict__ tmp, int N, int K, int ldt,
                                                   float* __restrict__ grad) {
  const int total = N * K;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
    const int n = i / K, k = i - n * K;
    grad[i] += tmp[(size_t)n * ldt + k];
  }
}

// ---- loss/rainbow_loss.py:84-111 backward, reduction 'none': loss_b = rlw w_b sum mask (p - t)^2 / nf_b^2
// dpred = gloss_b rlw w_b 2 mask (p - t) / nf_b^2 ----
__global__ __launch_bounds__(256) void k_edm_loss_bwd(int N, const float* __restrict__ pred, const float* __restrict__ tgt,
                                                      const float* __restrict__ flags, const float* __restrict__ weights,
                                                      const float* __restrict__ gloss, float rlw,
                                                      float* __restrict__ dpred) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  float nf = 0.f;
  for (int i = threadIdx.x; i < N; i += 256) nf += flags[(size_t)b * N + i];
  nf = group_sum<64>(nf);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = nf;
  __syncthreads();
  nf = red[0] + red[1] + red[2] + red[3];
  const float k = 2.0f * gloss[b] * rlw * (weights ? weights[b] : 1.0f) / (nf * nf);
  const size_t base = (size_t)b * N * N;
  for (int i = blockIdx.x * 256 + threadIdx.x; i < N * N; i += gridDim.x * 256) {
    const int r = i / N, c = i - r * N;
    const bool m = flags[(size_t)b * N + r] != 0.f && flags[(size_t)b * N + c] != 0.f;
    dpred[base + i] = m ? k * (pred[base + i] - tgt[base + i]) : 0.f;
  }
}

inline unsigned gs_grid(size_t items) { return (unsigned)max((size_t)1, min((size_t)4096, (items + 255) / 256)); }

}  // namespace

int launch_cast_pad(const float* x, int rows, int C, int ldx, bf16* out, int ldo, hipStream_t s) {
  if (rows <= 0) return 0;
  if (C % 4 || ldo % 4 || ldx % 4 || ldo < C) return SWG_E_SHAPE;
  SwgProfScope ps(KID_BWD_ROWS, 0.0, (double)rows * (4.0 * C + 2.0 * ldo), s);
  hipLaunchKernelGGL(k_cast_pad, dim3(gs_grid((size_t)rows * (ldo / 4))), dim3(256), 0, s, x, rows, C, ldx, out, ldo);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_transpose_f32(const float* src, int M, int N, int ld, bf16* dst, int ldd, hipStream_t s) {
  if (M <= 0 || N <= 0) return 0;
  if (ldd < M) return SWG_E_SHAPE;
  SwgProfScope ps(KID_BWD_ROWS, 0.0, (double)M * N * 6.0, s);
  hipLaunchKernelGGL((k_transpose_bf16<float>), dim3(cdiv(ldd, 32), cdiv(N, 32)), dim3(256), 0, s, src, M, N, ld, dst, ldd);
  SWG_LAUNCH_CHECK();
  return 0;
}
int launch_transpose_bf16(const bf16* src, int M, int N, int ld, bf16* dst, int ldd, hipStream_t s) {
  if (M <= 0 || N <= 0) return 0;
  if (ldd < M) return SWG_E_SHAPE;
  SwgProfScope ps(KID_BWD_ROWS, 0.0, (double)M * N * 4.0, s);
  hipLaunchKernelGGL((k_transpose_bf16<bf16>), dim3(cdiv(ldd, 32), cdiv(N, 32)), dim3(256), 0, s, src, M, N, ld, dst, ldd);
  SWG_LAUNCH_CHECK();
  return 0;
}

static inline int colsum_rpb(int M) { return max(64, min(1024, cdiv(M, 256))); }
int launch_colsum_f32(const float* src, int M, int N, int ld, float* out, hipStream_t s) {
  if (M <= 0 || N <= 0) return 0;
  const int rpb = colsum_rpb(M);
  SwgProfScope ps(KID_BWD_ROWS, 0.0, (double)M * N * 4.0, s);
  hipLaunchKernelGGL((k_colsum<float>), dim3(cdiv(N, 64), cdiv(M, rpb)), dim3(256), 0, s, src, M, N, ld, rpb, out);
  SWG_LAUNCH_CHECK();
  return 0;
}
int launch_colsum_bf16(const bf16* src, int M, int N, int ld, float* out, hipStream_t s) {
  if (M <= 0 || N <= 0) return 0;
  const int rpb = colsum_rpb(M);
  SwgProfScope ps(KID_BWD_ROWS, 0.0, (double)M * N * 2.0, s);
  hipLaunchKernelGGL((k_colsum<bf16>), dim3(cdiv(N, 64), cdiv(M, rpb)), dim3(256), 0, s, src, M, N, ld, rpb, out);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_gelu_fwd(const bf16* h, bf16* a, size_t n, hipStream_t s) {
  if (n == 0) return 0;
  if (n % 8) return SWG_E_SHAPE;
  SwgProfScope ps(KID_BWD_ROWS, 0.0, 4.0 * (double)n, s);
  hipLaunchKernelGGL(k_gelu_fwd, dim3(gs_grid(n / 8)), dim3(256), 0, s, h, a, n / 8);
  SWG_LAUNCH_CHECK();
  return 0;
}
int launch_gelu_bwd(const bf16* h, const bf16* da, bf16* dh, size_t n, hipStream_t s) {
  if (n == 0) return 0;
  if (n % 8) return SWG_E_SHAPE;
  SwgProfScope ps(KID_BWD_ROWS, 0.0, 6.0 * (double)n, s);
  hipLaunchKernelGGL(k_gelu_bwd, dim3(gs_grid(n / 8)), dim3(256), 0, s, h, da, dh, n / 8);
  SWG_LAUNCH_CHECK();
  return 0;
}

static inline int fused_rpb(int rows, int col_blocks) {  // >= ~2048 workgroups, at least 64 rows each
  return max(64, cdiv(rows, max(1, 2048 / col_blocks)));
}
int launch_cast_pad_colsum(const float* x, int rows, int C, int ldx, bf16* out, int ldo, float* db, hipStream_t s) {
  if (rows <= 0) return 0;
  if (C % 4 || ldo % 4 || ldx % 4 || ldo < C) return SWG_E_SHAPE;
  if (!x || !out || !db) return SWG_E_NULL;
  const int cb = cdiv(ldo, 64), rpb = fused_rpb(rows, cb);
  SwgProfScope ps(KID_BWD_ROWS, 0.0, (double)rows * (4.0 * C + 2.0 * ldo), s);
  hipLaunchKernelGGL(k_cast_pad_colsum, dim3(cb, cdiv(rows, rpb)), dim3(256), 0, s, x, rows, C, ldx, out, ldo, rpb, db);
  SWG_LAUNCH_CHECK();
  return 0;
}
int launch_gelu_bwd_colsum(const bf16* h, const bf16* da, bf16* dh, bf16* a_out, int rows, int N, int ld, float* db,
                           hipStream_t s) {
  if (rows <= 0) return 0;
  if (ld % 8 || N > ld) return SWG_E_SHAPE;
  if (!h || !da || !dh || !db) return SWG_E_NULL;
  const int cb = cdiv(ld, 128), rpb = fused_rpb(rows, cb);
  SwgProfScope ps(KID_BWD_ROWS, 0.0, (a_out ? 8.0 : 6.0) * (double)rows * ld, s);
  hipLaunchKernelGGL(k_gelu_bwd_colsum, dim3(cb, cdiv(rows, rpb)), dim3(256), 0, s, h, da, dh, a_out, rows, N, ld, rpb, db);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_ln_bwd(const float* x, const float* dy, const float* g, const float* dres, int rows, int C, float* dx,
                  float* dg, float* db, hipStream_t s, int perm_res, int perm_mask, bf16* dx_bf16) {
  if (rows <= 0) return 0;
  if (C % 4 || C > 3072) return SWG_E_UNSUPPORTED;
  if (dx_bf16) dx = reinterpret_cast<float*>(dx_bf16);  // bf16 [rows, C] INSTEAD of the fp32 output (GEMM operand)
  if (perm_res && (C % 16 || perm_res % 2 || dres)) return SWG_E_SHAPE;
  if (!perm_res) perm_mask = 0;
  if (!x || !dy || !g || !dx || !dg || !db) return SWG_E_NULL;
  const int nch = C / 4;
  const int lanes = nch <= 32 ? 8 : nch <= 64 ? 16 : nch <= 128 ? 32 : 64;
  static const int grid_cap = [] { const char* v = getenv("SWG_LNB_GRID"); return v ? atoi(v) : 512; }();  // 2 per CU: more only adds column-sum atomics (1024: +0.2 ms per step)
  const int grid = min(cdiv(rows, 4 * (64 / lanes)), grid_cap);
  const size_t lds = (size_t)2 * C * sizeof(float);
  SwgProfScope ps(KID_BWD_ROWS, 0.0, (double)rows * C * (dres ? 16.0 : 12.0), s);
#define LNB(LN, VV) \
  hipLaunchKernelGGL((k_ln_bwd<LN, VV>), dim3(grid), dim3(256), lds, s, x, dy, g, dres, rows, C, dx, dg, db, perm_res, perm_mask, dx_bf16)
  if (nch <= 24) LNB(8, 3);
  else if (nch <= 32) LNB(8, 4);
  else if (nch <= 48) LNB(16, 3);
  else if (nch <= 64) LNB(16, 4);
  else if (nch <= 96) LNB(32, 3);
  else if (nch <= 128) LNB(32, 4);
  else if (nch <= 192) LNB(64, 3);
  else if (nch <= 256) LNB(64, 4);
  else if (nch <= 384) LNB(64, 6);
  else if (nch <= 512) LNB(64, 8);
  else LNB(64, 12);
#undef LNB
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_cond_bwd(const float* x, const float* dxc, const float* cond, int cond_stride, int cond_off, int B, int L, int C,
                    float* dx, float* dcond, int dcond_stride, hipStream_t s) {
  if (B <= 0 || L <= 0) return 0;
  if (!x || !dxc || !cond || !dcond) return SWG_E_NULL;
  const int chunks = max(1, min(L, cdiv(2048, B)));  // ~2048 workgroups
  const int tpb = cdiv(L, chunks);
  SwgProfScope ps(KID_BWD_ROWS, 0.0, (double)B * L * C * 12.0, s);
  if (C % 4) return SWG_E_SHAPE;
  hipLaunchKernelGGL(k_cond_bwd, dim3(B, cdiv(L, tpb)), dim3(256), (size_t)2 * C * sizeof(float), s, x, dxc, cond, cond_stride,
                     cond_off, C, L, tpb, dx, dcond, dcond_stride);
  SWG_LAUNCH_CHECK();
  return 0;
}

// fused LN1 + conditioning backward (k_ln_cond_bwd); C <= 1024, else the caller uses the two separate passes
bool ln_cond_bwd_supported(int C) { return C % 4 == 0 && C <= 1024; }
int launch_ln_cond_bwd(const float* x, const float* dxn, const float* dres, const float* g, const float* cond,
                       int cond_stride, int cond_off, int B, int L, int C, float* dx, float* dg, float* db, float* dcond,
                       int dcond_stride, hipStream_t s) {
  if (B <= 0 || L <= 0) return 0;
  if (!ln_cond_bwd_supported(C)) return SWG_E_UNSUPPORTED;
  if (!x || !dxn || !dres || !g || !cond || !dx || !dg || !db || !dcond) return SWG_E_NULL;
  static const int wg_target = [] { const char* v = getenv("SWG_LCB_WGS"); return v ? atoi(v) : 1024; }();  // 4096: +0.17 ms per step
  const int chunks = max(1, min(cdiv(L, 32), cdiv(wg_target, B)));  // ~wg_target workgroups, at least 32 rows each
  const int tpb = cdiv(L, chunks);
  const dim3 grid(cdiv(L, tpb), B);
  const size_t lds = (size_t)4 * C * sizeof(float);
  SwgProfScope ps(KID_BWD_ROWS, 0.0, (double)B * L * C * 16.0, s);
#define LCB(LN, VV)                                                                                                  \
  hipLaunchKernelGGL((k_ln_cond_bwd<LN, VV>), grid, dim3(256), lds, s, x, dxn, dres, g, cond, cond_stride, cond_off, C, L, \
                     tpb, dx, dg, db, dcond, dcond_stride)
  const int nch = C / 4;
  if (nch <= 32) LCB(32, 1);
  else if (nch <= 64) LCB(64, 1);
  else if (nch <= 128) LCB(64, 2);
  else if (nch <= 192) LCB(64, 3);
  else LCB(64, 4);
#undef LCB
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_perm22(const float* src, float* dst, int B, int res, int C, int dir, hipStream_t s) {
  if (B <= 0) return 0;
  if (res % 2 || C % 4) return SWG_E_SHAPE;
  SwgProfScope ps(KID_BWD_ROWS, 0.0, (double)B * res * res * C * 8.0, s);
  hipLaunchKernelGGL(k_perm22, dim3(gs_grid((size_t)B * res * res * (C / 4))), dim3(256), 0, s, src, dst, B, res, C, dir);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_embed_im2col(const swg_net* net, int B, const float* x, const float* sc, const float* c_in, int c_in_stride,
                        bf16* out, int Kp, hipStream_t s) {
  const int hp = net->N / net->p;
  SwgProfScope ps(KID_BWD_ROWS, 0.0, (double)B * net->N * net->N * 8.0, s);
  hipLaunchKernelGGL(k_embed_im2col, dim3(gs_grid((size_t)B * hp * hp * Kp)), dim3(256), 0, s, x, sc, c_in, c_in_stride, B,
                     net->N, net->p, net->in_ch, net->self_cond, out, Kp);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_readout_dpix(const float* dD, const float* flags, const float* coef, int B, int N, int p, bf16* out, int Kp,
                        hipStream_t s) {
  const int hp = N / p;
  SwgProfScope ps(KID_BWD_ROWS, 0.0, (double)B * N * N * 8.0, s);
  hipLaunchKernelGGL(k_readout_dpix, dim3(gs_grid((size_t)B * hp * hp * Kp)), dim3(256), 0, s, dD, flags, coef, B, N, p, out,
                     Kp);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_add_slice(const float* tmp, int N, int K, int ldt, float* grad, hipStream_t s) {
  if (N <= 0 || K <= 0) return 0;
  hipLaunchKernelGGL(k_add_slice, dim3(gs_grid((size_t)N * K)), dim3(256), 0, s, tmp, N, K, ldt, grad);
  SWG_LAUNCH_CHECK();
  return 0;
}

extern "C" int swg_edm_loss_bwd(int32_t B, int32_t N, const float* pred, const float* target, const float* node_flags,
                                const float* weights, const float* grad_loss, float regression_loss_weight,
                                float* grad_pred, void* stream) {
  if (!pred || !target || !node_flags || !grad_loss || !grad_pred) return SWG_E_NULL;
  if (B <= 0 || N <= 0) return SWG_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  SwgProfScope ps(KID_BWD_ROWS, 0.0, (double)B * N * N * 12.0, s);
  hipLaunchKernelGGL(k_edm_loss_bwd, dim3(min(cdiv(N * N, 256), 64), B), dim3(256), 0, s, N, pred, target, node_flags,
                     weights, grad_loss, regression_loss_weight, grad_pred);
  SWG_LAUNCH_CHECK();
  return 0;
}
