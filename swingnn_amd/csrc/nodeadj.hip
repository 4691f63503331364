// Head and tail of the molecule network (NodeAdjSwinGNN, model/swin_gnn/swin_gnn.py:960-1021) and the node-feature
// Heun updates of NodeAdjEDMSampler (runner/mcmc_sampler/edm.py:367-426).  The trunk between them is the SwinGNN
// trunk (api.hip).  These graphs are small (N = 16 / 48): everything here is a memory-bound row kernel; the C0 x C0
// layers of the read-out run on the MFMA GEMM (gemm.hip) between them.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------
// Input planes + PatchEmbed.  Convolution input channels, in the order of :985-996:
//   [sc_adj (Ca)] adj (Ca) | row planes: [sc_node (Cn)] node (Cn) | column planes: [sc_node (Cn)] node (Cn)
// where a row plane holds node[b, r, .] at pixel (r, c) and a column plane node[b, c, .] (:991-993), the node planes
// are masked by flags[r] && flags[c] (:994), adj / node carry the c_in factor of the preconditioner
// (precond.py:100) and the self-conditioning tensors do not.  Then conv(k = s = p) -> LN -> silu(shift + y(1+scale)).
// ------------------------------------------------------------------------------------------------
struct NaEmbedArgs {
  const float* adj; const float* node; const float* sc_adj; const float* sc_node; const float* flags;
  const float* c_in; int c_in_stride;
  const float* cond; int cond_stride, cond_off;
  const float* w; const float* bias; const float* g; const float* b;
  float* tok;
  int B, N, p, in_ch, C, G, Ca, Cn;
};

// value of convolution input channel ci at pixel (r, c) of graph bi (channel order above)
__device__ __forceinline__ float na_input_value(const NaEmbedArgs& a, int bi, int ci, int r, int c) {
  const int nA = a.G * a.Ca, nF = a.G * a.Cn;
  const float cin = a.c_in ? a.c_in[(size_t)bi * a.c_in_stride] : 1.0f;
  if (ci < nA) {
    const bool is_sc = a.G == 2 && ci < a.Ca;
    const int ch = is_sc ? ci : ci - (a.G - 1) * a.Ca;
    const size_t off = (((size_t)bi * a.Ca + ch) * a.N + r) * a.N + c;
    if (is_sc) return a.sc_adj ? a.sc_adj[off] : 0.f;
    return cin * a.adj[off];
  }
  const int cf = ci - nA;
  const int col_plane = cf / nF;
  const int cj = cf - col_plane * nF;
  const bool is_sc = a.G == 2 && cj < a.Cn;
  const int ch = is_sc ? cj : cj - (a.G - 1) * a.Cn;
  const int n = col_plane ? c : r;
  if (a.flags[(size_t)bi * a.N + r] == 0.f || a.flags[(size_t)bi * a.N + c] == 0.f) return 0.f;
  const size_t off = ((size_t)bi * a.N + n) * a.Cn + ch;
  if (is_sc) return a.sc_node ? a.sc_node[off] : 0.f;
  return cin * a.node[off];
}

// the same taps as a GEMM operand (backward recompute): bf16 [T, Kp], column k = (channel, i, j) of the conv weight
__global__ __launch_bounds__(256) void k_na_im2col(NaEmbedArgs a, bf16* __restrict__ out, int Kp) {
  const int hp = a.N / a.p, L = hp * hp, pp = a.p * a.p, kin = a.in_ch * pp;
  const size_t total = (size_t)a.B * L * Kp;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int k = (int)(i % Kp);
    const size_t row = i / Kp;
    float v = 0.f;
    if (k < kin) {
      const int bi = (int)(row / L), rem = (int)(row - (size_t)bi * L);
      const int Y = rem / hp, X = rem - Y * hp;
      const int ci = k / pp, ij = k - ci * pp;
      const int ii = ij / a.p, jj = ij - ii * a.p;
      v = na_input_value(a, bi, ci, Y * a.p + ii, X * a.p + jj);
    }
    out[i] = (bf16)v;
  }
}

template <int LANES, int V>
__global__ __launch_bounds__(256) void k_na_embed(NaEmbedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int RPB = 256 / LANES;
  const int hp = a.N / a.p;
  const int L = hp * hp;
  const int pp = a.p * a.p;
  const int kin = a.in_ch * pp;
  float* sW = smem;               // [kin][C]
  float* sIn = smem + kin * a.C;  // [RPB][kin]
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
    const float v = na_input_value(a, bi, ci, Y * a.p + ii, X * a.p + jj);
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

// ------------------------------------------------------------------------------------------------
// Output layers of the read-out MLPs (:913-914, fc2 with Ca / Cn <= 8 outputs): one 8-lane group per row of the
// hidden activation h (bf16 [rows, ldh], written by the GEMM with the GELU epilogue), fp32 weights.
//   MODE 0 (adjacency): rows are pixels in token-patch order m = (b L + Y hp + X) p^2 + i p + j; the Ca values go to
//          F[b, ca, Y p + i, X p + j] (the permutes of :1001-1003); mask + symmetrise + preconditioning follow in
//          k_sym_precond.
//   MODE 1 (nodes): rows are (b, n); out[b, n, cn] = flag * (c_skip * node + c_out * y)  (mask_nodes :1015,
//          precond.py:103-105).
// ------------------------------------------------------------------------------------------------
struct NaHeadArgs {
  const bf16* h; int ldh;
  const float* w2; const float* b2;
  int rows, C, Co;
  // MODE 0
  float* F; int N, p;
  // MODE 1
  const float* node; const float* flags; const float* coef; float* out;
};

template <int MODE>
__global__ __launch_bounds__(256) void k_na_head_out(NaHeadArgs a) {
  const int row = blockIdx.x * 32 + ((int)threadIdx.x >> 3);
  const int g = threadIdx.x & 7;
  if (row >= a.rows) return;
  // a lane holds channels 8g + 64k .. +7, k < 2  (C <= 128)
  float hv[16];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int c0 = 8 * g + 64 * k;
    if (c0 < a.C) {  // C % 4 == 0 and ldh = ceil32(C): the 8-wide load stays inside the (zero-padded) row
      const bf16x8 v = *reinterpret_cast<const bf16x8*>(a.h + (size_t)row * a.ldh + c0);
#pragma unroll
      for (int e = 0; e < 8; ++e) hv[8 * k + e] = (c0 + e < a.C) ? (float)v[e] : 0.f;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) hv[8 * k + e] = 0.f;
    }
  }
  for (int o = 0; o < a.Co; ++o) {
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c0 = 8 * g + 64 * k;
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (c0 + e < a.C) acc = fmaf(hv[8 * k + e], a.w2[(size_t)o * a.C + c0 + e], acc);
    }
    acc = group_sum<8>(acc) + a.b2[o];
    if (g == (o & 7)) {
      if (MODE == 0) {
        const int pp = a.p * a.p, hp = a.N / a.p, L = hp * hp;
        const int t = row / pp, ij = row - t * pp;
        const int bi = t / L, rem = t - bi * L;
        const int Y = rem / hp, X = rem - Y * hp;
        const int i = ij / a.p, j = ij - i * a.p;
        a.F[(((size_t)bi * a.Co + o) * a.N + (Y * a.p + i)) * a.N + (X * a.p + j)] = acc;
      } else {
        const int bi = row / a.N;
        const float c_skip = a.coef ? a.coef[4 * bi + 0] : 0.f;
        const float c_out = a.coef ? a.coef[4 * bi + 1] : 1.f;
        const size_t off = (size_t)row * a.Co + o;
        float v = 0.f;
        if (a.flags[row] != 0.f) v = __fmaf_rn(c_skip, a.coef ? a.node[off] : 0.f, __fmul_rn(c_out, acc));
        a.out[off] = v;
      }
    }
  }
}

// Padding-aware pooling (:1006-1007): pooled[b, r, :] = flag_r / N * sum_c flag_c * rep[b, r, c, :], rep rows in
// token-patch order (bf16 [B L p^2, ldr]); one block per (b, r), one thread per channel.
__global__ __launch_bounds__(128) void k_na_pool(const bf16* __restrict__ rep, int ldr, int N, int p,
                                                 const float* __restrict__ flags, bf16* __restrict__ pooled) {
  const int bi = blockIdx.x / N, r = blockIdx.x - bi * N;
  const int ch = threadIdx.x;
  if (ch >= ldr) return;
  const int hp = N / p, L = hp * hp, pp = p * p;
  const int Y = r / p, i = r - Y * p;
  float acc = 0.f;
  if (flags[(size_t)bi * N + r] != 0.f) {
    for (int c = 0; c < N; ++c) {
      if (flags[(size_t)bi * N + c] == 0.f) continue;
      const int X = c / p, j = c - X * p;
      const size_t m = ((size_t)bi * L + Y * hp + X) * pp + i * p + j;
      acc += (float)rep[m * ldr + ch];
    }
  }
  pooled[(size_t)blockIdx.x * ldr + ch] = (bf16)(acc / (float)N);
}


// ---- backward of the tail ---------------------------------------------------------------------------------
// dF of the adjacency head in token-patch row order: dy[m, ca] = mask * c_out * 0.5 (dD[b,ca,r,c] + dD[b,ca,c,r])
// (derivative of mask -> symmetrise -> c_skip x + c_out F -> mask, :1016-1018 and precond.py:102-104)
__global__ __launch_bounds__(256) void k_na_dadj(const float* __restrict__ dD, const float* __restrict__ flags,
                                                 const float* __restrict__ coef, int B, int N, int p, int Ca,
                                                 float* __restrict__ dy) {
  const int hp = N / p, L = hp * hp, pp = p * p;
  const size_t total = (size_t)B * N * N * Ca;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ca = (int)(i % Ca);
    const size_t m = i / Ca;
    const int ij = (int)(m % pp);
    const size_t t = m / pp;
    const int bi = (int)(t / L), rem = (int)(t - (size_t)bi * L);
    const int Y = rem / hp, X = rem - Y * hp;
    const int ii = ij / p, jj = ij - ii * p;
    const int r = Y * p + ii, c = X * p + jj;
    float v = 0.f;
    if (flags[(size_t)bi * N + r] != 0.f && flags[(size_t)bi * N + c] != 0.f) {
      const size_t base = ((size_t)bi * Ca + ca) * N * N;
      const float c_out = coef ? coef[4 * bi + 1] : 1.f;
      v = 0.5f * c_out * (dD[base + (size_t)r * N + c] + dD[base + (size_t)c * N + r]);
    }
    dy[i] = v;
  }
}

// node head: dy[row, cn] = flag * c_out * dD[row, cn]
__global__ __launch_bounds__(256) void k_na_dnode(const float* __restrict__ dD, const float* __restrict__ flags,
                                                  const float* __restrict__ coef, size_t total, int N, int Cn,
                                                  float* __restrict__ dy) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const size_t row = i / Cn;
  const int bi = (int)(row / N);
  dy[i] = flags[row] != 0.f ? (coef ? coef[4 * bi + 1] : 1.f) * dD[i] : 0.f;
}

// gradient of the narrow output layer w.r.t. its input: dact[row, c] = sum_o dy[row, o] W2[o, c]  (bf16 [rows, ld],
// padding columns zero); one thread per (row, 8 channels)
__global__ __launch_bounds__(256) void k_na_head_bwd(const float* __restrict__ dy, const float* __restrict__ w2, int rows,
                                                     int C, int Co, bf16* __restrict__ dact, int ld) {
  const int chunks = ld / 8;
  const size_t total = (size_t)rows * chunks;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % chunks);
    const size_t row = i / chunks;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int o = 0; o < Co; ++o) {
      const float d = dy[row * Co + o];
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (8 * ch + e < C) acc[e] = fmaf(d, w2[(size_t)o * C + 8 * ch + e], acc[e]);
    }
    bf16x8 v;
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (bf16)acc[e];
    *reinterpret_cast<bf16x8*>(dact + row * ld + 8 * ch) = v;
  }
}

// backward of k_na_pool: drep[m(b, r, c), ch] = flag_r flag_c / N * dpooled[b N + r, ch]   (fp32 [P, C], full overwrite)
__global__ __launch_bounds__(256) void k_na_pool_bwd(const float* __restrict__ dpooled, const float* __restrict__ flags,
                                                     int B, int N, int p, int C, float* __restrict__ drep) {
  const int hp = N / p, L = hp * hp, pp = p * p;
  const size_t total = (size_t)B * N * N * C;
  const float inv = 1.0f / (float)N;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int ch = (int)(i % C);
    const size_t m = i / C;
    const int ij = (int)(m % pp);
    const size_t t = m / pp;
    const int bi = (int)(t / L), rem = (int)(t - (size_t)bi * L);
    const int Y = rem / hp, X = rem - Y * hp;
    const int ii = ij / p, jj = ij - ii * p;
    const int r = Y * p + ii, c = X * p + jj;
    float v = 0.f;
    if (flags[(size_t)bi * N + r] != 0.f && flags[(size_t)bi * N + c] != 0.f)
      v = inv * dpooled[((size_t)bi * N + r) * C + ch];
    drep[i] = v;
  }
}

// ---- node-feature halves of the training objective and loss --------------------------------------------------
// noisy = clean + mask(sigma_b * raw)   (runner/objectives/edm.py:243-247; clean is not re-masked)
__global__ __launch_bounds__(256) void k_nodes_noisy(size_t total, int N, int Cn, const float* __restrict__ clean,
                                                     const float* __restrict__ raw, const float* __restrict__ flags,
                                                     const float* __restrict__ sigmas, float* __restrict__ noisy) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const size_t row = i / Cn;
  noisy[i] = clean[i] + (flags[row] != 0.f ? sigmas[row / N] * raw[i] : 0.f);
}

// loss_b = w_b sum_{n,c} flag (pred - target)^2 / (sum flags_b) / Cn * node_loss_weight  (loss/rainbow_loss.py:211-248);
// one block per graph, fixed summation order
__global__ __launch_bounds__(256) void k_nodes_loss(int N, int Cn, const float* __restrict__ pred,
                                                    const float* __restrict__ target, const float* __restrict__ flags,
                                                    const float* __restrict__ weights, float lw, float* __restrict__ loss) {
  __shared__ float red[8];
  const int bi = blockIdx.x;
  float acc = 0.f, cnt = 0.f;
  for (int i = threadIdx.x; i < N * Cn; i += 256) {
    const int n = i / Cn;
    if (flags[(size_t)bi * N + n] != 0.f) {
      const float d = pred[(size_t)bi * N * Cn + i] - target[(size_t)bi * N * Cn + i];
      acc = fmaf(d, d, acc);
    }
  }
  for (int n = threadIdx.x; n < N; n += 256) cnt += flags[(size_t)bi * N + n];
  acc = group_sum<64>(acc);
  cnt = group_sum<64>(cnt);
  if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = acc; red[4 + (threadIdx.x >> 6)] = cnt; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float s = (red[0] + red[1]) + (red[2] + red[3]);
    const float c = (red[4] + red[5]) + (red[6] + red[7]);
    loss[bi] = (weights ? weights[bi] : 1.f) * s / c / (float)Cn * lw;
  }
}

__global__ __launch_bounds__(256) void k_nodes_loss_bwd(size_t total, int N, int Cn, const float* __restrict__ pred,
                                                        const float* __restrict__ target, const float* __restrict__ flags,
                                                        const float* __restrict__ weights, const float* __restrict__ gl,
                                                        float lw, float* __restrict__ gp) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const size_t row = i / Cn;
  const int bi = (int)(row / N);
  float v = 0.f;
  if (flags[row] != 0.f) {
    float cnt = 0.f;
    for (int n = 0; n < N; ++n) cnt += flags[(size_t)bi * N + n];
    v = 2.f * (pred[i] - target[i]) * (weights ? weights[bi] : 1.f) * gl[bi] * lw / cnt / (float)Cn;
  }
  gp[i] = v;
}

// ---- node-feature Heun updates: x [B, N, Cn], flags [B, N] ----
__global__ __launch_bounds__(256) void k_nodes_churn(size_t total, int Cn, const float* __restrict__ x,
                                                     const float* __restrict__ raw, const float* __restrict__ flags,
                                                     float coef, float* __restrict__ xh) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  xh[i] = flags[i / Cn] != 0.f ? x[i] + coef * raw[i] : 0.f;
}

__global__ __launch_bounds__(256) void k_nodes_euler(size_t total, int Cn, const float* __restrict__ xh,
                                                     const float* __restrict__ den, const float* __restrict__ flags,
                                                     float inv_t, float h, int mask_out, float* __restrict__ d_cur,
                                                     float* __restrict__ xp) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const bool m = flags[i / Cn] != 0.f;
  const float xv = xh[i];
  const float d = m ? (inv_t * xv - inv_t * den[i]) : 0.f;
  if (d_cur) d_cur[i] = d;
  const float o = xv + h * d;
  xp[i] = (mask_out && !m) ? 0.f : o;
}

__global__ __launch_bounds__(256) void k_nodes_correct(size_t total, int Cn, const float* __restrict__ xh,
                                                       const float* __restrict__ xp, const float* __restrict__ d_cur,
                                                       const float* __restrict__ den2, const float* __restrict__ flags,
                                                       float inv_tp, float h, float* __restrict__ xn) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  float o = 0.f;
  if (flags[i / Cn] != 0.f) {
    const float dp = inv_tp * xp[i] - inv_tp * den2[i];
    o = xh[i] + h * (0.5f * d_cur[i] + 0.5f * dp);
  }
  xn[i] = o;
}

}  // namespace


static NaEmbedArgs na_embed_args(const swg_net* net, const swg_nodeadj* na, int B, const float* adj, const float* node,
                                 const float* sc_adj, const float* sc_node, const float* flags, const float* c_in,
                                 int c_in_stride, const float* cond, int cond_stride, float* tok) {
  const int G = net->self_cond ? 2 : 1;
  NaEmbedArgs a = {adj, node, sc_adj, sc_node, flags, c_in, c_in_stride, cond, cond_stride, net->pe_cond_off,
                   net->pe_w, net->pe_b, net->pe_ln.g, net->pe_ln.b, tok,
                   B, net->N, net->p, net->in_ch, net->C0, G, na->Ca, na->Cn};
  return a;
}

static inline unsigned na_grid(size_t items) { return (unsigned)max((size_t)1, min((size_t)4096, (items + 255) / 256)); }

int launch_na_im2col(const swg_net* net, const swg_nodeadj* na, int B, const float* adj, const float* node,
                     const float* sc_adj, const float* sc_node, const float* flags, const float* c_in, int c_in_stride,
                     bf16* out, int Kp, hipStream_t s) {
  const int G = net->self_cond ? 2 : 1;
  if (net->in_ch != G * (na->Ca + 2 * na->Cn) || net->N % net->p) return SWG_E_SHAPE;
  NaEmbedArgs a = na_embed_args(net, na, B, adj, node, sc_adj, sc_node, flags, c_in, c_in_stride, nullptr, 0, nullptr);
  const int hp = net->N / net->p;
  const size_t total = (size_t)B * hp * hp * Kp;
  SwgProfScope ps(KID_BWD_ROWS, 0.0, 2.0 * total, s);
  hipLaunchKernelGGL(k_na_im2col, dim3(na_grid(total)), dim3(256), 0, s, a, out, Kp);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_na_dadj(const float* dD, const float* flags, const float* coef, int B, int N, int p, int Ca, float* dy,
                   hipStream_t s) {
  const size_t total = (size_t)B * N * N * Ca;
  SwgProfScope ps(KID_BWD_ROWS, 0.0, 12.0 * total, s);
  hipLaunchKernelGGL(k_na_dadj, dim3(na_grid(total)), dim3(256), 0, s, dD, flags, coef, B, N, p, Ca, dy);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_na_dnode(const float* dD, const float* flags, const float* coef, int B, int N, int Cn, float* dy,
                    hipStream_t s) {
  const size_t total = (size_t)B * N * Cn;
  SwgProfScope ps(KID_BWD_ROWS, 0.0, 8.0 * total, s);
  hipLaunchKernelGGL(k_na_dnode, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dD, flags, coef, total, N, Cn, dy);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_na_head_bwd(const float* dy, const float* w2, int rows, int C, int Co, bf16* dact, int ld, hipStream_t s) {
  if (ld % 8 || ld < C) return SWG_E_SHAPE;
  const size_t total = (size_t)rows * (ld / 8);
  SwgProfScope ps(KID_BWD_ROWS, 2.0 * rows * (double)C * Co, rows * (4.0 * Co + 2.0 * ld), s);
  hipLaunchKernelGGL(k_na_head_bwd, dim3(na_grid(total)), dim3(256), 0, s, dy, w2, rows, C, Co, dact, ld);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_na_pool_bwd(const float* dpooled, const float* flags, int B, int N, int p, int C, float* drep, hipStream_t s) {
  const size_t total = (size_t)B * N * N * C;
  SwgProfScope ps(KID_BWD_ROWS, 0.0, 4.0 * total, s);
  hipLaunchKernelGGL(k_na_pool_bwd, dim3(na_grid(total)), dim3(256), 0, s, dpooled, flags, B, N, p, C, drep);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_na_embed(const swg_net* net, const swg_nodeadj* na, int B, const float* adj, const float* node,
                    const float* sc_adj, const float* sc_node, const float* flags, const float* c_in, int c_in_stride,
                    const float* cond, int cond_stride, float* tok, hipStream_t s) {
  const int G = net->self_cond ? 2 : 1;
  if (net->in_ch != G * (na->Ca + 2 * na->Cn) || net->N % net->p || net->C0 % 4) return SWG_E_SHAPE;
  const int kin = net->in_ch * net->p * net->p;
  RowGeom geo = row_geom(net->C0 / 4);
  const size_t lds = (size_t)(kin * net->C0 + (256 / geo.lanes) * kin) * sizeof(float);
  if (lds > 64 * 1024) return SWG_E_UNSUPPORTED;
  NaEmbedArgs a = {adj, node, sc_adj, sc_node, flags, c_in, c_in_stride, cond, cond_stride, net->pe_cond_off,
                   net->pe_w, net->pe_b, net->pe_ln.g, net->pe_ln.b, tok,
                   B, net->N, net->p, net->in_ch, net->C0, G, na->Ca, na->Cn};
  const int hp = net->N / net->p;
  const int rows = B * hp * hp;
  SwgProfScope ps(KID_NODEADJ, 2.0 * rows * (double)net->C0 * kin,
                  4.0 * B * (double)net->N * net->N * G * na->Ca + 4.0 * rows * (double)net->C0, s);
#define CALL(LN, VV) hipLaunchKernelGGL((k_na_embed<LN, VV>), dim3(cdiv(rows, 256 / LN)), dim3(256), lds, s, a)
  SWG_ROW_DISPATCH(geo, CALL);
#undef CALL
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_na_head_adj(const bf16* h, int ldh, int B, int N, int p, int C, int Ca, const float* w2, const float* b2,
                       float* F, hipStream_t s) {
  if (C > 128 || C % 4 || ldh % 8 || ldh < C || !w2 || !b2) return SWG_E_SHAPE;
  NaHeadArgs a = {};
  a.h = h; a.ldh = ldh; a.w2 = w2; a.b2 = b2; a.rows = B * N * N; a.C = C; a.Co = Ca; a.F = F; a.N = N; a.p = p;
  SwgProfScope ps(KID_NODEADJ, 2.0 * a.rows * (double)C * Ca, a.rows * (2.0 * ldh + 4.0 * Ca), s);
  hipLaunchKernelGGL(k_na_head_out<0>, dim3(cdiv(a.rows, 32)), dim3(256), 0, s, a);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_na_head_node(const bf16* h, int ldh, int B, int N, int C, int Cn, const float* w2, const float* b2,
                        const float* node, const float* flags, const float* coef, float* out, hipStream_t s) {
  if (C > 128 || C % 4 || ldh % 8 || ldh < C || !w2 || !b2) return SWG_E_SHAPE;
  NaHeadArgs a = {};
  a.h = h; a.ldh = ldh; a.w2 = w2; a.b2 = b2; a.rows = B * N; a.C = C; a.Co = Cn; a.N = N;
  a.node = node; a.flags = flags; a.coef = coef; a.out = out;
  SwgProfScope ps(KID_NODEADJ, 2.0 * a.rows * (double)C * Cn, a.rows * (2.0 * ldh + 8.0 * Cn), s);
  hipLaunchKernelGGL(k_na_head_out<1>, dim3(cdiv(a.rows, 32)), dim3(256), 0, s, a);
  SWG_LAUNCH_CHECK();
  return 0;
}

int launch_na_pool(const bf16* rep, int ldr, int B, int N, int p, const float* flags, bf16* pooled, hipStream_t s) {
  if (ldr > 128) return SWG_E_UNSUPPORTED;
  SwgProfScope ps(KID_NODEADJ, 0.0, 2.0 * B * (double)N * N * ldr + 2.0 * B * (double)N * ldr, s);
  hipLaunchKernelGGL(k_na_pool, dim3(B * N), dim3(128), 0, s, rep, ldr, N, p, flags, pooled);
  SWG_LAUNCH_CHECK();
  return 0;
}

extern "C" {

int swg_heun_nodes_churn(int32_t B, int32_t N, int32_t Cn, const float* x, const float* raw, const float* node_flags,
                         float noise_coef, float* x_hat, void* stream) {
  if (!x || !raw || !node_flags || !x_hat) return SWG_E_NULL;
  if (B <= 0 || N <= 0 || Cn <= 0) return SWG_E_SHAPE;
  const size_t total = (size_t)B * N * Cn;
  SwgProfScope ps(KID_HEUN, 0.0, 12.0 * total, (hipStream_t)stream);
  hipLaunchKernelGGL(k_nodes_churn, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, total, Cn,
                     x, raw, node_flags, noise_coef, x_hat);
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_heun_nodes_euler(int32_t B, int32_t N, int32_t Cn, const float* x_hat, const float* denoised,
                         const float* node_flags, float t_hat, float h, int32_t mask_out, float* d_cur, float* x_prime,
                         void* stream) {
  if (!x_hat || !denoised || !node_flags || !x_prime) return SWG_E_NULL;
  if (B <= 0 || N <= 0 || Cn <= 0 || t_hat == 0.f) return SWG_E_SHAPE;
  const size_t total = (size_t)B * N * Cn;
  SwgProfScope ps(KID_HEUN, 0.0, 16.0 * total, (hipStream_t)stream);
  hipLaunchKernelGGL(k_nodes_euler, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, total, Cn,
                     x_hat, denoised, node_flags, 1.0f / t_hat, h, mask_out, d_cur, x_prime);
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_heun_nodes_correct(int32_t B, int32_t N, int32_t Cn, const float* x_hat, const float* x_prime,
                           const float* d_cur, const float* denoised2, const float* node_flags, float t_prime, float h,
                           float* x_next, void* stream) {
  if (!x_hat || !x_prime || !d_cur || !denoised2 || !node_flags || !x_next) return SWG_E_NULL;
  if (B <= 0 || N <= 0 || Cn <= 0 || t_prime == 0.f) return SWG_E_SHAPE;
  const size_t total = (size_t)B * N * Cn;
  SwgProfScope ps(KID_HEUN, 0.0, 20.0 * total, (hipStream_t)stream);
  hipLaunchKernelGGL(k_nodes_correct, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, total,
                     Cn, x_hat, x_prime, d_cur, denoised2, node_flags, 1.0f / t_prime, h, x_next);
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_nodes_objective(int32_t B, int32_t N, int32_t Cn, const float* clean, const float* raw_noise,
                        const float* node_flags, const float* sigmas, float* noisy, void* stream) {
  if (!clean || !raw_noise || !node_flags || !sigmas || !noisy) return SWG_E_NULL;
  if (B <= 0 || N <= 0 || Cn <= 0) return SWG_E_SHAPE;
  const size_t total = (size_t)B * N * Cn;
  hipLaunchKernelGGL(k_nodes_noisy, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, total, N, Cn,
                     clean, raw_noise, node_flags, sigmas, noisy);
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_nodes_loss(int32_t B, int32_t N, int32_t Cn, const float* pred, const float* target, const float* node_flags,
                   const float* weights, float node_loss_weight, float* loss, void* stream) {
  if (!pred || !target || !node_flags || !loss) return SWG_E_NULL;
  if (B <= 0 || N <= 0 || Cn <= 0) return SWG_E_SHAPE;
  hipLaunchKernelGGL(k_nodes_loss, dim3(B), dim3(256), 0, (hipStream_t)stream, N, Cn, pred, target, node_flags, weights,
                     node_loss_weight, loss);
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_nodes_loss_bwd(int32_t B, int32_t N, int32_t Cn, const float* pred, const float* target, const float* node_flags,
                       const float* weights, const float* grad_loss, float node_loss_weight, float* grad_pred,
                       void* stream) {
  if (!pred || !target || !node_flags || !grad_loss || !grad_pred) return SWG_E_NULL;
  if (B <= 0 || N <= 0 || Cn <= 0) return SWG_E_SHAPE;
  const size_t total = (size_t)B * N * Cn;
  hipLaunchKernelGGL(k_nodes_loss_bwd, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, total, N,
                     Cn, pred, target, node_flags, weights, grad_loss, node_loss_weight, grad_pred);
  SWG_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
