// Backward of the window-attention core (model/swin_gnn/swin_gnn.py:116-136 between the two Linear layers) on the
// packed qkv buffer in TOKEN order, the partition / shift folded into addressing exactly as in wattn.hip.
//
//   S = scale Q K^T + bias + mask, P = softmax(S), O = P V
//   dV = P^T dO,  dP = dO V^T,  dS = P (dP - rowsum(P dP)),  dQ = scale dS K,  dK = scale dS^T Q,
//   d(relative_position_bias_table)[idx(i,j), h] += dS[i,j]
//
// First version of the training path: one wavefront per (window, head), fp32 VALU on LDS-resident tiles (the
// attention core is ~15 % of a block's FLOPs; the four Linear-layer gradients around it run on the MFMA GEMM).
// Query phase: lane = query row (S, P, dP, dS rows, dQ).  Key phase: lane = key row (dK, dV from the columns of dS
// and P in LDS, stride 65 -> conflict-free both ways).  A workgroup keeps its head fixed and walks over windows, so
// the bias-table gradient is accumulated in LDS and flushed with one atomic per entry per workgroup.
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
  int nr;  // LDS tile rows: ceil16(win * win)
};

constexpr int HDP = 32;     // padded head dim
constexpr int PLD = 65;     // row stride of the n x n matrices
constexpr int NBMAX = 232;  // >= (2*8-1)^2

__device__ __forceinline__ void load_row(const bf16* p, int hd, float* r) {
#pragma unroll
  for (int d = 0; d < HDP; d += 4) {
    if (d < hd) {
      const uint2 raw = *reinterpret_cast<const uint2*>(p + d);
      const bf16x4 v = *reinterpret_cast<const bf16x4*>(&raw);
      r[d] = (float)v[0]; r[d + 1] = (float)v[1]; r[d + 2] = (float)v[2]; r[d + 3] = (float)v[3];
    } else {
      r[d] = r[d + 1] = r[d + 2] = r[d + 3] = 0.f;
    }
  }
}
__device__ __forceinline__ void put_row(float* dst, const float* r) {
#pragma unroll
  for (int d = 0; d < HDP; d += 4) *reinterpret_cast<float4*>(dst + d) = make_float4(r[d], r[d + 1], r[d + 2], r[d + 3]);
}
__device__ __forceinline__ void store_row(bf16* p, int hd, const float* r, float k) {
#pragma unroll
  for (int d = 0; d < HDP; d += 4)
    if (d < hd) *reinterpret_cast<uint2*>(p + d) = pack_bf16x4(r[d] * k, r[d + 1] * k, r[d + 2] * k, r[d + 3] * k);
}

__global__ __launch_bounds__(64) void k_wattn_bwd(WbArgs a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  // rows of the per-window tiles: NR = ceil16(tokens per window), so that small windows (4x4: 9 KB instead of 66 KB)
  // leave room for many workgroups per CU
  const int NR = a.nr;
  float* Ks = sm;
  float* Vs = Ks + NR * HDP;
  float* Qs = Vs + NR * HDP;
  float* Ds = Qs + NR * HDP;
  float* Pm = Ds + NR * HDP;
  float* Sm = Pm + NR * PLD;
  float* bias = Sm + NR * PLD;
  float* dbias = bias + NBMAX;
  int* rid = reinterpret_cast<int*>(dbias + NBMAX);

  const int lane = threadIdx.x;
  const int h = blockIdx.x;
  const int n = a.win * a.win, w2 = 2 * a.win - 1, nb = w2 * w2;
  const int nwx = a.res / a.win, nW = nwx * nwx, Ltok = a.res * a.res;
  const int hc = h * a.hd;
  for (int t = lane; t < nb; t += 64) {
    bias[t] = a.rpb[(size_t)t * a.heads + h];
    dbias[t] = 0.f;
  }
  const int il = min(lane, n - 1);
  const int iy = il / a.win, ix = il - iy * a.win;
  const bool act = lane < n;

  for (int u = blockIdx.y; u < a.B * nW; u += gridDim.y) {
    const int b = u / nW, wi = u - b * nW;
    const int wy = wi / nwx, wx = wi - wy * nwx;
    const int ys = wy * a.win + iy, xs = wx * a.win + ix;
    int y = ys + a.shift, x = xs + a.shift;
    if (y >= a.res) y -= a.res;
    if (x >= a.res) x -= a.res;
    const size_t tok = (size_t)b * Ltok + y * a.res + x;
    int myrid = 0;
    if (a.shift > 0) {
      const int ry = ys < a.res - a.win ? 0 : (ys < a.res - a.shift ? 1 : 2);
      const int rx = xs < a.res - a.win ? 0 : (xs < a.res - a.shift ? 1 : 2);
      myrid = 3 * ry + rx;
    }
    __syncthreads();  // the previous window's LDS reads are done
    rid[lane] = myrid;
    float q[HDP], go[HDP];
    {
      float t[HDP];
      const bf16* row = a.qkv + tok * a.ldq + hc;
      load_row(row, act ? a.hd : 0, q);
      load_row(a.dO + tok * a.ldo + hc, act ? a.hd : 0, go);
      if (lane < NR) {
        put_row(Qs + lane * HDP, q);
        load_row(row + a.C, act ? a.hd : 0, t);
        put_row(Ks + lane * HDP, t);
        load_row(row + 2 * a.C, act ? a.hd : 0, t);
        put_row(Vs + lane * HDP, t);
        put_row(Ds + lane * HDP, go);
      }
    }
    __syncthreads();

    bf16* drow = a.dqkv + tok * a.lddq + hc;
    if (act) {  // ---- query phase ----
      float* Sr = Sm + lane * PLD;
      float* Pr = Pm + lane * PLD;
      float mx = -3.0e38f;
      for (int j = 0; j < n; ++j) {
        const float4* kr = reinterpret_cast<const float4*>(Ks + j * HDP);
        float s = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < HDP / 4; ++d4) {
          const float4 kv = kr[d4];
          s += q[4 * d4] * kv.x + q[4 * d4 + 1] * kv.y + q[4 * d4 + 2] * kv.z + q[4 * d4 + 3] * kv.w;
        }
        const int jy = j / a.win, jx = j - jy * a.win;
        s = s * a.scale + bias[(iy - jy + a.win - 1) * w2 + (ix - jx + a.win - 1)];
        if (a.shift > 0 && rid[j] != myrid) s += -100.0f;  // swin_gnn.py:226
        Sr[j] = s;
        mx = fmaxf(mx, s);
      }
      float sum = 0.f;
      for (int j = 0; j < n; ++j) {
        const float e = __expf(Sr[j] - mx);
        Pr[j] = e;
        sum += e;
      }
      const float inv = 1.0f / sum;
      float delta = 0.f;
      for (int j = 0; j < n; ++j) {
        const float4* vr = reinterpret_cast<const float4*>(Vs + j * HDP);
        float dp = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < HDP / 4; ++d4) {
          const float4 vv = vr[d4];
          dp += go[4 * d4] * vv.x + go[4 * d4 + 1] * vv.y + go[4 * d4 + 2] * vv.z + go[4 * d4 + 3] * vv.w;
        }
        const float p = Pr[j] * inv;
        Pr[j] = p;
        Sr[j] = dp;
        delta += p * dp;
      }
      float dq[HDP];
#pragma unroll
      for (int d = 0; d < HDP; ++d) dq[d] = 0.f;
      for (int j = 0; j < n; ++j) {
        const float ds = Pr[j] * (Sr[j] - delta);
        Sr[j] = ds;
        const float4* kr = reinterpret_cast<const float4*>(Ks + j * HDP);
#pragma unroll
        for (int d4 = 0; d4 < HDP / 4; ++d4) {
          const float4 kv = kr[d4];
          dq[4 * d4] += ds * kv.x; dq[4 * d4 + 1] += ds * kv.y; dq[4 * d4 + 2] += ds * kv.z; dq[4 * d4 + 3] += ds * kv.w;
        }
        const int jy = j / a.win, jx = j - jy * a.win;
        // one wavefront per workgroup and the lanes of one step hit distinct table entries: plain read-modify-write
        dbias[(iy - jy + a.win - 1) * w2 + (ix - jx + a.win - 1)] += ds;
      }
      store_row(drow, a.hd, dq, a.scale);
    }
    __syncthreads();
    if (act) {  // ---- key phase ----
      float dk[HDP], dv[HDP];
#pragma unroll
      for (int d = 0; d < HDP; ++d) dk[d] = dv[d] = 0.f;
      for (int i = 0; i < n; ++i) {
        const float ds = Sm[i * PLD + lane], p = Pm[i * PLD + lane];
        const float4* qr = reinterpret_cast<const float4*>(Qs + i * HDP);
        const float4* gr = reinterpret_cast<const float4*>(Ds + i * HDP);
#pragma unroll
        for (int d4 = 0; d4 < HDP / 4; ++d4) {
          const float4 qv = qr[d4], gv = gr[d4];
          dk[4 * d4] += ds * qv.x; dk[4 * d4 + 1] += ds * qv.y; dk[4 * d4 + 2] += ds * qv.z; dk[4 * d4 + 3] += ds * qv.w;
          dv[4 * d4] += p * gv.x; dv[4 * d4 + 1] += p * gv.y; dv[4 * d4 + 2] += p * gv.z; dv[4 * d4 + 3] += p * gv.w;
        }
      }
      store_row(drow + a.C, a.hd, dk, a.scale);
      store_row(drow + 2 * a.C, a.hd, dv, 1.0f);
      if (h == 0)  // K padding of the following GEMM operand
        for (int c = 3 * a.C; c < a.lddq; c += 4) *reinterpret_cast<uint2*>(a.dqkv + tok * a.lddq + c) = make_uint2(0u, 0u);
    }
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
  const int nr = round_up(n, 16);
  WbArgs a = {qkv, ldq, dO, ldo, dqkv, lddq, rpb, drpb, B, res, win, shift, heads, C, hd, 1.0f / sqrtf((float)hd), nr};
  const size_t lds = (size_t)(4 * nr * HDP + 2 * nr * PLD + 2 * NBMAX + 64) * sizeof(float);
  const size_t lds_max = (size_t)(4 * 64 * HDP + 2 * 64 * PLD + 2 * NBMAX + 64) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)k_wattn_bwd, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  const int nW = (res / win) * (res / win);
  const int gy = max(1, min(B * nW, cdiv(4096, heads)));
  const double T = (double)B * res * res;
  SwgProfScope ps(KID_WATTN_BWD, 10.0 * n * C * T, T * (3.0 * C * 2 + C * 2 + 3.0 * C * 2), s);
  hipLaunchKernelGGL(k_wattn_bwd, dim3(heads, gy), dim3(64), lds, s, a);
  SWG_LAUNCH_CHECK();
  return 0;
}
