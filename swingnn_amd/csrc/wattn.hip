// Window-attention core (model/swin_gnn/swin_gnn.py:108-136 without the two Linear layers) on a packed
// qkv buffer in TOKEN order.  window_partition / window_reverse (:28-57) and the cyclic shift (:247-270)
// never materialise: a window's token list is computed from (window, shift) and used as the gather list
// for Q/K/V and as the scatter list for the output.  The relative-position bias column of the head sits
// in LDS; the shift mask (:207-226) is evaluated in-register from region ids (no [nW,n,n] mask tensor).
//
// One wavefront per (graph, window, head).  S^T = K Q^T is computed with the key on the accumulator rows
// and the query on the lane, so the softmax reduction over keys is in-lane plus two xor-shuffles, and the
// exponentiated tile is directly the B operand of O^T = V^T P^T (k-slot permutation matched on the V side).
#include "common.h"

namespace {

struct WattnArgs {
  const bf16* qkv; int ldq;
  int B, res, win, shift, heads, C, hd;
  const float* rpb;
  bf16* o; int ldo;
  float scale;
};

constexpr int VS_LD = 34;  // bf16 elements per V row in LDS (68 B: breaks the 4-way conflict of 64-B rows)

struct WaveLds {
  int tok[64];
  int rid[64];
  float bias[228];
  bf16 v[64 * VS_LD];
};

template <bool FAST>
__device__ __forceinline__ bf16x8 load_frag(const bf16* p, int d0, int hd) {
  if (FAST) return *reinterpret_cast<const bf16x8*>(p + d0);
  uint2 lo = make_uint2(0u, 0u), hi = make_uint2(0u, 0u);
  if (d0 < hd) lo = *reinterpret_cast<const uint2*>(p + d0);
  if (d0 + 4 < hd) hi = *reinterpret_cast<const uint2*>(p + d0 + 4);
  uint4 r = make_uint4(lo.x, lo.y, hi.x, hi.y);
  return *reinterpret_cast<bf16x8*>(&r);
}

template <int NT, bool FAST>
__global__ __launch_bounds__(256) void k_wattn_core(WattnArgs a) {
  constexpr int KS = (NT + 1) / 2;  // 32-key steps of the PV product
  __shared__ __attribute__((aligned(16))) WaveLds lds[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l15 = lane & 15, lg = lane >> 4;
  WaveLds& L = lds[wave];

  const int nwx = a.res / a.win;
  const int nW = nwx * nwx;
  const int total = a.B * nW * a.heads;
  int u = blockIdx.x * 4 + wave;
  const bool live = u < total;
  u = min(u, total - 1);
  const int h = u % a.heads;
  const int wi = (u / a.heads) % nW;
  const int b = u / (a.heads * nW);
  const int wy = wi / nwx, wx = wi - wy * nwx;
  const int n = a.win * a.win;
  const int Ltok = a.res * a.res;

  // ---- gather list + region ids ----
  {
    const int i = min(lane, n - 1);
    const int iy = i / a.win, ix = i - iy * a.win;
    const int ys = wy * a.win + iy, xs = wx * a.win + ix;
    int y = ys + a.shift, x = xs + a.shift;  // shifted[y] = x[(y+s) mod H]  (torch.roll by -s)
    if (y >= a.res) y -= a.res;
    if (x >= a.res) x -= a.res;
    L.tok[lane] = b * Ltok + y * a.res + x;
    int rid = 0;
    if (a.shift > 0) {
      const int ry = ys < a.res - a.win ? 0 : (ys < a.res - a.shift ? 1 : 2);
      const int rx = xs < a.res - a.win ? 0 : (xs < a.res - a.shift ? 1 : 2);
      rid = 3 * ry + rx;
    }
    L.rid[lane] = rid;
    const int nb = (2 * a.win - 1) * (2 * a.win - 1);
    for (int t = lane; t < nb; t += 64) L.bias[t] = a.rpb[(size_t)t * a.heads + h];
  }
  __syncthreads();

  // ---- Q / K fragments straight from global (16 B per lane), V rows into LDS ----
  bf16x8 qf[NT], kf[NT];
  const int hc = h * a.hd;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const bf16* row = a.qkv + (size_t)L.tok[t * 16 + l15] * a.ldq;
    qf[t] = load_frag<FAST>(row + hc, 8 * lg, a.hd);
    kf[t] = load_frag<FAST>(row + a.C + hc, 8 * lg, a.hd);
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int item = lane + 64 * i;
    const int key = item >> 2, part = item & 3;
    bf16x8 vv = {};
    if (key < n) vv = load_frag<FAST>(a.qkv + (size_t)L.tok[key] * a.ldq + 2 * a.C + hc, 8 * part, a.hd);
    uint32_t* dst = reinterpret_cast<uint32_t*>(&L.v[key * VS_LD + 8 * part]);
    const uint4 raw = *reinterpret_cast<uint4*>(&vv);
    dst[0] = raw.x; dst[1] = raw.y; dst[2] = raw.z; dst[3] = raw.w;
  }

  // ---- S^T = K Q^T : acc[kt][qt], lane: query qt*16+l15, keys kt*16 + 4*lg + r ----
  float p[NT][NT][4];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int qt = 0; qt < NT; ++qt) {
      f32x4 z = {0.f, 0.f, 0.f, 0.f};
      z = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[kt], qf[qt], z, 0, 0, 0);
      p[kt][qt][0] = z[0]; p[kt][qt][1] = z[1]; p[kt][qt][2] = z[2]; p[kt][qt][3] = z[3];
    }

  // key-side coordinates of this lane's accumulator rows
  int jy[NT][4], jx[NT][4], jr[NT][4];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = kt * 16 + 4 * lg + r;
      const int jc = min(j, n - 1);
      jy[kt][r] = jc / a.win;
      jx[kt][r] = jc - jy[kt][r] * a.win;
      jr[kt][r] = (j < n) ? L.rid[jc] : -1;
    }

  float inv_sum[NT];
  const int w2 = 2 * a.win - 1;
#pragma unroll
  for (int qt = 0; qt < NT; ++qt) {
    const int i = min(qt * 16 + l15, n - 1);
    const int iy = i / a.win, ix = i - iy * a.win;
    const int ri = L.rid[i];
    float mx = -3.0e38f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = -3.0e38f;
        if (jr[kt][r] >= 0) {
          s = p[kt][qt][r] * a.scale + L.bias[(iy - jy[kt][r] + a.win - 1) * w2 + (ix - jx[kt][r] + a.win - 1)];
          if (a.shift > 0 && jr[kt][r] != ri) s += -100.0f;  // swin_gnn.py:226
        }
        p[kt][qt][r] = s;
        mx = fmaxf(mx, s);
      }
    mx = rows_max(mx);  // over lanes l, l^16, l^32, l^48 (row swaps in the VALU instead of two ds_bpermute)
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = (jr[kt][r] >= 0) ? __expf(p[kt][qt][r] - mx) : 0.f;
        p[kt][qt][r] = e;
        sum += e;
      }
    {  // same combination order as the xor-16, xor-32 butterfly it replaces
      auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
      sum = __uint_as_float(b[0]) + __uint_as_float(b[1]);
      auto c = __builtin_amdgcn_permlane32_swap(__float_as_uint(sum), __float_as_uint(sum), false, false);
      sum = __uint_as_float(c[0]) + __uint_as_float(c[1]);
    }
    inv_sum[qt] = fast_rcp(sum);
  }

  __syncthreads();  // V rows visible

  // ---- O^T = V^T P^T : the k-slot j of lane group lg is key 32*ks + (j<4 ? 4*lg+j : 16+4*lg+j-4) on both sides ----
  bf16x8 vf[2][KS];
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int key = 32 * ks + (j < 4 ? 4 * lg + j : 16 + 4 * lg + (j - 4));
        f[j] = L.v[key * VS_LD + dt * 16 + l15];
      }
      vf[dt][ks] = f;
    }

#pragma unroll
  for (int qt = 0; qt < NT; ++qt) {
    bf16x8 pf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8 f;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f[j] = (bf16)p[2 * ks][qt][j];
        f[4 + j] = (2 * ks + 1 < NT) ? (bf16)p[(2 * ks + 1 < NT) ? 2 * ks + 1 : 0][qt][j] : (bf16)0.f;
      }
      pf[ks] = f;
    }
    const int i = qt * 16 + l15;
    const bool row_ok = live && i < n;
    bf16* orow = a.o + (size_t)L.tok[min(i, n - 1)] * a.ldo;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
      f32x4 oacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) oacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[dt][ks], pf[ks], oacc, 0, 0, 0);
      const int d0 = dt * 16 + 4 * lg;
      if (row_ok && d0 < a.hd) {
        const float s = inv_sum[qt];
        *reinterpret_cast<uint2*>(orow + hc + d0) = pack_bf16x4(oacc[0] * s, oacc[1] * s, oacc[2] * s, oacc[3] * s);
      }
    }
  }

  // zero the K-padding columns [C, ldo) of the rows this window owns (head 0 only)
  if (live && h == 0 && a.ldo > a.C && lane < n) {
    bf16* orow = a.o + (size_t)L.tok[lane] * a.ldo;
    for (int c = a.C; c < a.ldo; c += 4) *reinterpret_cast<uint2*>(orow + c) = make_uint2(0u, 0u);
  }
}

}  // namespace

int launch_wattn_core(const bf16* qkv, int ldq, int B, int res, int win, int shift, int heads, int C, const float* rpb,
                      bf16* o, int ldo, hipStream_t s) {
  if (win <= 0 || res % win || C % heads) return SWG_E_SHAPE;
  const int n = win * win, hd = C / heads;
  if (n > 64 || hd > 32 || hd % 4 || ldq % 4 || ldo % 4 || C % 4) return SWG_E_UNSUPPORTED;
  if (shift < 0 || shift >= win) return SWG_E_SHAPE;
  WattnArgs a = {qkv, ldq, B, res, win, shift, heads, C, hd, rpb, o, ldo, 1.0f / sqrtf((float)hd)};
  const int nW = (res / win) * (res / win);
  const int total = B * nW * heads;
  const int NT = cdiv(n, 16);
  const bool fast = (hd == 32) && (ldq % 8 == 0) && (C % 8 == 0);
  dim3 grid(cdiv(total, 4)), block(256);
  const double T = (double)B * res * res;
  SwgProfScope ps(KID_WATTN_CORE, 4.0 * n * C * T, T * (3.0 * C * 2 + C * 2), s);
#define WCALL(NTv)                                                                              \
  do {                                                                                          \
    if (fast) hipLaunchKernelGGL((k_wattn_core<NTv, true>), grid, block, 0, s, a);              \
    else hipLaunchKernelGGL((k_wattn_core<NTv, false>), grid, block, 0, s, a);                  \
  } while (0)
  switch (NT) {
    case 1: WCALL(1); break;
    case 2: WCALL(2); break;
    case 3: WCALL(3); break;
    case 4: WCALL(4); break;
    default: return SWG_E_UNSUPPORTED;
  }
#undef WCALL
  SWG_LAUNCH_CHECK();
  return 0;
}
