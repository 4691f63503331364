// Optimizer tail of the train step (runner/trainer/trainer.py:136-140) as two multi-tensor kernels over every
// parameter of the model: (1) the global L2 gradient norm of clip_grad_norm_(max_norm, norm_type=2), (2) gradient
// scaling by the clip coefficient + Adam (utils/learning_utils.py:245-248: betas, eps, L2 weight decay) + the
// exponential moving averages of the parameters (:253-274), one read of p / g / m / v and of each EMA copy, one
// write of p / m / v and each EMA copy.  HBM-bound: 4 B x (4 reads + 3 writes + 2 per EMA) per parameter.
//
// Parameters and gradients stay the caller's separate tensors (ordinary nn.Parameters); they are addressed through
// device pointer tables.  Work is cut into chunks of CHUNK elements; chunk c belongs to tensor chunk_tensor[c] and
// starts at element chunk_off[c].  Moments and EMA copies live in flat buffers at flat_off[tensor].
#include "common.h"

namespace {

constexpr int CHUNK = 2048;  // elements per workgroup: 256 threads x 2 float4

// persistent workgroups over the chunk list (one atomic per WORKGROUP, not per chunk: 17 k same-address atomics were
// 3/4 of this kernel's time), float4 loads where the chunk is 16-byte aligned
__global__ __launch_bounds__(256) void k_grad_sqnorm(const float* const* __restrict__ g_ptrs,
                                                     const int32_t* __restrict__ numel,
                                                     const int32_t* __restrict__ chunk_tensor,
                                                     const int32_t* __restrict__ chunk_off, int n_chunks,
                                                     float* __restrict__ sqnorm) {
  __shared__ float red[4];
  float s = 0.f;
  for (int c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const int t = chunk_tensor[c];
    const int off = chunk_off[c];
    const int n = min(CHUNK, numel[t] - off);
    const float* g = g_ptrs[t] + off;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
      const int n4 = n >> 2;
      for (int i = threadIdx.x; i < n4; i += 256) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
      }
      for (int i = 4 * n4 + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    } else {
      for (int i = threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    }
  }
  s = group_sum<64>(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(sqnorm, red[0] + red[1] + red[2] + red[3]);
}

struct AdamArgs {
  float* const* p_ptrs;
  const float* const* g_ptrs;
  const int32_t* numel;
  const int64_t* flat_off;
  const int32_t* chunk_tensor;
  const int32_t* chunk_off;
  float* m;
  float* v;
  float* ema;
  int64_t total;
  int n_ema;
  float ema_w[8];  // 1 - decay_k (1 = copy)
  const float* sqnorm;
  float max_norm, lr, b1, b2, eps, wd;
  float bc1, bc2s;  // 1 - b1^t, sqrt(1 - b2^t)
};

__global__ __launch_bounds__(256) void k_adam_ema(AdamArgs a) {
  const int c = blockIdx.x;
  const int t = a.chunk_tensor[c];
  const int off = a.chunk_off[c];
  const int n = min(CHUNK, a.numel[t] - off);
  float* p = a.p_ptrs[t] + off;
  const float* g = a.g_ptrs[t] + off;
  const int64_t fo = a.flat_off[t] + off;
  float clip = 1.f;
  if (a.sqnorm) {  // torch.nn.utils.clip_grad_norm_: coef = max_norm / (norm + 1e-6), clamped to 1
    clip = fminf(a.max_norm / (sqrtf(*a.sqnorm) + 1e-6f), 1.0f);
  }
  const float step = a.lr / a.bc1;
  auto update = [&](float& pv, float gv, float& mv, float& vv) {
    gv *= clip;
    if (a.wd != 0.f) gv = fmaf(a.wd, pv, gv);
    mv = a.b1 * mv + (1.f - a.b1) * gv;
    vv = a.b2 * vv + (1.f - a.b2) * gv * gv;
    pv -= step * mv / (sqrtf(vv) / a.bc2s + a.eps);
  };
  int done = 0;
  // 16-byte form (same arithmetic per element): parameter, gradient and the flat moment / EMA slices all aligned
  if (((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(a.m + fo) |
        reinterpret_cast<uintptr_t>(a.v + fo) | reinterpret_cast<uintptr_t>(a.ema ? a.ema + fo : nullptr)) & 15) == 0 &&
      (a.total & 3) == 0) {
    const int n4 = n >> 2;
    for (int i = threadIdx.x; i < n4; i += 256) {
      float4 pv = reinterpret_cast<float4*>(p)[i];
      const float4 gv = reinterpret_cast<const float4*>(g)[i];
      float4 mv = reinterpret_cast<float4*>(a.m + fo)[i];
      float4 vv = reinterpret_cast<float4*>(a.v + fo)[i];
      update(pv.x, gv.x, mv.x, vv.x);
      update(pv.y, gv.y, mv.y, vv.y);
      update(pv.z, gv.z, mv.z, vv.z);
      update(pv.w, gv.w, mv.w, vv.w);
      reinterpret_cast<float4*>(a.m + fo)[i] = mv;
      reinterpret_cast<float4*>(a.v + fo)[i] = vv;
      reinterpret_cast<float4*>(p)[i] = pv;
      for (int k = 0; k < a.n_ema; ++k) {
        float4* e = reinterpret_cast<float4*>(a.ema + (int64_t)k * a.total + fo) + i;
        float4 ev = *e;
        ev.x += a.ema_w[k] * (pv.x - ev.x);
        ev.y += a.ema_w[k] * (pv.y - ev.y);
        ev.z += a.ema_w[k] * (pv.z - ev.z);
        ev.w += a.ema_w[k] * (pv.w - ev.w);
        *e = ev;
      }
    }
    done = 4 * n4;
  }
  for (int i = done + threadIdx.x; i < n; i += 256) {
    float pv = p[i], mv = a.m[fo + i], vv = a.v[fo + i];
    update(pv, g[i], mv, vv);
    a.m[fo + i] = mv;
    a.v[fo + i] = vv;
    p[i] = pv;
    for (int k = 0; k < a.n_ema; ++k) {
      float* e = a.ema + (int64_t)k * a.total + fo + i;
      const float ev = *e;
      *e = ev + a.ema_w[k] * (pv - ev);  // lerp(ema, p, 1 - decay)
    }
  }
}

}  // namespace

extern "C" {

int swg_grad_sqnorm(const void* g_ptrs, const int32_t* numel, const int32_t* chunk_tensor, const int32_t* chunk_off,
                    int32_t n_chunks, float* sqnorm, void* stream) {
  if (!g_ptrs || !numel || !chunk_tensor || !chunk_off || !sqnorm) return SWG_E_NULL;
  if (n_chunks <= 0) return SWG_E_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  hipError_t he = hipMemsetAsync(sqnorm, 0, sizeof(float), s);
  if (he != hipSuccess) return (int)he;
  SwgProfScope ps(KID_OPTIM, 0.0, 4.0 * n_chunks * (double)CHUNK, s);
  hipLaunchKernelGGL(k_grad_sqnorm, dim3(min(n_chunks, 2048)), dim3(256), 0, s, (const float* const*)g_ptrs, numel,
                     chunk_tensor, chunk_off, n_chunks, sqnorm);
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_adam_ema_step(const void* p_ptrs, const void* g_ptrs, const int32_t* numel, const int64_t* flat_off,
                      const int32_t* chunk_tensor, const int32_t* chunk_off, int32_t n_chunks, float* m, float* v,
                      float* ema, int64_t total, int32_t n_ema, const float* ema_decay, const float* sqnorm,
                      float max_norm, float lr, float beta1, float beta2, float eps, float weight_decay, int32_t step,
                      void* stream) {
  if (!p_ptrs || !g_ptrs || !numel || !flat_off || !chunk_tensor || !chunk_off || !m || !v) return SWG_E_NULL;
  if (n_chunks <= 0 || step < 1 || n_ema < 0 || n_ema > 8 || total <= 0) return SWG_E_SHAPE;
  if (n_ema > 0 && (!ema || !ema_decay)) return SWG_E_NULL;
  AdamArgs a = {};
  a.p_ptrs = (float* const*)p_ptrs; a.g_ptrs = (const float* const*)g_ptrs;
  a.numel = numel; a.flat_off = flat_off; a.chunk_tensor = chunk_tensor; a.chunk_off = chunk_off;
  a.m = m; a.v = v; a.ema = ema; a.total = total; a.n_ema = n_ema;
  for (int k = 0; k < n_ema; ++k) a.ema_w[k] = 1.0f - ema_decay[k];
  a.sqnorm = sqnorm; a.max_norm = max_norm; a.lr = lr; a.b1 = beta1; a.b2 = beta2; a.eps = eps; a.wd = weight_decay;
  a.bc1 = (float)(1.0 - pow((double)beta1, (double)step));
  a.bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)step));
  hipStream_t s = (hipStream_t)stream;
  SwgProfScope ps(KID_OPTIM, 0.0, 4.0 * n_chunks * (double)CHUNK * (7.0 + 2.0 * n_ema), s);
  hipLaunchKernelGGL(k_adam_ema, dim3(n_chunks), dim3(256), 0, s, a);
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_optim_chunk(void) { return CHUNK; }

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// Refresh of the DERIVED weight operands after an optimizer step (no counterpart in the reference: the reference's
// kernels read the fp32 nn.Parameters directly).  One launch rewrites every bf16 operand the kernels consume from the
// fp32 parameters: K-padded copies, transposed copies (backward) and the fragment-order images of the fused kernels
// (gather through a precomputed index map).  Table-driven like the optimizer kernels above.
// ------------------------------------------------------------------------------------------------------------------
namespace {

__global__ __launch_bounds__(256) void k_refresh_derived(const swg_refresh_item* __restrict__ items,
                                                         const int32_t* __restrict__ chunk_item,
                                                         const int32_t* __restrict__ chunk_c0,
                                                         const int32_t* __restrict__ chunk_c1) {
  __shared__ float tile[64][65];
  const swg_refresh_item it = items[chunk_item[blockIdx.x]];
  const int c0 = chunk_c0[blockIdx.x], c1 = chunk_c1[blockIdx.x];
  __bf16* dst = reinterpret_cast<__bf16*>(it.dst);
  if (it.mode == 2) {  // dst[k, n] (row stride ld) = src[n, k], zero for n >= N; tile (c0: k block, c1: n block)
    const int k0 = c0 * 64, n0 = c1 * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int j = ty; j < 64; j += 4) {
      const int n = n0 + j, k = k0 + tx;
      tile[j][tx] = (n < it.N && k < it.K) ? it.src0[(size_t)n * it.K + k] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
      const int k = k0 + i, n = n0 + tx;
      if (k < it.K && n < it.ld) dst[(size_t)k * it.ld + n] = (__bf16)tile[tx][i];
    }
    return;
  }
  const int64_t base = (int64_t)c0 * 16384;
  const int64_t end = base + 16384 < it.count ? base + 16384 : it.count;
  if (it.mode == 1) {  // dst[n, k] (row stride ld) = src[n, k], zero for k >= K
    for (int64_t i = base + threadIdx.x; i < end; i += 256) {
      const int n = (int)(i / it.ld), k = (int)(i - (int64_t)n * it.ld);
      dst[i] = (__bf16)(k < it.K ? it.src0[(size_t)n * it.K + k] : 0.f);
    }
    return;
  }
  // mode 0: gather; idx value v: 0 = zero, otherwise element v-1 of the concatenation (src0 [n0] | src1)
  for (int64_t i = base + threadIdx.x; i < end; i += 256) {
    const int v = it.idx[i];
    float x = 0.f;
    if (v > 0) x = (v - 1 < it.n0) ? it.src0[v - 1] : it.src1[v - 1 - it.n0];
    dst[i] = (__bf16)x;
  }
}

}  // namespace

extern "C" int swg_refresh_derived(const swg_refresh_item* items, const int32_t* chunk_item, const int32_t* chunk_c0,
                                   const int32_t* chunk_c1, int32_t n_chunks, void* stream) {
  if (n_chunks <= 0) return 0;
  if (!items || !chunk_item || !chunk_c0 || !chunk_c1) return SWG_E_NULL;
  hipLaunchKernelGGL(k_refresh_derived, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, items, chunk_item, chunk_c0,
                     chunk_c1);
  SWG_LAUNCH_CHECK();
  return 0;
}
