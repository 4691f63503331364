// Attribute codecs of the molecule path on the device (utils/attribute_code.py:14-329 of the reference): the
// int <-> ddpm / one-hot / bits encodings of node [B,N(,C)] and adjacency [B,(C,)N,N] attributes, and the
// post-sampling quantisation of runner/sampler/sampler_node_adj.py:178-200.  Pure HBM-bound element work: one
// thread per (graph, node) or (graph, row, col) entry walks the C channels; channel planes are read with the entry
// index fastest (coalesced for the [B,C,N,N] adjacency layout).  Integer results are exact.
#include "common.h"

namespace {

struct CodecArgs {
  const float* in;
  const float* flags;  // [B, N] node flags, 0 / 1
  float* out;
  const float* lo;     // ddpm -> int: lower / upper interval bounds [K] (fp32 casts of the reference's doubles)
  const float* hi;
  int B, N, C, K;
  int adj;             // 1: adjacency entries [B,(C,)N,N] (channel stride N*N); 0: node entries [B,N(,C)] (stride 1)
  int ddpm_range;      // one-hot / bits: channels are -1 / +1 (else 0 / 1)
  int binarise;        // decode of a raw sample: clamp to [-1,1], then x > 0 -> +1 else -1 (sampler_node_adj.py:179-181)
  int clamp_int;       // bits -> int: clamp to [0, K-1] (attribute_code.py:203-204)
};

__device__ __forceinline__ bool entry_valid(const CodecArgs& a, int b, int e, int& base, int& cstride) {
  if (a.adj) {
    const int r = e / a.N, c = e - r * a.N;
    base = b * a.C * a.N * a.N + e;
    cstride = a.N * a.N;
    return a.flags[b * a.N + r] != 0.f && a.flags[b * a.N + c] != 0.f;
  }
  base = (b * a.N + e) * a.C;
  cstride = 1;
  return a.flags[b * a.N + e] != 0.f;
}

// KIND 0: ddpm -> int (:124-177)   KIND 1: one-hot -> int (:213-238)   KIND 2: bits -> int (:180-210)
template <int KIND>
__global__ __launch_bounds__(256) void k_attr_decode(CodecArgs a) {
  const int per = a.adj ? a.N * a.N : a.N;
  const int e = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (e >= per) return;
  int base, cs;
  const bool ok = entry_valid(a, b, e, base, cs);
  float r = 0.f;
  if (KIND == 0) {
    float x = a.in[a.adj ? b * per + e : b * a.N + e];
    if (a.binarise) x = x < -1.0f ? -1.0f : (x > 1.0f ? 1.0f : x);  // torch.clamp: a NaN stays a NaN
    float q = -1.0f;  // torch.full_like(in, -1.0): what a NaN keeps (:154)
    for (int i = 0; i < a.K; ++i)
      if (x > a.lo[i] && x <= a.hi[i]) q = (float)i;
    r = ok ? q : 0.f;
  } else {
    float best = 0.f;
    int arg = 0;
    float acc = 0.f;
    for (int c = 0; c < a.C; ++c) {
      float x = a.in[base + c * cs];
      if (a.binarise) x = (fminf(fmaxf(x, -1.0f), 1.0f) > 0.0f) ? 1.0f : -1.0f;
      float v = a.ddpm_range ? (x + 1.0f) * 0.5f : x;  // (:196, :227)
      if (a.ddpm_range && !ok) v = 0.f;                // masked after the rescale (:197, :228)
      if (KIND == 1) {
        if (c == 0 || v > best) { best = v; arg = c; }  // first maximum, as torch.argmax
      } else {
        acc += (float)(1 << (a.C - 1 - c)) * v;         // bin2dec (:320-329), most significant bit first
      }
    }
    r = (KIND == 1) ? (float)arg : acc;
    if (KIND == 2 && a.clamp_int) r = fminf(fmaxf(r, 0.f), (float)(a.K - 1));
    if (!ok) r = 0.f;
  }
  a.out[b * per + e] = r;
}

// KIND 0: int -> ddpm (:241-255)   KIND 1: int -> one-hot (:284-305)   KIND 2: int -> bits (:258-281)
template <int KIND>
__global__ __launch_bounds__(256) void k_attr_encode(CodecArgs a) {
  const int per = a.adj ? a.N * a.N : a.N;
  const int e = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (e >= per) return;
  int base, cs;
  const bool ok = entry_valid(a, b, e, base, cs);
  const float xi = a.in[b * per + e];
  if (KIND == 0) {
    a.out[b * per + e] = ok ? 2.0f * xi / ((float)a.K - 1.0f) - 1.0f : 0.f;
    return;
  }
  const int i = (int)xi;
  for (int c = 0; c < a.C; ++c) {
    float v = (KIND == 1) ? (i == c ? 1.0f : 0.0f) : (((i >> (a.C - 1 - c)) & 1) ? 1.0f : 0.0f);
    if (a.ddpm_range) v = 2.0f * v - 1.0f;
    a.out[base + c * cs] = ok ? v : 0.f;
  }
}

int check(const CodecArgs& a) {
  if (!a.in || !a.flags || !a.out) return SWG_E_NULL;
  if (a.B <= 0 || a.N <= 0 || a.C <= 0 || a.K < 2 || a.C > 4096) return SWG_E_SHAPE;
  return 0;
}
// bits: the integer must stay exact in fp32 and in the shift
int check_bits(int encoding, int C) { return (encoding == 2 && C > 24) ? SWG_E_SHAPE : 0; }

}  // namespace

extern "C" {

int swg_attr_decode(int32_t encoding, int32_t B, int32_t N, int32_t C, int32_t num_types, int32_t is_adj,
                    int32_t ddpm_range, int32_t binarise, int32_t clamp_int, const float* in, const float* node_flags,
                    const float* lo, const float* hi, float* out_int, void* stream) {
  CodecArgs a = {in, node_flags, out_int, lo, hi, B, N, C, num_types, is_adj, ddpm_range, binarise, clamp_int};
  if (int e = check(a)) return e;
  if (int e = check_bits(encoding, C)) return e;
  if (encoding == 0 && (!lo || !hi || C != 1)) return lo && hi ? SWG_E_SHAPE : SWG_E_NULL;
  const int per = is_adj ? N * N : N;
  const dim3 grid(cdiv(per, 256), B);
  hipStream_t s = (hipStream_t)stream;
  if (encoding == 0) hipLaunchKernelGGL(k_attr_decode<0>, grid, dim3(256), 0, s, a);
  else if (encoding == 1) hipLaunchKernelGGL(k_attr_decode<1>, grid, dim3(256), 0, s, a);
  else if (encoding == 2) hipLaunchKernelGGL(k_attr_decode<2>, grid, dim3(256), 0, s, a);
  else return SWG_E_UNSUPPORTED;
  SWG_LAUNCH_CHECK();
  return 0;
}

int swg_attr_encode(int32_t encoding, int32_t B, int32_t N, int32_t C, int32_t num_types, int32_t is_adj,
                    int32_t ddpm_range, const float* in_int, const float* node_flags, float* out, void* stream) {
  CodecArgs a = {in_int, node_flags, out, nullptr, nullptr, B, N, C, num_types, is_adj, ddpm_range, 0, 0};
  if (int e = check(a)) return e;
  if (int e = check_bits(encoding, C)) return e;
  if (encoding == 0 && C != 1) return SWG_E_SHAPE;
  const int per = is_adj ? N * N : N;
  const dim3 grid(cdiv(per, 256), B);
  hipStream_t s = (hipStream_t)stream;
  if (encoding == 0) hipLaunchKernelGGL(k_attr_encode<0>, grid, dim3(256), 0, s, a);
  else if (encoding == 1) hipLaunchKernelGGL(k_attr_encode<1>, grid, dim3(256), 0, s, a);
  else if (encoding == 2) hipLaunchKernelGGL(k_attr_encode<2>, grid, dim3(256), 0, s, a);
  else return SWG_E_UNSUPPORTED;
  SWG_LAUNCH_CHECK();
  return 0;
}

}  // extern "C"
