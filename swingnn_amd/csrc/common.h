// Shared device helpers for the gfx950 SwinGNN kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/swingnn_hip.h"
#include "prof.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define SWG_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

// Opt-in to more than 64 KiB of dynamic LDS, once per (kernel instantiation, DEVICE): `done` is the call site's
// bit-per-device record (the attribute belongs to the device's copy of the code object, so a second GPU in the same
// process needs its own call; a plain `static bool` would skip it).  Thread-safe; returns 0 or a hipError_t.
#include <atomic>
static inline int ensure_dyn_lds(std::atomic<uint64_t>& done, const void* fn, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  const uint64_t bit = 1ull << (dev & 63);
  if (done.load(std::memory_order_acquire) & bit) return 0;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) return (int)e;
  done.fetch_or(bit, std::memory_order_release);
  return 0;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int round_up(int a, int b) { return cdiv(a, b) * b; }
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }
// byte counts go through this, never through an unqualified max(): in a .hip translation unit max(size_t, size_t) can
// bind to the int overload and silently drops every term above 2^31 (a workspace plan that shrinks as the batch grows)
static inline size_t smax(size_t a, size_t b) { return a > b ? a : b; }
// every pointer 16-byte aligned (NULL counts as aligned): the gate of the float4 forms of the elementwise kernels
template <class... P>
static inline bool aligned16(const P*... p) {
  return ((reinterpret_cast<uintptr_t>(p) | ...) & 15) == 0;
}

// x * sigmoid(x) with the hardware reciprocal and exponential (v_rcp_f32 / v_exp_f32, 1 ulp each).  NOT __frcp_rn:
// hipcc lowers that to the correctly rounded division (v_div_scale x2, v_div_fmas, v_div_fixup, 4 fmas: ~12 VALU
// instructions per element, which made the conditioning prologues the largest VALU item of the fused kernels).
__device__ __forceinline__ float fast_rcp(float v) { return __builtin_amdgcn_rcpf(v); }
__device__ __forceinline__ float silu_f(float v) {
  return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.44269504088896340736f));
}
// 1/sqrt(var + eps) of LayerNorm (v_rsq_f32, 1 ulp)
__device__ __forceinline__ float ln_rstd(float var) { return __frsqrt_rn(var + 1e-5f); }
// nn.GELU() (exact erf form, model/swin_gnn/swin_gnn.py:10,15) evaluated as x * sigmoid(x * P(x^2)) with a
// degree-4 minimax P fitted to logit(Phi(x)):  |abs err| <= 3.5e-6 for every finite x (far below the bf16
// rounding of the value it feeds), saturating correctly for |x| -> inf.  One v_exp, one v_rcp and eight
// full-rate ops, no branches: the fused MLP kernels are VALU-bound on this function.
// The coefficients carry the factor -log2(e) so that the exponential is a bare exp2.
__device__ __forceinline__ float gelu_f(float x) {
  const float s = x * x;
  float q = fmaf(-3.228973355362541e-06f, s, 8.823775715427473e-05f);
  q = fmaf(q, s, 0.00036027710302732885f);
  q = fmaf(q, s, -0.10522669553756714f);
  q = fmaf(q, s, -2.3020453453063965f);
  const float e = __builtin_amdgcn_exp2f(x * q);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

// sum over a group of LANES consecutive lanes (LANES power of two <= 64)
// Sum over an aligned group of LANES lanes; every lane ends with the total.  Groups of 16 / 32 / 64 lanes use the gfx950
// row-swap instructions and DPP row rotations (VALU only) in the SAME combination order as the xor butterfly below
// (32, 16 across rows, then 8, 4, 2, 1 inside a row), so the results are bit-identical to it; a ds_bpermute per step
// costs an LDS latency each, ten of them per LayerNorm row.
template <int LANES>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (LANES >= 16) {
    if constexpr (LANES == 64) {
      auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    }
    if constexpr (LANES >= 32) {
      auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
      v = __uint_as_float(b[0]) + __uint_as_float(b[1]);
    }
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x121, 0xf, 0xf, false));  // row_ror:1
    return v;
  } else {
#pragma unroll
    for (int o = LANES / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
  }
}

// Reductions over the four 16-lane rows of a wavefront (lanes l, l^16, l^32, l^48) with the gfx950 row-swap
// instructions: v_permlane{16,32}_swap exchange register rows in the VALU, no LDS round trip (ds_bpermute costs an
// LDS latency per step, which one wave per SIMD cannot hide).  Every lane ends with the full result.
__device__ __forceinline__ float rows_sum(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15: the 16 tokens of a lane group in the token-on-lane layouts):
// four v_add with a row-rotate DPP modifier, no LDS crossbar.  Every lane ends with the total.
__device__ __forceinline__ float row16_sum(float v) {
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xf, 0xf, false));  // row_ror:8
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x124, 0xf, 0xf, false));  // row_ror:4
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x122, 0xf, 0xf, false));  // row_ror:2
  v += __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x121, 0xf, 0xf, false));  // row_ror:1
  return v;
}
__device__ __forceinline__ float rows_max(float v) {
  auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  v = __builtin_fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __builtin_fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

__device__ __forceinline__ uint2 pack_bf16x4(float a, float b, float c, float d) {
  bf16x4 v = {(bf16)a, (bf16)b, (bf16)c, (bf16)d};
  return *reinterpret_cast<uint2*>(&v);
}

// LDS-DMA of 16 bytes per lane (global_load_lds_dwordx4): LDS destination = wave-uniform byte address + lane*16.
// Issued through inline asm ON PURPOSE: hipcc tracks the builtin form as a pending LDS write and then puts
// `s_waitcnt vmcnt(0)` in front of any ds_read it cannot prove disjoint from the destination (e.g. a bias vector in
// another part of LDS), which exposes the whole transfer latency.  The caller orders the data itself:
// `s_waitcnt vmcnt(0)` by every issuing wave, then a workgroup barrier, then the ds_reads.
__device__ __forceinline__ void lds_dma16(const void* gsrc_lane, unsigned lds_byte_addr_uniform) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_byte_addr_uniform);
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc_lane), "s"(dst)
      : "memory");
}
// Same transfer with a WAVE-UNIFORM source base (SGPR pair) + a per-lane 32-bit byte offset (lane * 16 for a linear
// 1-KiB piece): the address arithmetic of a piece stays on the scalar unit (the per-lane form costs a 64-bit VALU add,
// a v_readfirstlane and often v_accvgpr reads of spilled constants per piece -- issue slots the MFMA groups need).
__device__ __forceinline__ void lds_dma16_u(const void* gsrc_uniform, unsigned lane_byte_off, unsigned lds_byte_addr_uniform) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(lane_byte_off), "s"(gsrc_uniform), "s"(lds_byte_addr_uniform)
      : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) void*)p;
}

// Diagnostic build only (-DSWG_STAMPS on ONE translation unit, scripts/build_diag.sh): s_memtime stamps of chosen
// waves go to a buffer of their own that no kernel reads; scripts/diag_stamps.py prints the phase timeline.  The
// kernel defines `dbg` (bool) and `dbgslot` (0..3) and calls STAMP(i), i < 64.  The shipped library is built without
// it: STAMP() expands to nothing.
#ifdef SWG_STAMPS
__device__ unsigned long long g_stamp[4 * 64];
#define STAMP(i)                                                           \
  do {                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                     \
    if (dbg) {                                                             \
      unsigned long long t_ = __builtin_amdgcn_s_memtime();                \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   \
      if ((threadIdx.x & 63) == 0) g_stamp[dbgslot * 64 + (i)] = t_;       \
    }                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                     \
  } while (0)
extern "C" int swg_debug_stamps(unsigned long long* host_out, int n) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamp), sizeof(unsigned long long) * n);
}
#else
#define STAMP(i)
#endif

// ---- row-kernel launch geometry: a row of `chunks` float4 is handled by LANES lanes x V chunks ----
struct RowGeom {
  int lanes, v;
};
static inline RowGeom row_geom(int chunks) {
  RowGeom g;
  int lanes = 4;
  while (lanes < 64 && cdiv(chunks, lanes) > 3) lanes *= 2;
  g.lanes = lanes;
  g.v = cdiv(chunks, lanes);
  return g;
}

// dispatch on (lanes, v); V in {1,2,3} for lanes < 64, {1,2,3,4,6,8} for lanes == 64
#define SWG_ROW_DISPATCH(G, CALL)                                                       \
  do {                                                                                  \
    const int L__ = (G).lanes, V__ = (G).v;                                             \
    if (L__ == 4 && V__ == 1) { CALL(4, 1); }                                           \
    else if (L__ == 4 && V__ == 2) { CALL(4, 2); }                                      \
    else if (L__ == 4 && V__ == 3) { CALL(4, 3); }                                      \
    else if (L__ == 8 && V__ == 2) { CALL(8, 2); }                                      \
    else if (L__ == 8 && V__ == 3) { CALL(8, 3); }                                      \
    else if (L__ == 16 && V__ == 2) { CALL(16, 2); }                                    \
    else if (L__ == 16 && V__ == 3) { CALL(16, 3); }                                    \
    else if (L__ == 32 && V__ == 2) { CALL(32, 2); }                                    \
    else if (L__ == 32 && V__ == 3) { CALL(32, 3); }                                    \
    else if (L__ == 64 && V__ == 2) { CALL(64, 2); }                                    \
    else if (L__ == 64 && V__ == 3) { CALL(64, 3); }                                    \
    else if (L__ == 64 && V__ == 4) { CALL(64, 4); }                                    \
    else if (L__ == 64 && (V__ == 5 || V__ == 6)) { CALL(64, 6); }                      \
    else if (L__ == 64 && (V__ == 7 || V__ == 8)) { CALL(64, 8); }                      \
    else return SWG_E_UNSUPPORTED;                                                      \
  } while (0)

// ---- internal launchers (defined in the .hip files, used by api.hip) ---------------------------
// rowops.hip
int launch_cond_ln(const float* x, int rows, int C, int L, const float* cond, int cond_stride, int cond_off,
                   const float* g, const float* b, float* x1, bf16* xn, int ldn, hipStream_t s);
int launch_merge_ln(const float* x, int B, int res, int C, const float* g, const float* b, bf16* xn, int ldn,
                    hipStream_t s);
int launch_concat_cast(const float* a, const float* b, int rows, int Ch, bf16* out, int ldo, hipStream_t s);
int launch_ln_f32(const float* x, int rows, int C, const float* g, const float* b, float* y, hipStream_t s);
int launch_breakup_ln(const float* y, int B, int res, int Co, const float* g, const float* b, bf16* out, int ldo,
                      hipStream_t s);
int launch_breakup_ln2(const float* y, int B, int res, int Cin, const float* g1, const float* b1, const float* g2,
                       const float* b2, bf16* out, int ldo, hipStream_t s, float* z_out = nullptr);
int launch_gemv_rows(const float* W, const float* bias, const float* x, int Bx, int N, int K, int act, float* out,
                     hipStream_t s);
int launch_pos_embed(const float* labels, int Bc, const float* freqs, int nf, float* out, hipStream_t s);
int launch_edm_coef(const float* labels, int B, int uniform, int mode, float* coef, float* c_noise, hipStream_t s);
int launch_patch_embed(const swg_net* net, int B, const float* x, const float* sc, const float* c_in, int c_in_stride,
                       const float* cond, int cond_stride, float* tok, hipStream_t s);
int launch_final_readout(const swg_net* net, int B, const float* tok, float* F, hipStream_t s);
int launch_sym_precond(int B, int N, const float* F, const float* x, const float* flags, const float* coef, float* out,
                       float* sym_err, hipStream_t s);
int launch_sym_precond_ch(int B, int Ca, int N, const float* F, const float* x, const float* flags, const float* coef,
                          float* out, float* sym_err, hipStream_t s);
// nodeadj.hip : head and tail of the molecule (node + adjacency) network, model/swin_gnn/swin_gnn.py:960-1021
int launch_na_embed(const swg_net* net, const swg_nodeadj* na, int B, const float* adj, const float* node,
                    const float* sc_adj, const float* sc_node, const float* flags, const float* c_in, int c_in_stride,
                    const float* cond, int cond_stride, float* tok, hipStream_t s);
int launch_na_head_adj(const bf16* h, int ldh, int B, int N, int p, int C, int Ca, const float* w2, const float* b2,
                       float* F, hipStream_t s);
int launch_na_head_node(const bf16* h, int ldh, int B, int N, int C, int Cn, const float* w2, const float* b2,
                        const float* node, const float* flags, const float* coef, float* out, hipStream_t s);
int launch_na_pool(const bf16* rep, int ldr, int B, int N, int p, const float* flags, bf16* pooled, hipStream_t s);
int launch_na_im2col(const swg_net* net, const swg_nodeadj* na, int B, const float* adj, const float* node,
                     const float* sc_adj, const float* sc_node, const float* flags, const float* c_in, int c_in_stride,
                     bf16* out, int Kp, hipStream_t s);
int launch_na_dadj(const float* dD, const float* flags, const float* coef, int B, int N, int p, int Ca, float* dy,
                   hipStream_t s);
int launch_na_dnode(const float* dD, const float* flags, const float* coef, int B, int N, int Cn, float* dy,
                    hipStream_t s);
int launch_na_head_bwd(const float* dy, const float* w2, int rows, int C, int Co, bf16* dact, int ld, hipStream_t s);
int launch_na_pool_bwd(const float* dpooled, const float* flags, int B, int N, int p, int C, float* drep, hipStream_t s);
// gemm.hip : out[M,N] = A[M,Kp] @ W[N,Kp]^T (+bias)(gelu)(+residual)
struct GemmEpi {
  const float* bias;      // [N] or null
  const float* residual;  // fp32 [M, ldr] or null
  int ldr;
  // optional conditioning of the residual: r <- silu(shift + r (1 + scale)) with scale = cond[m / cond_L][cond_off + n],
  // shift = cond[m / cond_L][cond_off + N + n]  (the block shortcut of swin_gnn.py:238-242, recomputed in the epilogue)
  const float* cond;
  int cond_stride, cond_off, cond_L;
  int act;                // 0 none, 1 gelu
  float* out_f32;         // fp32 [M, ldo] or null
  bf16* out_bf16;         // bf16 [M, ldo] or null; columns [N, n_store) are written as zero
  int ldo;
  int n_store;
};
int launch_gemm(const bf16* A, int lda, const bf16* W, int M, int N, int Kp, const GemmEpi& e, hipStream_t s);
// weight-gradient form: out[M,N] (dense fp32) += A[M,Kp] @ W[N,Kp]^T, reduction split over the grid into `partial`
size_t gemm_splitk_ws_floats(int M, int N);
int launch_gemm_splitk(const bf16* A, int lda, const bf16* W, int ldw, int M, int N, int Kp, float* out, float* partial,
                       size_t partial_floats, hipStream_t s);
// weight-gradient form on token-major operands (no transposed copies): out[Mo,No] += sum_t P[t,m] Q[t,n]
int launch_gemm_tn_splitk(const bf16* P, int ldp, const bf16* Q, int ldq, int rows, int Mo, int No, float* out,
                          float* partial, size_t partial_floats, hipStream_t s, float* colsum_p = nullptr);
// bwd_rows.hip : memory-bound kernels of the backward pass
int launch_cast_pad(const float* x, int rows, int C, int ldx, bf16* out, int ldo, hipStream_t s);
int launch_transpose_f32(const float* src, int M, int N, int ld, bf16* dst, int ldd, hipStream_t s);
int launch_transpose_bf16(const bf16* src, int M, int N, int ld, bf16* dst, int ldd, hipStream_t s);
int launch_colsum_f32(const float* src, int M, int N, int ld, float* out, hipStream_t s);
int launch_colsum_bf16(const bf16* src, int M, int N, int ld, float* out, hipStream_t s);
int launch_gelu_fwd(const bf16* h, bf16* a, size_t n, hipStream_t s);
int launch_gelu_bwd(const bf16* h, const bf16* da, bf16* dh, size_t n, hipStream_t s);
int launch_cast_pad_colsum(const float* x, int rows, int C, int ldx, bf16* out, int ldo, float* db, hipStream_t s);
bool mlp_bwd_front_supported(int C);
int launch_mlp_bwd_front(int C, int T, const float* x, const float* dy, const bf16* img, const float* b1, const float* g,
                         const float* b, bf16* xn, bf16* dyb, int ldc, bf16* act, bf16* dh, int ldh, float* part,
                         hipStream_t s);
size_t mlp_bwd_front_part_floats(int C, int T);
int mlp_bwd_front_tokens_per_wg(int C);
bool ln_cond_bwd_supported(int C);
int launch_ln_cond_bwd(const float* x, const float* dxn, const float* dres, const float* g, const float* cond,
                       int cond_stride, int cond_off, int B, int L, int C, float* dx, float* dg, float* db, float* dcond,
                       int dcond_stride, hipStream_t s);
int launch_gelu_bwd_colsum(const bf16* h, const bf16* da, bf16* dh, bf16* a_out, int rows, int N, int ld, float* db,
                           hipStream_t s);
int launch_ln_bwd(const float* x, const float* dy, const float* g, const float* dres, int rows, int C, float* dx,
                  float* dg, float* db, hipStream_t s, int perm_res = 0, int perm_mask = 0,
                  bf16* dx_bf16 = nullptr);
int launch_cond_bwd(const float* x, const float* dxc, const float* cond, int cond_stride, int cond_off, int B, int L, int C,
                    float* dx, float* dcond, int dcond_stride, hipStream_t s);
int launch_perm22(const float* src, float* dst, int B, int res, int C, int dir, hipStream_t s);
int launch_embed_im2col(const swg_net* net, int B, const float* x, const float* sc, const float* c_in, int c_in_stride,
                        bf16* out, int Kp, hipStream_t s);
int launch_readout_dpix(const float* dD, const float* flags, const float* coef, int B, int N, int p, bf16* out, int Kp,
                        hipStream_t s);
int launch_add_slice(const float* tmp, int N, int K, int ldt, float* grad, hipStream_t s);
// wattn_bwd.hip : backward of the window attention core
int launch_wattn_bwd(const bf16* qkv, int ldq, const bf16* dO, int ldo, int B, int res, int win, int shift, int heads,
                     int C, const float* rpb, bf16* dqkv, int lddq, float* drpb, hipStream_t s);
// wattn.hip : window attention core on a qkv buffer
int launch_wattn_core(const bf16* qkv, int ldq, int B, int res, int win, int shift, int heads, int C, const float* rpb,
                      bf16* o, int ldo, hipStream_t s);
// wattn_res.hip : the same in one pass for C = 96 with all weights resident in LDS (persistent 8-wave workgroups)
bool wattn_res_supported(int C, int heads, int win, int res);
int launch_wattn_res(const swg_block* blk, int B, const float* x, const float* cond, int cond_stride, float* out,
                     hipStream_t s);
// wattn_ring.hip : conditioning + LN1 + QKV + window attention (+ proj + shortcut for C = 96 / 192) on a weight
// fragment ring; C = 384 writes o (bf16, pi slots) for the projection GEMM
bool wattn_ring_supported(int C, int heads, int win, int res);
bool wattn_ring_fuses_proj(int C);
bool wattn_ring_wants_xn(int C);
int launch_wattn_ring(const swg_block* blk, int B, const float* x, const bf16* xn, int ldxn, const float* cond,
                      int cond_stride, float* out, bf16* o, int ldo, hipStream_t s);
// embed_mfma.hip : MFMA forms of patch embedding and final LN + read-out for p = 4, C0 = 96
bool embed_mfma_supported(const swg_net* net);
int launch_patch_embed_mfma(const swg_net* net, int B, const float* x, const float* sc, const float* c_in,
                            int c_in_stride, const float* cond, int cond_stride, float* tok, hipStream_t s);
int launch_final_readout_mfma(const swg_net* net, int B, const float* tok, float* F, hipStream_t s);
// PatchBreakup tail (norm + scatter + post_norm + post_linear) in one pass, Co = 96
bool breakup_post_supported(int Co);
int launch_breakup_post(const float* y, int B, int res, int Co, const float* g1, const float* b1, const float* g2,
                        const float* b2, const bf16* w, int ldw, float* out, hipStream_t s);
// mlp_fused.hip : x_out = x + fc2(GELU(fc1(LN2(x)))) in one pass (C = 96 / 192)
bool mlp_fused_supported(int C);
int launch_mlp_fused(int C, int T, const float* x, float* out, const bf16* img, const float* b1, const float* b2,
                     const float* g, const float* b, hipStream_t s);
// merge_fused.hip : PatchMerging (2x2 gather + LayerNorm(4C) + Linear(4C -> 2C)) in one pass, C = 96 / 192
bool merge_fused_supported(int C, int res);
int launch_merge_fused(int C, int B, int res, const float* x, const bf16* img, const float* g, const float* b, float* out,
                       hipStream_t s);
// expand_fused.hip : concat + PatchBreakup (pre_linear, norm, 2x2 scatter, post_norm, post_linear) in one pass, Cin = 384
bool expand_fused_supported(int Cin);
int launch_expand_fused(int Cin, int B, int res, const float* x, const float* skip, const bf16* img_pre,
                        const bf16* img_post, const float* g1, const float* b1, const float* g2, const float* b2,
                        float* out, hipStream_t s);
// mlp_stream.hip : the same for the wide stages (C = 384), LDS ring + skewed GELU pipeline, fragment-order image
bool mlp_stream_supported(int C);
// PROJ form of k_mlp_fused (head_dim-20 blocks, C = 60 / 120): attention tail + MLP half in one pass
bool mlp_fused_proj_supported(int C);
int launch_mlp_fused_proj(int C, int T, int L, const float* x_in, const bf16* o, int ldo, const float* cond, int cond_stride,
                          int cond_off, const float* proj_b, float* out, const bf16* pimg, const bf16* img, const float* b1,
                          const float* b2, const float* g, const float* b, hipStream_t s);
bool mlp_stream_proj_supported(int C);
int launch_mlp_stream_proj(int C, int T, int L, const float* x_in, const bf16* o, int ldo, const float* cond, int cond_stride,
                           int cond_off, const float* proj_b, float* out, const bf16* img, const float* b1,
                           const float* b2, const float* g, const float* b, hipStream_t s);
int launch_mlp_stream(int C, int T, const float* x, float* out, const bf16* img, const float* b1, const float* b2,
                      const float* g, const float* b, hipStream_t s);
