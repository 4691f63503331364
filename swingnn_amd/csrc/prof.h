// Optional per-kernel profiler: when enabled, every internal launcher brackets its launch with a pair
// of HIP events on the launch stream and records the algorithmic FLOPs / bytes of that launch.
// Off by default (one global flag test per launch).  Must be off during HIP graph capture.
#pragma once
#include <hip/hip_runtime.h>

enum SwgKernelId {
  KID_GEMM = 0,
  KID_WATTN_CORE,
  KID_ROW_COND_LN,
  KID_ROW_MERGE_LN,
  KID_ROW_CONCAT,
  KID_ROW_LN_F32,
  KID_ROW_BREAKUP_LN,
  KID_GEMV,
  KID_POS_EMBED,
  KID_EDM_COEF,
  KID_PATCH_EMBED,
  KID_READOUT,
  KID_SYM_PRECOND,
  KID_HEUN,
  KID_FUSED_WATTN,
  KID_FUSED_MLP,
  KID_MLP_STREAM,
  KID_WATTN_RING,
  KID_WATTN_RES,
  KID_BREAKUP_POST,
  KID_BWD_ROWS,
  KID_WATTN_BWD,
  KID_OPTIM,
  KID_NODEADJ,
  // one id per kernel SYMBOL where a launcher picks between instantiations (bench.py's roofline names one symbol)
  KID_GEMM_NT64,
  KID_GEMM_NT_BIG,
  KID_GEMM_TN,
  KID_MLP_STREAM_384,
  KID_WATTN_RING_384,
  KID_MLP_PROJ_384,
  KID_MERGE_FUSED,
  KID_EXPAND_FUSED,
  KID_MLP_PROJ_192,
  KID_WATTN_RING_HD20,
  KID_COUNT
};

struct SwgProfScope {
  int slot;
  hipStream_t s;
  SwgProfScope(int kid, double flops, double bytes, hipStream_t stream);
  ~SwgProfScope();
};
