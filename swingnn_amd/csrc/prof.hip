// Per-kernel profiler behind swg_prof_* (measurement aid; no counterpart in the reference).
#include "prof.h"

#include <mutex>
#include <vector>

#include "../../include/swingnn_hip.h"

namespace {

struct Rec {
  int kid;
  double flops, bytes;
  hipEvent_t a, b;
};

struct Prof {
  bool on = false;
  std::vector<Rec> recs;       // event pool, reused
  size_t used = 0;
  double ms[KID_COUNT] = {}, flops[KID_COUNT] = {}, bytes[KID_COUNT] = {};
  long launches[KID_COUNT] = {};
  std::mutex mu;
} P;

const char* kNames[KID_COUNT] = {"k_gemm_nt<32>",    "k_wattn_core",  "k_row<cond_ln>", "k_row<merge_ln>",
                                 "k_row<concat>",  "k_row<ln_f32>", "k_row<breakup_ln>", "k_gemv_rows",
                                 "k_pos_embed",    "k_edm_coef",    "k_patch_embed",  "k_final_readout",
                                 "k_sym_precond",  "k_heun",        "k_wattn_fused",  "k_mlp_fused",
                                 "k_mlp_stream<192>", "k_wattn_ring<192>", "k_wattn_res",   "k_breakup_post",
                                 "bwd_rows",       "k_wattn_bwd",   "k_adam_ema",
                                 "k_nodeadj",      "k_gemm_nt<64>", "k_gemm_nt_big",  "k_gemm_tn",
                                 "k_mlp_stream<384>", "k_wattn_ring<384>", "k_mlp_stream<384,proj>", "k_merge_linear", "k_expand_fused",
                                 "k_mlp_stream<192,proj>", "k_wattn_ring<hd20>"};

void drain() {  // caller holds the lock; synchronises the recorded events
  for (size_t i = 0; i < P.used; ++i) {
    Rec& r = P.recs[i];
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    P.ms[r.kid] += t;
    P.flops[r.kid] += r.flops;
    P.bytes[r.kid] += r.bytes;
    P.launches[r.kid] += 1;
  }
  P.used = 0;
}

}  // namespace

SwgProfScope::SwgProfScope(int kid, double flops, double bytes, hipStream_t stream) : slot(-1), s(stream) {
  if (!P.on) return;
  std::lock_guard<std::mutex> g(P.mu);
  if (P.used == P.recs.size()) {
    if (P.recs.size() >= 65536) drain();
    else {
      Rec r = {};
      if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
      P.recs.push_back(r);
    }
  }
  Rec& r = P.recs[P.used];
  r.kid = kid;
  r.flops = flops;
  r.bytes = bytes;
  slot = (int)P.used++;
  (void)hipEventRecord(r.a, s);
}

SwgProfScope::~SwgProfScope() {
  if (slot < 0) return;
  std::lock_guard<std::mutex> g(P.mu);
  (void)hipEventRecord(P.recs[slot].b, s);
}

extern "C" {

int swg_prof_enable(int32_t on) {
  std::lock_guard<std::mutex> g(P.mu);
  if (!on && P.on) drain();
  P.on = on != 0;
  return 0;
}

int swg_prof_reset(void) {
  std::lock_guard<std::mutex> g(P.mu);
  drain();
  for (int k = 0; k < KID_COUNT; ++k) P.ms[k] = P.flops[k] = P.bytes[k] = 0.0, P.launches[k] = 0;
  return 0;
}

int swg_prof_kernels(void) { return KID_COUNT; }

const char* swg_prof_name(int32_t kid) { return (kid >= 0 && kid < KID_COUNT) ? kNames[kid] : ""; }

int swg_prof_read(int32_t kid, double* ms, double* flops, double* bytes, int64_t* launches) {
  if (kid < 0 || kid >= KID_COUNT) return SWG_E_SHAPE;
  std::lock_guard<std::mutex> g(P.mu);
  drain();
  if (ms) *ms = P.ms[kid];
  if (flops) *flops = P.flops[kid];
  if (bytes) *bytes = P.bytes[kid];
  if (launches) *launches = P.launches[kid];
  return 0;
}

}  // extern "C"
