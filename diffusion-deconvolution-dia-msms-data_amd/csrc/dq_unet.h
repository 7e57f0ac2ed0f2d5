// Host-side state of one network instance: plan, device tables and the activation arena layout.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <vector>
#include "dq_plan.h"

namespace dq {

constexpr int TBUF_FLOATS = 100;  // per-sample scratch of the time-embedding kernels (see k_time.hip)

struct DevTables {
  int64_t* ss_w_off = nullptr;  // [ss_total] offset of row r's 16 weights in the flat parameter buffer
  int64_t* ss_b_off = nullptr;  // [ss_total] offset of row r's bias
};

int launch_time_embed_fwd(const Plan& p, const DevTables& dt, const float* params, const int64_t* t, int t_scalar, float* tbuf,
                          float* ss, int B, const int* step_tab, const int* step_ptr, hipStream_t s);
int launch_ss_heads(const float* temb, const float* w, const float* bias, float* ss, int B, int m, hipStream_t s);
int launch_ss_heads_strided(const float* temb, const float* w, const float* bias, float* ss, int stride, int B, int m, hipStream_t s);
int launch_time_mlp_fwd(const float* w1, const float* b1, const float* w2, const float* b2, const int64_t* t, float* tbuf, int B,
                        hipStream_t s);
int launch_time_embed_bwd(const Plan& p, const DevTables& dt, const float* params, float* grads, float* tbuf, const float* dss,
                          int B, hipStream_t s);

// Activation arena: offsets (in floats) of every tensor the forward produces for a given (B, RT).  The backward's
// gradient of a tensor lives at the same offset in a second arena of the same size ("twin").
struct ResBuf {
  int64_t u1, a1, u2, out;
  int64_t gpart = 0, gpart_floats = 0;  // per-block norm-gain sums of the backward (k_res_bwd / k_res_bwd_cp / k_block_bwd)
  int64_t wpart = 0, wpart_floats = 0;  // per-workgroup slots of the backward that forms its weight gradients itself (k_res_bwd_wg); 0: not that path
};
struct LevelBuf {
  ResBuf r0, r1; int64_t la, la_pre, la_tmp, rs;
  int64_t cpart = 0, cpart_floats = 0;  // per-workgroup slots of the resample conv's backward (k_conv_bwd_wg); 0: not that path
};
struct WideResBuf { int64_t u1 = 0, a1 = 0, u2 = 0, out = 0; };  // (B, mid_c, P) each: a ResnetBlock of the wide bottleneck  // la_pre: saved pre-norm LA output; la_tmp: backward scratch (twin only)
struct Arena {
  int B = 0, RT = 0;
  int64_t floats = 0;       // total arena size
  int64_t zero_floats = 0;  // prefix whose gradient twin must be zeroed before a backward (the accumulated-into tensors)
  int64_t tbuf, ss, cat0, ms1n, ms1_u, ms1_a, ms1f, h0;
  std::vector<LevelBuf> downs, ups;
  int64_t mid_in, xn, qv, kk, o, lse, delta, attn_out, mid_back, eps, partials, head_part, loss, coef, xa, xb, wg, wg_floats, la_part, la_part_floats, ts_tab, step, c2_stage, c1_stage, wtmp, la_prep, wimg, timg, bb_part, bb_part_floats, ms1_scratch;
  ResBuf mid1, mid2, fin;
  // wide bottleneck (Plan::wide_mid): P = RT padded to a multiple of 4; every tensor below is (B, channels, P)
  int P = 0;
  WideResBuf wmid1, wmid2;
  int64_t w_mid_in = 0, w_xcol = 0, w_xn = 0, w_qv = 0, w_o = 0, w_attn_out = 0, w_stats = 0, w_gemm_part = 0, w_gemm_part_floats = 0;
};
void layout_arena(const Plan& p, int B, int RT, Arena& a);

}  // namespace dq

struct dq_plan {
  dq::Plan plan;
  dq::DevTables dev;
  float* alpha_bars_dev = nullptr;          // (T) fp32, for q_sample
  std::vector<float> alpha_bars_host;
  dq::Arena arena;                           // cached for the last (B, RT)
  // hipGraph of ONE sampling step (network forward + DDIM update + step counter), replayed num_steps times; valid while
  // every pointer baked into its kernel arguments is unchanged
  hipGraphExec_t step_exec = nullptr;
  hipGraph_t step_graph = nullptr;
  // side stream for the weight-gradient kernels of the backward (forked / joined with events from this ring)
  static constexpr int NUM_EVENTS = 256;
  hipStream_t side_stream = nullptr;
  hipEvent_t events[NUM_EVENTS] = {};
  unsigned ev_next = 0;
  bool side_used = false;
  const void* twin_zeroed = nullptr;  // the gradient twin a forked forward (dq_train_step) cleared on the side stream; consumed by unet_backward
  float* debug_tail_addr = nullptr; float debug_tail_value = 0.f; int debug_tail_us = 0;  // dq_debug_side_tail_store (test hook)
  bool no_side = false;  // dq_plan_set_side_stream(plan, 0): weight-gradient launches on the caller's stream (captured train steps)
  hipStream_t cap_stream = nullptr;  // capture-only stream (the caller's may be the uncapturable legacy default stream)
  const void* g_params = nullptr; const void* g_rope = nullptr; const void* g_ws = nullptr;
  int g_B = 0, g_RT = 0, g_norm = -1, g_pred = -1;
  unsigned g_opt_epoch = 0;  // dq::options_epoch() at capture time (a dq_set_option call may change the dispatch baked into the graph)
};
