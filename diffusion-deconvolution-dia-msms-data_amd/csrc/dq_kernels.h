// Internal launch API of libdq_hip.so (host side).  Every launcher is asynchronous on the given stream,
// allocates nothing and returns 0 on success (non-zero after dq::set_error()).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dq {

constexpr int MSE_MAX_BLOCKS = 1024;  // size of the partial-sum scratch used by the loss / grad-norm reductions

// ---- k_stream.hip
int launch_q_sample(const float* alpha_bars, const float* x0, const int64_t* t, const float* noise, float* x_t, int B,
                    int64_t per_sample, int normalize, hipStream_t s);
int launch_ddim_step(const float* x_t, const float* eps, float* x_prev, const float* coef_dev, int64_t n, const int* step_ptr,
                     hipStream_t s, int pred_x0 = 0, float* eps_out = nullptr);
// x_prev may alias x_t (element-wise); step_ptr (nullable): row index into coef_dev; pred_x0: ``eps`` is the network's x0
// prediction and the derived eps goes to eps_out (nullable)
int launch_inc_step(int* p, hipStream_t s);
int launch_sample_finish(const float* x, const float* ms2_cond, float* out_x, float* out_noise, int64_t n, int normalize,
                         hipStream_t s);
int launch_mse_fwd_bwd(const float* eps, const float* noise, float* loss_out, float* grad_out, float* partials, int64_t n,
                       hipStream_t s, const float* lw = nullptr, const int64_t* t = nullptr, int64_t per_sample = 0, float tm = 1.f,
                       float ta = 0.f, int* defer_sum = nullptr);  // lw: per-timestep loss weights (x0 objective); target' = target*tm + ta
// defer_sum: the kernel leaves its per-block partials, *defer_sum receives their count and loss_out is NOT written -- the caller runs
// launch_sum_partials(partials, count, 1 / n, loss_out, stream) when it suits (dq_train_step: on the side stream)
int launch_sum_partials(const float* partials, int count, float scale, float* out, hipStream_t s);
// the MS1 term of train_step with ms1_loss_weight = w > 0 (k_stream.hip): on entry loss_out / grad hold the MSE part, on exit the
// combined loss (1 - w) * MSE + w * additional and its gradient w.r.t. the network output; x_t null = x0 objective;
// scratch: 5 * B * RT + B floats
int launch_ms1_loss(const float* out, const float* x_t, const float* ms1, float cm, float ca, const float* lw, const int64_t* t, float w,
                    int B, int RT, int MZ, float* grad, float* loss_out, float* scratch, hipStream_t s);
int launch_adamw_clip(float* p, const float* g, float* m, float* v, int64_t n, float* partials, float gscale, float max_norm,
                      double lr, double b1, double b2, double eps, double wd, int step, float* gnorm_out, hipStream_t s);

// the same with lr and the step count in device memory (graph replay): increments *step_dev, then uses it
int launch_adamw_clip_dev(float* p, const float* g, float* m, float* v, int64_t n, float* partials, float gscale, float max_norm,
                          const float* lr_dev, double b1, double b2, double eps, double wd, int* step_dev, float* gnorm_out, hipStream_t s);

int launch_axpy(float* dst, const float* src, int64_t n, hipStream_t s);  // dst += src
int launch_zero(float* dst, int64_t n, hipStream_t s);  // dst = 0 (a kernel, not a memset node)
int launch_copy(float* dst, const float* src, int64_t n, hipStream_t s);  // dst = src (a kernel, not a memcpy node)

// ---- k_conv.hip
enum ConvMode { CONV_S1 = 0, CONV_DOWN = 1, CONV_UP = 2 };  // stride-1 'same' | k4 s2 p1 | nearest x2 then k3 p1
enum Act { ACT_NONE = 0, ACT_SILU = 1, ACT_GELU = 2 };

struct ConvFwd {
  // input = channel-concat of A (cinA) and B (cinB); tensors are (rows, C, n) with n contiguous
  const float* inA = nullptr; const float* inB = nullptr; int cinA = 0, cinB = 0;
  const float* w = nullptr;     // (cout, cinA+cinB, K)
  const float* bias = nullptr;  // (cout) or null
  int cout = 0, K = 1, mode = CONV_S1;
  int rows = 0, n_in = 0, n_out = 0;
  float* u_out = nullptr;       // pre-norm conv output (saved for backward) or null
  float* y_out = nullptr;
  const float* g = nullptr;     // RMSNorm gain (cout) or null => no norm
  const float* ss = nullptr;    // per-sample [scale(cout) | shift(cout)] or null
  int ss_stride = 0, rows_per_sample = 1;
  int act = ACT_NONE;
  // residual added after the activation: identity (resA has cout channels, res_w null) or 1x1 conv over cat(resA,resB)
  const float* resA = nullptr; const float* resB = nullptr; int rcinA = 0, rcinB = 0;
  const float* res_w = nullptr; const float* res_b = nullptr;
};
int launch_conv_fwd(const ConvFwd& a, hipStream_t s);

// pointwise backward of [RMSNorm -> scale/shift -> act] (any of them optional): du from (u, dy)
struct BlockBwd {
  const float* u = nullptr; const float* dy = nullptr; float* du = nullptr;
  int C = 0, rows = 0, n = 0, rows_per_sample = 1;
  const float* g = nullptr; float* dg = nullptr;                           // norm gain and its grad (+=)
  const float* ss = nullptr; float* dss = nullptr; int ss_stride = 0;     // per-sample scale/shift and grads (+=)
  int act = ACT_NONE;
  float* dbias = nullptr;  // optional: sum of du over rows and positions (+=)
  int accumulate = 0;      // 1: du += instead of du =
  const float* add_src = nullptr;  // accumulate: the addend is read from HERE instead of du (du = add_src + d; du itself is only written)
  // per-block partial sums [dg | dscale | dshift | dbias] (needed when any of dg / dss / dbias is set): >= 64 * groups * 4 C
  // floats; the launcher sums them in block order right behind the kernel (no float atomics: repeatable to the bit)
  float* part = nullptr; int64_t part_floats = 0;
  struct PartReduce* defer_reduce = nullptr;  // nullable: the launcher hands the reduction of `part` back instead of launching it (the caller runs it, e.g. on a side stream)
};
int launch_block_bwd(const BlockBwd& a, hipStream_t s);

// ordered sums of per-block partials part[(b * gx + x) * nv + i] (b < B groups, x < gx blocks per group): up to three "global"
// segments [start, start + len) -> dst[j] += sum over every block, and one per-group segment [s0, s0 + sn) ->
// sdst[b * sstride + j] += sum over the group's blocks.  Fixed summation order.
struct PartReduce {
  const float* part = nullptr; int B = 0, gx = 0, nv = 0;
  int nseg = 0; int seg_start[3] = {0, 0, 0}, seg_len[3] = {0, 0, 0}; float* seg_dst[3] = {nullptr, nullptr, nullptr};
  int s0 = 0, sn = 0; float* sdst = nullptr; int sstride = 0;
};
int launch_part_reduce(const PartReduce& a, hipStream_t s);

// dX (+=) of a conv: dinA/dinB receive the gradient of the concat input (either may be null => skipped)
struct ConvBwdData {
  const float* du = nullptr; const float* w = nullptr;
  int cout = 0, K = 1, mode = CONV_S1, rows = 0, n_in = 0, n_out = 0;
  float* dinA = nullptr; float* dinB = nullptr; int cinA = 0, cinB = 0;
  int accumulate = 1;  // 1: += ; 0: =
};
int launch_conv_bwd_data(const ConvBwdData& a, hipStream_t s);

// dW (+=) and dbias (+=; optional) of a conv: per-block partials + an ordered reduce (no atomics)
struct ConvWgrad {
  const float* du = nullptr; const float* inA = nullptr; const float* inB = nullptr; int cinA = 0, cinB = 0;
  int cout = 0, K = 1, mode = CONV_S1, rows = 0, n_in = 0, n_out = 0;
  float* dw = nullptr; float* dbias = nullptr;
  float* scratch = nullptr; int64_t scratch_floats = 0;  // >= WGRAD_MAX_PARTS * (cout*cin*K + cout) floats
};
constexpr int WGRAD_MAX_PARTS = 512;
int launch_conv_wgrad(const ConvWgrad& a, hipStream_t s);
// up to three stride-1 convs over the same (rows, n) in one launch + one merged reduce (each with its own scratch region)
int launch_conv_wgrad_multi(const ConvWgrad* w, int count, hipStream_t s);

// ---- k_res.hip : fused ResnetBlock over m/z rows whose length divides 256
struct ResFwd {
  const float* inA = nullptr; const float* inB = nullptr; int cinA = 0, cinB = 0;  // input = cat(A, B)
  const float* w1 = nullptr; const float* b1 = nullptr; const float* g1 = nullptr;   // block1 conv (C, cin, 3), bias, norm gain
  const float* w2 = nullptr; const float* b2 = nullptr; const float* g2 = nullptr;   // block2 conv (C, C, 3)
  const float* wr = nullptr; const float* br = nullptr;                              // res_conv (C, cin) or null => identity
  const float* ss = nullptr; int ss_stride = 0;                                      // per-sample [scale(C) | shift(C)] of block1
  float* u1 = nullptr; float* a1 = nullptr; float* u2 = nullptr;                     // saved for the backward (nullable)
  float* out = nullptr;
  int C = 0, rows = 0, n = 0, rows_per_sample = 1;
};
struct ResBwd {
  const float* dout = nullptr; const float* u1 = nullptr; const float* u2 = nullptr;
  const float* w1 = nullptr; const float* w2 = nullptr; const float* wr = nullptr;
  const float* g1 = nullptr; const float* g2 = nullptr; const float* ss = nullptr; int ss_stride = 0;
  float* du1 = nullptr; float* du2 = nullptr;          // written (the weight-gradient kernels read them)
  float* dA = nullptr; float* dB = nullptr; int cinA = 0, cinB = 0;  // += gradient of the block input (nullable)
  // dA_store / dB_store: this launch is the FIRST writer of that gradient tensor in the backward pass -> plain store, the old
  // contents (zeros) are not read (13 MB per tensor and launch at the wide levels)
  int dA_store = 0, dB_store = 0;
  float* dg1 = nullptr; float* dg2 = nullptr; float* dss = nullptr;  // destinations of the ordered sums (res_part_reduce)
  // gpart: every block's sums go to gpart[block][4 C] = [dg2 | dg1 | dscale | dshift] (block = sample * blocks_per_sample + x)
  // instead of float atomics (2 C of them per block on one cache line from 3,200 blocks cost 0.17 ms per step, and atomics
  // are not repeatable); *gblocks receives the blocks per sample; launch_part_reduce (res_part_reduce) adds the ordered sums
  // to dg2 / dg1 / dss
  float* gpart = nullptr; int64_t gpart_floats = 0; int* gblocks = nullptr;
  int C = 0, rows = 0, n = 0, rows_per_sample = 1;
};
bool res_fusable(int n, int C, int rows_per_sample);  // rows_per_sample == 1: the bottleneck (one RT row per sample: <= 512 positions, any length at 16 channels)
// k_res_rt.hip: the bottleneck's 16-channel blocks (one RT row per sample, identity residual, no skip input) with the RT position as the lane
// column of v_mfma_f32_16x16x4; launch_res_fwd / _bwd dispatch to it
bool res_rt_usable(int C, int cinA, int cinB, bool has_wr, int rows_per_sample);
// the front of the bottleneck's Residual(PreNorm(Attention)) behind the first block: xn = RMSNorm(out) gn -> qv = W_qv xn (B, 256, RT), RoPE on
// q; k = W_k ms1f (B, 128, RT) + RoPE when kk is given
struct ResRtQkv {
  const float* gn = nullptr; const float* wqv = nullptr; float* xn = nullptr; float* qv = nullptr; const float* rope = nullptr;
  const float* wk = nullptr; const float* ms1f = nullptr; float* kk = nullptr;
};
// the back of the attention with the SECOND block: forward -- the block's input attn_out = res + W_o o + b_o is formed in the launch (and written
// to `out`); backward -- d o = W_o^T d attn_out follows the block's d x
struct ResRtOut {
  const float* o = nullptr; const float* w = nullptr; const float* b = nullptr; const float* res = nullptr; float* out = nullptr;  // forward
  float* d_o = nullptr;                                                                                                              // backward
};
int launch_res_rt_fwd(const ResFwd& a, hipStream_t s, const ResRtQkv* q = nullptr, const ResRtOut* ao = nullptr);
// the back of that front in front of the first block's backward: d out = add + PreNorm^T(W_qv^T RoPE^T(dqv)); d gn partials to
// gn_part[(sample * gblocks + workgroup) * 16 + c] (launch_part_reduce)
struct ResRtPre {
  const float* dqv = nullptr; const float* wqv = nullptr; const float* x = nullptr; const float* gn = nullptr; const float* add = nullptr;
  const float* rope = nullptr; float* gn_part = nullptr; int64_t gn_part_floats = 0;
};
int launch_res_rt_bwd(const ResBwd& a, hipStream_t s, const ResRtPre* q = nullptr, const ResRtOut* ao = nullptr);
// k_res_cp.hip: channel-parallel variant for the deep levels (n <= 8, C = 12 / 16); launch_res_fwd / _bwd dispatch to it
bool res_cp_usable(int n, int C, int cinA, int cinB);
int launch_res_fwd_cp(const ResFwd& a, hipStream_t s);
int launch_res_bwd_cp(const ResBwd& a, hipStream_t s);
// k_res_rows.hip: the deep levels' backward data path with the m/z row as the lane column of v_mfma_f32_16x16x4 (12 / 16 channels, rows of
// 2 / 4 / 8 positions, 16-byte aligned tensors); launch_res_bwd dispatches to it
bool res_rows_bwd_usable(const ResBwd& a);
int launch_res_rows_bwd(const ResBwd& a, hipStream_t s);
// k_res_v4.hip: 4-positions-per-thread forward for the wide levels (C = 4 / 8, n >= 8)
bool res_v4_usable(int n, int C, int cinA, int cinB);
int launch_res_fwd_v4(const ResFwd& a, hipStream_t s);
// k_res_mm.hip: forward with the convolutions on the 4x4x1 matrix pipe (rows of 1..64 positions, C = 4 / 8 / 12 / 16)
bool res_mm_usable(int n, int C, int cinA, int cinB, int rows_per_sample, bool has_wr);
int launch_res_fwd_mm(const ResFwd& a, hipStream_t s);
enum LevelPre { LEVEL_PRE_NONE = 0, LEVEL_PRE_DOWN = 1, LEVEL_PRE_UP = 2, LEVEL_PRE_S1 = 3, LEVEL_PRE_INIT = 4 };
// k_level.hip: [resample conv that produces the level's input] -> ResnetBlock (-> ResnetBlock) in ONE launch, convolutions on the
// matrix pipe (rows of 1..64 positions).  blk[i].inA / cinA are unused (a block's first input is in registers); blk[i].out == null:
// that block's output is not written (inference, up path); pre_out: where the input stage's result is kept (training) or null.
struct LevelFwd {
  const float* in = nullptr; int cp = 0;                    // stage input (rows, cp, n_in); NONE: the first block's input (rows, C, n)
  const float* pw = nullptr; const float* pb = nullptr;     // stage conv weight (C, cp, K) and bias (C)
  const float* params = nullptr;                            // the flat parameter buffer every weight / bias / gain pointer points into
  float* pre_out = nullptr;
  int pre = LEVEL_PRE_NONE, nblocks = 1;
  ResFwd blk[2];
  // LEVEL_PRE_INIT (C == 4, inference): in = x_t (rows, n); cond = the mixture (rows, n), normalised as cond * cm + ca and conditioned with
  // init_cond_proj's per-sample [scale, shift] at ss_init (same buffer / stride as the blocks' vectors); pw / pb = init_conv (4, 2, 7);
  // pre_out = h0 (always written)
  const float* cond = nullptr; float cm = 1.f, ca = 0.f; const float* ss_init = nullptr;
  // ... in a train step the stage also stores its two input channels [conditioned mixture | x_t] as cat0 (rows, 2, n): the init conv's weight
  // gradient and the input affine's backward read them (what k_prep_inputs + k_conv_fwd<4,7,0> produced in two launches in front of this one)
  float* cat0_out = nullptr;
  // ... and may form x_t itself (model.py:349-352, k_q_sample's arithmetic): in = x0 (rows, n), qs_noise (rows, n), qs_ab = the alpha-bar table,
  // qs_t = the samples' timesteps, qs_norm: x0 -> 2 x0 - 1 first
  const float* qs_noise = nullptr; const float* qs_ab = nullptr; const int64_t* qs_t = nullptr; int qs_norm = 0;
  // head epilogue (C == 4, inference; behind the last block): eps = final_conv(out) (ew (1, 4, 1), eb) -> eps_out (nullable), and with
  // x_t set the DDIM update of model.py:265-289 into x_out (coef: [sa, sb, sap, sbp] rows; step_ptr nullable: row index on the device)
  const float* ew = nullptr; const float* eb = nullptr; float* eps_out = nullptr;
  const float* x_t = nullptr; float* x_out = nullptr; const float* coef = nullptr; const int* step_ptr = nullptr; int pred_x0 = 0;
  // training head (with ew / eb; the block's output is still written): loss_z = the regression target (rows, n).  d = eps - z; the squared-error
  // sums go to loss_part, one float per WAVE of the grid (*loss_parts_out receives their number: sum them in index order); grad_out (rows, n) =
  // d * loss_gscale (= d loss / d eps) and dout (rows, 4, n) = ew[c] * grad_out (final_conv's backward data path): what k_conv_fwd<1,1,0>,
  // k_mse_fwd_bwd and k_conv_bwd_data<4,1,0> did in three launches behind this one
  const float* loss_z = nullptr; float* loss_part = nullptr; float* grad_out = nullptr; float* dout = nullptr; float loss_gscale = 0.f;
  int* loss_parts_out = nullptr;
  int C = 0, rows = 0, n = 0, rows_per_sample = 1;
  // nullable: this launch's MFMA operand image (level_img_floats floats, 16-byte aligned) as launch_level_images built it from the SAME
  // parameter values -- the kernel's workgroups then copy it to LDS instead of gathering it from the parameter tensors themselves
  const float* img = nullptr;
};
bool level_fwd_usable(int C, int n, int rows_per_sample, int pre_mode, int cp, int nblocks, const ResFwd* blk);
int launch_level_fwd(const LevelFwd& a, hipStream_t s);
constexpr int LEVEL_LOSS_PARTS = 8192;     // floats behind LevelFwd::loss_part (one per wave of a resident round: 6 x 256 workgroups x 4)
constexpr int LEVEL_IMG_MAX = 20;          // launches per launch_level_images call
constexpr int LEVEL_IMG_FLOATS = 8192;     // upper bound of level_img_floats over the built instantiations (16 channels, 32-channel blocks)
int64_t level_img_floats(const LevelFwd& a);
int launch_level_images(const LevelFwd* calls, int count, hipStream_t s);  // writes calls[i].img (must be set) for every call, ONE launch
int launch_res_fwd(const ResFwd& a, hipStream_t s);
int launch_res_bwd(const ResBwd& a, hipStream_t s);

// k_tiny.hip: the levels with rows of 1 or 2 positions as a chain of dense layers on v_mfma_f32_32x32x2 (lane = (position | row half, row),
// register = channel): [resample conv] -> ResnetBlock -> ResnetBlock in ONE launch like k_level_fwd and, at n == 1,
// Residual(PreNorm(LinearAttention)) (linear over one position) and the last down level's k3 conv written in the bottleneck's
// (B, C, RT) layout.  `lv` carries the stage and the blocks exactly as for launch_level_fwd (its img field is unused).
struct TinyFwd {
  LevelFwd lv;
  int la = 0;  // n == 1: the level's LinearAttention rides along
  const float* w_qkv = nullptr; const float* w_out = nullptr; const float* b_out = nullptr; const float* g_pre = nullptr; const float* g_out = nullptr;
  float* la_y = nullptr; float* la_ypre = nullptr;  // (rows, C, 1); la_ypre nullable (training: the pre-norm output the backward reads)
  const float* post_w = nullptr; const float* post_b = nullptr; float* post_out = nullptr;  // n == 1, behind the LinearAttention: k3 conv (C, C, 3) -> (B, C, RT)
  int in_folded = 0;          // LEVEL_PRE_NONE, n == 1: lv.in is (B, C, RT)
  float* in_copy = nullptr;   // nullable: the input again as (rows, C, 1) (training: the backward reads it in that layout)
  const float* img = nullptr; // operand image (tiny_img_floats floats, 16-byte aligned) from launch_tiny_images on the SAME parameter values
};
constexpr int TINY_IMG_MAX = 6;  // four forward launches + two backward ones
constexpr int TINY_IMG_FLOATS = 16384;  // (whole rounds of 256 x 16 bytes: the kernels copy a slot without guards)
bool tiny_fwd_usable(const TinyFwd& t);
int64_t tiny_img_floats(const TinyFwd& t);
int launch_tiny_images(const TinyFwd* calls, int count, hipStream_t s);
int launch_tiny_fwd(const TinyFwd& t, hipStream_t s);

// k_tiny.hip, backward: the DATA path of a level with rows of one position in one launch ([last down level's k3 conv] -> LinearAttention ->
// ResnetBlock 1 -> ResnetBlock 0 -> [Downsample]); weight gradients stay with the side-stream kernels, which read the d u1 / d u2 / d rs /
// d out tensors this launch writes; norm gains and per-sample d(scale, shift) leave as ResBwd::gpart partials (*gblocks = blocks per
// sample), the LinearAttention's gradients as one k_linattn_bwd1-format slot per workgroup (tiny_bwd_slots of them).
struct TinyBwd {
  const float* params = nullptr; const float* img = nullptr;
  int C = 0, rows = 0, rows_per_sample = 1, pre = LEVEL_PRE_NONE, cp = 0, cs = 0;
  const float* x = nullptr; const float* ypre = nullptr; const float* dy = nullptr;      // block-1 output, saved pre-norm LinearAttention output, d la
  const float* w_qkv = nullptr; const float* w_out = nullptr; const float* g_pre = nullptr; const float* g_out = nullptr;
  float* la_part = nullptr; int64_t la_part_floats = 0;
  const float* post_w = nullptr; const float* dmid = nullptr; float* drs_out = nullptr;   // LEVEL_PRE_DOWN (last down level): its k3 conv, d mid_in (B, C, RT), d rs (rows, C) out
  float* dfold = nullptr;                                                                 // LEVEL_PRE_NONE: input gradient as (B, C, RT)
  float* din_rows = nullptr; float* dprev = nullptr; const float* stage_w = nullptr;      // LEVEL_PRE_DOWN: input gradient (rows, C); previous level's d la (rows, cp, 2) +=; Downsample weight
  const float* r0out_g = nullptr;                                                         // nullable: skip gradient already in d r0.out
  // LEVEL_PRE_NONE, nullable pair: the Upsample conv BEHIND this level (nearest x2 + k3, 16 -> 16, rows of 2 positions) -- its backward data path runs
  // in front of the LinearAttention part: d la = Upsample^T d rs with dup = d rs (rows, C, 2), and dy is not read
  const float* up_w = nullptr; const float* dup = nullptr;
  struct Blk {
    const float* w1 = nullptr; const float* w2 = nullptr; const float* wr = nullptr; const float* g1 = nullptr; const float* g2 = nullptr;
    const float* ss = nullptr; int ss_stride = 0;
    const float* u1 = nullptr; const float* u2 = nullptr; float* du1 = nullptr; float* du2 = nullptr;
    float* dB = nullptr; int dB_acc = 0; float* dout_st = nullptr; float* gpart = nullptr; int64_t gpart_floats = 0;
  } blk[2];
  int* gblocks = nullptr;
};
bool tiny_bwd_usable(const TinyBwd& t);
int tiny_bwd_slots(const TinyBwd& t);
int64_t tiny_bwd_img_floats(const TinyBwd& t);
int launch_tiny_bwd_images(const TinyBwd* calls, int count, hipStream_t s);
int launch_tiny_bwd(const TinyBwd& t, hipStream_t s);
// k_res_wg.hip: ResnetBlock backward of the wide levels (C = 4 / 8, rows of 8..256 positions) with the block's weight gradients formed
// in the same launch on the 4x4x1 matrix pipe.  Every workgroup leaves [c1.w | c1.b | g1 | c2.w | c2.b | g2 | res.w | res.b | dscale |
// dshift] in its own slot of `part`; launch_res_wg_reduce adds the slots up in block order into the flat gradient buffer (the
// block's parameters are contiguous there, in this order) and into the per-sample d(scale, shift).
struct ResBwdWg {
  const float* dout = nullptr; const float* u1 = nullptr; const float* u2 = nullptr;
  const float* inA = nullptr; const float* inB = nullptr; int cinA = 0, cinB = 0;   // the block's input (forward activations)
  const float* w1 = nullptr; const float* w2 = nullptr; const float* wr = nullptr;
  const float* g1 = nullptr; const float* g2 = nullptr; const float* ss = nullptr; int ss_stride = 0;
  float* dA = nullptr; float* dB = nullptr; int dA_store = 0, dB_store = 0;         // as ResBwd
  float* part = nullptr; int64_t part_floats = 0;                                   // >= res_wg_part_floats(...)
  float* dparams = nullptr;   // gradient of block1.proj.weight in the flat gradient buffer (the slot's first tensor)
  float* dss = nullptr;       // this block's [dscale | dshift] of sample 0 (stride ss_stride)
  int C = 0, rows = 0, n = 0, rows_per_sample = 1;
  int tiles_ps = 0, tpb = 0, nv = 0;  // (filled by the launcher)
};
struct ResWgReduce { const float* part; int B, gx, nv, nglob, C; float* dst; float* dss; int ss_stride; };
constexpr int RES_WG_REDUCE_MAX = 32;
struct ResWgReduceMulti { ResWgReduce it[RES_WG_REDUCE_MAX]; };
bool res_wg_usable(int n, int C, int cinA, int cinB, int rows_per_sample);
int64_t res_wg_part_floats(int C, int cin, bool wr, int B, int rows_per_sample, int n);
// launches the backward; *red_out receives the descriptor of the slot reduction (launch_res_wg_reduce: right away or collected)
int launch_res_bwd_wg(const ResBwdWg& a, hipStream_t s, ResWgReduce* red_out);
int launch_res_wg_reduce(const ResWgReduce* items, int count, hipStream_t s);
// k_conv_wg.hip: backward of a level's resample conv (data gradient + weight / bias gradient in one launch, matrix pipe).  `n` is the
// conv's OUTPUT row length; pre = LEVEL_PRE_DOWN (k4 s2) / _UP (nearest x2 + k3) / _S1 (k3).  Slots [dW | dbias] -> launch_res_wg_reduce.
struct ConvBwdWg {
  const float* dy = nullptr;   // (rows, C, n) gradient of the conv output
  const float* in = nullptr;   // (rows, cp, n_in) forward input
  const float* w = nullptr;    // (C, cp, K)
  float* din = nullptr; int accumulate = 0;   // (rows, cp, n_in): = or += ; null: not needed
  float* part = nullptr; int64_t part_floats = 0;
  float* dparams = nullptr;    // gradient of the conv weight in the flat gradient buffer (the bias gradient follows it)
  int C = 0, pre = 0, cp = 0, rows = 0, n = 0, rows_per_sample = 1;
  int tiles_ps = 0, tpb = 0, nv = 0;  // (filled by the launcher)
};
bool conv_wg_usable(int C, int pre, int cp, int n, int rows_per_sample);
int64_t conv_wg_part_floats(int C, int pre, int cp, int B, int rows_per_sample, int n);
int launch_conv_bwd_wg(const ConvBwdWg& a, hipStream_t s, ResWgReduce* red_out);
// the descriptor of a fused ResnetBlock backward's partial sums for launch_part_reduce
inline PartReduce res_part_reduce(const float* gpart, int gx, int B, int C, float* dg2, float* dg1, float* dss, int ss_stride) {
  PartReduce r;
  r.part = gpart; r.B = B; r.gx = gx; r.nv = 4 * C; r.nseg = 2;
  r.seg_start[0] = 0; r.seg_len[0] = C; r.seg_dst[0] = dg2;
  r.seg_start[1] = C; r.seg_len[1] = C; r.seg_dst[1] = dg1;
  r.s0 = 2 * C; r.sn = 2 * C; r.sdst = dss; r.sstride = ss_stride;
  return r;
}

// standalone RMSNorm forward (PreNorm of the bottleneck attention)
int launch_rmsnorm_fwd(const float* x, const float* g, float* y, int C, int rows, int n, hipStream_t s);

// (rows=B*RT, C, n) <-> (B, C*n, RT) fold of the bottleneck (unet1d.py:1144-1148); add=1 accumulates; pitch: row pitch of the
// (B, C*n, .) side (0 = RT; the wide bottleneck pads it, and the pad columns of a to_mid output are zeroed)
int launch_fold(const float* in, float* out, int B, int RT, int cn, int to_mid, int add, hipStream_t s, int pitch = 0);

// first layer inputs: cat0 = [cond_n*(scale+1)+shift, x] as (rows, 2, MZ); ms1n = ms1*cm+ca
int launch_prep_inputs(const float* x, const float* cond, const float* ms1, const float* ss, int ss_stride, int ss_off, float cm,
                       float ca, float* cat0, float* ms1n, int B, int RT, int MZ, hipStream_t s);
int launch_ms1_norm(const float* ms1, float cm, float ca, float* ms1n, int64_t n, hipStream_t s);  // ms1n = ms1 * cm + ca
// d(scale), d(shift) of init_cond_proj from dcat0 channel 0 (+= into dss; part: >= 64 * B floats of per-block partials)
int launch_prep_inputs_bwd(const float* dcat0, const float* cond, float cm, float ca, float* dss, int ss_stride, int ss_off, int B,
                           int RT, int MZ, float* part, int64_t part_floats, hipStream_t s);

// ---- k_wide.hip : the bottleneck when it is wide (mid_dim * downsampled_n channels not in {16, 32, 64}); tensors (B, C, P), P = RT
// padded to a multiple of 4, pad columns written as zeros
int launch_im2col3(const float* x, float* xcol, int B, int C, int RT, int P, hipStream_t s);                  // (B,C,P) -> (B,3C,P)
int launch_col2im3(const float* dxcol, float* dx, int B, int C, int RT, int P, int accumulate, hipStream_t s);  // transpose of it
int launch_repitch(float* dst, int dpitch, const float* src, int spitch, int64_t rows, int n, hipStream_t s);
int launch_wnorm_fwd(const float* u, const float* g, const float* ss, int ss_stride, int act, const float* res, float* y, int B, int C,
                     int RT, int P, hipStream_t s);
// scratch: B * (2 P + 2 C) floats; dg / dss / dbias +=
int launch_wnorm_bwd(const float* u, const float* dy, const float* g, const float* ss, int ss_stride, int act, float* du, float* dg,
                     float* dss, float* dbias, float* scratch, int B, int C, int RT, int P, hipStream_t s);
int launch_rowsum(const float* x, int64_t rows, int RT, int P, float* out, hipStream_t s);
int launch_sum_b(const float* part, int B, int C, float* dst, hipStream_t s);

// ---- k_time.hip (declared in dq_unet.h: needs the plan)

// ---- k_linattn.hip
struct LinAttn {
  const float* x = nullptr; float* y = nullptr;  // (rows, C, n)
  const float* w_qkv = nullptr; const float* w_out = nullptr; const float* b_out = nullptr;
  const float* g_pre = nullptr; const float* g_out = nullptr;
  int C = 0, rows = 0, n = 0;
  float* ypre = nullptr;  // optional: pre-norm output Wo*out + b (rows, C, n), saved for the backward
  // optional: this layer's LA_PREP_FLOATS prepared weights from launch_linattn_prepare (W2 = Wo Wv per head and the MFMA operand
  // image of Wq | Wk); without it every block of the forward derives them itself (a 64-load-deep prologue per 4 rows)
  const float* prep = nullptr;
};
int launch_linattn_fwd(const LinAttn& a, hipStream_t s);
// k_la_small.hip: rows of 2 / 4 (/ 8) positions at 8 / 12 / 16 channels with a prepared image (LinAttn::prep): every product on
// v_mfma_f32_32x32x2 with lane = (channel half, row) and one group of registers per position (launch_linattn_fwd dispatches to it)
bool la_small_usable(int C, int n);
int la_small_min_rows();  // launch_linattn_fwd takes this path from that many rows on
int launch_la_small_fwd(const LinAttn& a, hipStream_t s);
// k_la_rows_bwd.hip: the backward in the same spirit (one m/z row per lane column, every product on v_mfma_f32_16x16x4_f32); one slot per
// wave in the la_slot(C) layout; launch_linattn_bwd dispatches to it when the layer's prepared weights are at hand
struct LinAttnBwd;
bool la_rows_bwd_usable(int C, int n);
int la_rows_bwd_min_rows();
// k_la_rows_fwd.hip: the forward in the same layout (rows of 2 / 4 positions below la_small_min_rows; launch_linattn_fwd dispatches to it)
bool la_rows_fwd_usable(int C, int n);
int launch_la_rows_fwd(const LinAttn& a, hipStream_t s);
int launch_la_rows_bwd(const LinAttnBwd& a, int max_slots, int* slots_out, hipStream_t s);
constexpr int LA_PREP_BOUNDED = 1024 + 4096;          // 1.0f when the layer's softmax logits are bounded by 64 for every input (k_linattn_prepare)
constexpr int LA_PREP_BF16 = 1024 + 4096 + 8;         // split-bf16 operand image of Wq | Wk for 4 / 8 channels: 2048 la_nu(C) <= 6144 dwords (k_linattn.hip)
constexpr int LA_PREP_SMALL = 1024 + 4096 + 8 + 6144;  // operand image of k_la_small (rows of <= 8 positions, C = 8 / 12 / 16): [q | k | W2][head][step < C / 2][64 lanes] <= 6144 floats
// operand image of k_la_rows_bwd (rows of 2 / 4 positions, C = 8 / 12 / 16; v_mfma_f32_16x16x4_f32 A operands): [head][lane][LA_ROWS_LANE_FLOATS] =
// [Wq log2(e): 8 | Wk log2(e): 8 | W2^T: 4 | Wq^T: 8 | Wk^T: 8] (the first three groups hold 2 CPL / 2 CPL / CPL values, CPL = C / 4; k_la_rows_bwd.hip has the index algebra)
constexpr int LA_ROWS_LANE_FLOATS = 36;  // (b128 reads at this lane pitch are bank-conflict-free)
constexpr int LA_PREP_ROWS = 1024 + 4096 + 8 + 6144 + 6144;
constexpr int LA_PREP_FLOATS = LA_PREP_ROWS + 4 * 64 * LA_ROWS_LANE_FLOATS;  // [w2: 4 * 16 * 16][wqk: 2 * 4 * 8 * 2 * 32][bounded, 7 unused][bf16 image][small-row image][rows-backward image]
struct LaPrepItem { const float* w_qkv; const float* w_out; int C; float* prep; const float* g_pre; };
constexpr int LA_PREP_MAX = 16;
struct PrepCopy { const float* src; float* dst; int n; };  // plain copies riding in the same launch (aligned weight slots)
constexpr int PREP_COPY_MAX = 4;
// all LinearAttention layers of a forward (+ up to PREP_COPY_MAX copies): one launch
int launch_linattn_prepare(const LaPrepItem* items, int count, hipStream_t s, const PrepCopy* copies = nullptr, int n_copies = 0);
struct LinAttnBwd {
  LinAttn f;
  const float* dy = nullptr; float* dx = nullptr;  // dx += (dx_store: the only writer of dx in this backward -> plain store)
  int dx_store = 0;
  const float* ypre = nullptr;                     // saved by the forward
  float* dyp = nullptr; float* dxh = nullptr;      // scratch (rows, C, n) each
  float* part = nullptr; int64_t part_floats = 0;  // per-wave dW partial slots: >= LA_MAX_WAVES * 512 * C floats
  // defer_reduce: rows of <= 64 positions leave their slots unreduced and report the slot count in *waves_out (0 when the
  // launch reduced them itself: the long-row path); the caller sums them later with launch_linattn_dw_reduce_multi, and then
  // part_floats only needs la_part_reserve(C)
  int defer_reduce = 0; int* waves_out = nullptr;
  float** w2sum_out = nullptr;  // deferred: receives the address of the 4 C C floats (inside `part`) the reduce leaves the summed dW2 in
  float* dw_qkv = nullptr; float* dw_out = nullptr; float* db_out = nullptr; float* dg_pre = nullptr; float* dg_out = nullptr;
};
// rows of 128 / 256 positions (k_la_long.hip)
int launch_linattn_fwd_long(const LinAttn& a, hipStream_t s);
int launch_linattn_bwd_long(const float* x, const float* dyp, float* dxh, const float* w_qkv, const float* w_out, const float* g_pre,
                            float* part, int C, int rows, int n, int* waves_out, hipStream_t s);
constexpr int LA_MAX_WAVES = 2048;  // the backward grid is one resident round: <= 1024 waves, one partial slot each
int launch_linattn_bwd(const LinAttnBwd& a, hipStream_t s);
// the deferred slot reductions of up to LA_REDUCE_MAX LinearAttention backwards in ONE launch (grad += ordered slot sums)
// w2sum: 4 C C floats of scratch (the summed dW2 of the four heads; the launcher places it right behind the layer's slots);
// w_qkv / w_out: the layer's weights (dWv = Wo^T dW2, dWo = dW2 Wv^T are formed from the sum)
struct LaReduceItem {
  const float* part; int nslots, C; float* dw_qkv; float* dw_out; float* dg_out; float* db_out; float* dg_pre;
  float* w2sum; const float* w_qkv; const float* w_out;
};
constexpr int LA_REDUCE_MAX = 16;
int64_t la_part_reserve(int C);  // floats of slot scratch one deferred launch with C channels can use
int launch_linattn_dw_reduce_multi(const LaReduceItem* items, int count, hipStream_t s);

// ---- k_attn.hip : softmax attention over RT of the bottleneck (q,k,v,o in (B, 128, RT) conv layout)
int launch_rope(float* qk, const float* freqs, int B, int64_t batch_stride, int RT, float sign, hipStream_t s);
int launch_rope2(float* q, int64_t q_bs, float* k, int64_t k_bs, const float* freqs, int B, int RT, float sign, hipStream_t s);
int launch_attn_fwd(const float* q, int64_t q_bs, const float* k, int64_t k_bs, const float* v, int64_t v_bs, float* o, float* lse,
                    int B, int RT, hipStream_t s);
int launch_attn_bwd(const float* q, int64_t q_bs, const float* k, int64_t k_bs, const float* v, int64_t v_bs, const float* o,
                    const float* d_o, const float* lse, float* delta, float* dq, int64_t dq_bs, float* dk, int64_t dk_bs, float* dv,
                    int64_t dv_bs, int B, int RT, hipStream_t s);

}  // namespace dq
