/* libdq_hip.so -- C ABI of the MI355X (gfx950) implementation of dquartic's DDIM hot path.
 *
 * The reference (Roestlab/diffusion-deconvolution-dia-msms-data, "dquartic") is pure PyTorch and has no FFI of
 * its own; its boundary for this path is the Python object protocol of dquartic/model/model.py and
 * dquartic/model/unet1d.py.  Each entry point below names the reference method whose arithmetic it replaces
 * (file:line relative to the reference checkout).  The host-side mirror of that protocol
 * (diffusion-deconvolution-dia-msms-data_amd/dquartic/) binds these symbols with ctypes; INTEGRATION.md shows the stub
 * a reference maintainer would add.
 *
 * Conventions
 *   - plain C types only: device pointers (float* / int64_t*), sizes, a HIP stream passed as void*;
 *   - every call is asynchronous on the given stream and keeps no pointer after it returns, with two documented
 *     exceptions: dq_ddim_sample synchronises the stream twice on entry (its host-side coefficient / timestep tables must
 *     be on the device before the caller's arrays go out of scope, and a hipGraph capture must not see pending copies), and
 *     caches the captured step graph inside the plan;
 *   - the caller supplies every workspace (sizes from dq_unet_workspace_bytes / dq_tfm_workspace_bytes /
 *     dq_resblock_workspace_floats).  The library's only device allocation is made by the first call that uses a plan:
 *     two ~10 KB tables of parameter offsets (hipMalloc, freed by dq_plan_destroy); a side stream and its events are
 *     created by the first backward;
 *   - returns 0 on success; non-zero => dq_last_error() (thread-local text) says why;
 *   - tensors are contiguous fp32; MS2 windows are (B, RT, MZ) with MZ contiguous, MS1 chromatograms (B, RT),
 *     timesteps int64 (B);
 *   - parameters/gradients/AdamW moments are single flat fp32 buffers whose layout is described by
 *     dq_plan_param_info (tensor names == the reference's state_dict keys, reference registration order).  SURVEY 8b
 *     sketched a packed-weights handle (dq_weights_pack / dq_weights_free); the flat buffer replaces it: the caller's
 *     tensor IS the packed form, so there is nothing to pack, free or keep in sync;
 *   - calls that share a dq_plan must not run concurrently; distinct plans are independent.
 */
#ifndef DQ_HIP_H
#define DQ_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dq_plan dq_plan;

/* Text of the last error on this thread ("" if none). */
const char* dq_last_error(void);
/* ABI version of this header (bumped on any signature change).  2: pred_type arguments, dq_ddim_step_x0,
 * dq_mse_loss_weighted_fwd_bwd, dq_pair_batch.  3: dq_tfm_* (CustomTransformer), dq_gemm.
 * 4: dq_tfm_bwd takes an accumulate flag.  5: dq_ddim_sample takes num_timesteps (the plan no longer fixes T); stand-alone
 * building blocks (dq_rmsnorm_fwd, dq_time_mlp_fwd, dq_scale_shift_fwd, dq_prep_inputs_fwd, dq_conv_fwd, dq_resblock_*,
 * dq_rope, dq_attn_*); dq_train_step takes ms1_loss_weight, dq_ms1_loss_fwd_bwd;
 * dq_tfm_set_precision, dq_gemm_bf16x3.  8: dq_tfm_bwd_buckets, dq_tfm_num_buckets, dq_tfm_bucket_info.  9: dq_linattn_prepare,
 * dq_linattn_fwd_prepared.  10: dq_set_option, dq_get_option, dq_debug_side_tail_store. */
int dq_abi_version(void);
#define DQ_ABI_VERSION 10

/* Process-wide tuning options (no reference counterpart: the reference has one code path per op).  The library reads NO environment
 * variable for its dispatch; what can be tuned is set here, takes effect from the next call on, and invalidates cached sampling graphs.
 *   "la_small_min_rows"     rows (B * RT) from which LinearAttention over m/z rows of 2 / 4 / 8 positions runs in the
 *                           one-register-group-per-position form (k_la_small.hip) instead of the register-resident one (k_linattn.hip);
 *                           < 0 (default): the device rule, one 32-row tile per SIMD (32,768 rows on MI355X)
 *   "la_rows_bwd_min_rows"  the same for its backward (k_la_rows_bwd.hip against k_la_bwd.hip); < 0 (default): every row count
 * Both forms compute the same function (parity tests run both at every size).  Unknown key: non-zero / INT64_MIN. */
int dq_set_option(const char* key, int64_t value);
int64_t dq_get_option(const char* key);

/* DDIMDiffusionModel.pred_type (model.py:205-213, 269-280, 354-389); any other value is rejected ("Unknown pred_type"). */
enum { DQ_PRED_EPS = 0, DQ_PRED_X0 = 1 };

/* ---- network description -------------------------------------------------------------------------------------
 * Replaces UNet1d.__init__ (unet1d.py:918-1084) for simple=True, conditional=True, channels=1,
 * init_cond_channels=1, attn_cond_channels=1: builds the layer list and the flat parameter layout.
 * mz == downsample_dim.  num_timesteps is informational (kept for ABI continuity): the schedule tables come with each call.
 * Returns NULL on an unsupported configuration (see dq_last_error). */
dq_plan* dq_plan_create(int dim, int n_mults, const int* dim_mults, int mz, int num_timesteps);
void dq_plan_destroy(dq_plan* plan);
/* Number of trainable tensors / total trainable floats in the flat buffer. */
int dq_plan_num_params(const dq_plan* plan);
int64_t dq_plan_param_floats(const dq_plan* plan);
/* Tensor i: state_dict key (NUL-terminated into name[name_cap]), offset in floats, ndim, shape[4]. */
int dq_plan_param_info(const dq_plan* plan, int i, char* name, int name_cap, int64_t* offset, int* ndim, int64_t* shape);
/* Bytes of workspace dq_unet_fwd / dq_unet_bwd / dq_train_step / dq_ddim_sample need for (B, RT).
 * training != 0 adds the gradient twin of the activation arena. */
int64_t dq_unet_workspace_bytes(dq_plan* plan, int B, int RT, int training);

/* ---- K0: DDIMDiffusionModel.q_sample (model.py:225-242) ---------------------------------------------------------
 * x_t = sqrt(ab[t_b]) * x0' + sqrt(1 - ab[t_b]) * noise, x0' = 2*x0-1 if normalize_x0 (model.py:349) else x0. */
int dq_q_sample(const float* alpha_bars_dev, const float* x0, const int64_t* t, const float* noise, float* x_t, int B,
                int64_t per_sample, int normalize_x0, void* stream);

/* ---- K9: the update of DDIMDiffusionModel.p_sample (model.py:265-289, pred_type "eps") ---------------------------
 * coef_dev: 4 device floats [sqrt(ab_t), sqrt(1-ab_t), sqrt(ab_{t-1}), sqrt(1-ab_{t-1})]; coef_dev[2] < 0 means t == 0
 * (x_prev = x0_pred). */
int dq_ddim_step(const float* x_t, const float* eps, float* x_prev, const float* coef_dev, int64_t n, void* stream);
/* pred_type "x0" (model.py:274-278): the network output is x0_pred; eps = (x_t - sqrt(ab_t)*x0_pred)/sqrt(1-ab_t) is derived
 * and written to eps_out (nullable); x_prev as above. */
int dq_ddim_step_x0(const float* x_t, const float* x0_pred, float* x_prev, float* eps_out, const float* coef_dev, int64_t n,
                    void* stream);

/* ---- K1-K8: UNet1d.forward (unet1d.py:1086-1166) ---------------------------------------------------------------
 * params: flat parameter buffer; rope_freqs: the 8 non-trainable RoPE frequencies (device).
 * x, init_cond (B,RT,MZ); attn_cond (B,RT); t (B) int64 or NULL => every sample uses t_scalar.
 * init_cond/attn_cond are mapped v*cond_mul+cond_add on the fly (2,-1 reproduces model.py:310-311/350-351; 1,0 = raw).
 * out (B,RT,MZ) receives the prediction.  save_for_bwd != 0 also keeps the pre-norm tensors dq_unet_bwd reads. */
int dq_unet_fwd(dq_plan* plan, const float* params, const float* rope_freqs, const float* x, const int64_t* t, int t_scalar,
                const float* init_cond, const float* attn_cond, float cond_mul, float cond_add, float* out, int save_for_bwd,
                void* workspace, int64_t workspace_bytes, int B, int RT, void* stream);
/* Backward of the call above (same plan/workspace/arguments, workspace sized with training=1): accumulates
 * d loss / d params into grads (+=; zero it first) given grad_out = d loss / d out.  grad_x (optional, may be NULL)
 * receives d loss / d x.  Replaces loss.backward() through the network (model_interface.py:1120). */
int dq_unet_bwd(dq_plan* plan, const float* params, const float* rope_freqs, const float* init_cond, float cond_mul,
                float cond_add, const float* grad_out, float* grads, float* grad_x, void* workspace, int64_t workspace_bytes,
                int B, int RT, void* stream);

/* ---- K10: F.mse_loss(eps_pred, noise) and its gradient (model.py:361) ------------------------------------------
 * loss_out: 1 device float (mean over all n elements); grad_out (nullable): 2*(eps-noise)/n.
 * scratch: >= 1024 device floats. */
int dq_mse_loss_fwd_bwd(const float* eps, const float* noise, float* loss_out, float* grad_out, float* scratch, int64_t n,
                        void* stream);
/* pred_type "x0" (model.py:372-376, 404): loss = mean over samples b of loss_weight_dev[t_b] * MSE_b(pred, target*target_mul +
 * target_add); loss_weight_dev: the T-entry table DDIMDiffusionModel.loss_weight (SNR, model.py:205-210); grad_out nullable. */
int dq_mse_loss_weighted_fwd_bwd(const float* pred, const float* target, float target_mul, float target_add,
                                 const float* loss_weight_dev, const int64_t* t, float* loss_out, float* grad_out, float* scratch,
                                 int B, int64_t per_sample, void* stream);

/* ---- K11: clip_grad_norm_(max_norm) + AdamW step (model_interface.py:1121-1122, torch defaults) ----------------
 * grads are first multiplied by grad_scale (1/world_size after a summing all-reduce), the global L2 norm of the
 * scaled grads goes to gnorm_out (nullable, 1 device float), then coef = min(1, max_norm/(norm+1e-6)) (max_norm <= 0
 * disables clipping) and the decoupled-decay AdamW update with bias correction for `step` (1-based).
 * scratch: >= 1024 device floats. */
int dq_adamw_clip_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float* scratch,
                       float grad_scale, float max_norm, double lr, double beta1, double beta2, double eps, double weight_decay,
                       int step, float* gnorm_out, void* stream);
/* The same step with the learning rate (lr_dev[0] + lr_dev[1]: the host's double lr as a (hi, lo) fp32 pair, so the scalar factors formed
 * from it in double match dq_adamw_clip_step's bit for bit) and the step count (*step_dev, int32: incremented by the call, 0 before the first
 * step) in device memory: the call's arguments do not change from step to step, so a captured graph of it can be replayed.  scratch as above. */
int dq_adamw_clip_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float* scratch, float grad_scale,
                           float max_norm, const float* lr_dev, double beta1, double beta2, double eps, double weight_decay, int* step_dev,
                           float* gnorm_out, void* stream);
/* on = 0: the backward's remaining weight-gradient launches run on the caller's stream instead of the plan's side stream (a captured train
 * step is then one chain; the fork / join inside a graph was measured slower than the chain).  Default 1. */
int dq_plan_set_side_stream(dq_plan* plan, int on);

/* ---- the MS1 term of train_step with ms1_loss_weight = w in (0, 1] (model.py:364-371, 379-386, 398-402) -----------------
 * The reference's branch raises (torch.max(x, dim=-1) returns a tuple that is then divided), so the semantics are chosen
 * (DESIGN.md section 12): per sample b, D = x_t - pred ('eps': pass x_t) or pred ('x0': x_t = NULL), s_f[rt] = f over m/z of
 * D[rt][:] for f in {sum, mean, max (values)}, ms1n = ms1_cond*cond_mul+cond_add (B, RT),
 *   additional_b = sum_f mean_rt (s_f[rt] / max_rt s_f - ms1n[rt] / max_rt ms1n)^2 .
 * On entry loss_inout (1 device float) / grad_inout (B,RT,MZ; nullable) hold the MSE part (dq_mse_loss[_weighted]_fwd_bwd);
 * on exit loss = (1-w) * MSE + w * mean_b lw_b additional_b and its gradient w.r.t. pred (lw = loss_weight_dev[t_b], NULL = 1).
 * scratch: 5*B*RT + B floats. */
int dq_ms1_loss_fwd_bwd(const float* pred, const float* x_t, const float* ms1_cond, float cond_mul, float cond_add,
                        const float* loss_weight_dev, const int64_t* t, float ms1_loss_weight, float* loss_inout, float* grad_inout,
                        float* scratch, int B, int RT, int MZ, void* stream);

/* ---- DDIMDiffusionModel.train_step (model.py:326-406) fused with its backward ----------------------------------------
 * normalise x0/conds, q_sample, network forward, MSE loss, backward into grads (+=).  t (B) int64 and noise (B,RT,MZ) are
 * drawn by the caller (the reference draws randint then randn_like, model.py:344-346).  pred_type DQ_PRED_EPS: target =
 * noise, loss_weight_dev ignored (may be NULL); DQ_PRED_X0: target = normalised x0, every sample weighted by
 * loss_weight_dev[t_b] (model.py:209-210, 404).  ms1_loss_weight in [0, 1]: 0 = the MSE alone; > 0 adds the MS1 term
 * (dq_ms1_loss_fwd_bwd above).  loss_out: 1 device float = mean over the batch of the per-sample loss. */
int dq_train_step(dq_plan* plan, const float* params, const float* rope_freqs, const float* alpha_bars_dev, const float* x0,
                  const float* ms2_cond, const float* ms1_cond, const int64_t* t, const float* noise, int auto_normalize,
                  int pred_type, const float* loss_weight_dev, float ms1_loss_weight, float* grads, float* loss_out, void* workspace,
                  int64_t workspace_bytes, int B, int RT, void* stream);

/* ---- DDIMDiffusionModel.sample (model.py:293-324) --------------------------------------------------------------
 * Runs the whole strided loop natively over timesteps_host[num_steps] (host ints; the caller forms them as
 * trunc(linspace(T-1, 0, num_steps)), model.py:313): each step = network forward + K9 (landing on alpha_bars[t-1],
 * model.py:284), then the epilogue (model.py:319-322).  alpha_bars_host: num_timesteps host floats (DDIMDiffusionModel.alpha_bars;
 * every timestep must lie in [0, num_timesteps)); num_steps <= 1024.  x_T (B,RT,MZ) is not modified.  out_x = denoised in [0,1]; out_noise = mixture -
 * denoised.  traj_x / traj_eps (nullable): (num_steps,B,RT,MZ) per-step x_{t-1} and eps.  use_graph != 0 (and no trajectory
 * requested; traj_eps always holds eps_pred, derived from the x0 prediction under DQ_PRED_X0): one step is captured
 * into a hipGraph (cached in the plan while params/workspace/B/RT stay the same) and
 * replayed num_steps times; the conditions are staged inside the workspace, the step index lives on the device. */
int dq_ddim_sample(dq_plan* plan, const float* params, const float* rope_freqs, const float* alpha_bars_host, int num_timesteps,
                   const float* x_T, const float* ms2_cond, const float* ms1_cond, int auto_normalize, int pred_type,
                   const int32_t* timesteps_host, int num_steps, float* out_x, float* out_noise, float* traj_x, float* traj_eps,
                   int use_graph, void* workspace, int64_t workspace_bytes, int B, int RT, void* stream);

/* ---- batch formation from an HBM-resident dataset (the step right before the hot path) ------------
 * Replaces, for B (window 1, window 2) pairs, DIAMSDataset.__getitem__'s min-max normalisation (utils/data_loader.py:70-79:
 * MS2 min/max over both windows, MS1 min/max over window 1 only, (x - min) / (max - min), no epsilon: a constant pair gives
 * NaN like the reference) and the mixture ms2_cond = w1*ms2_1 + w2*ms2_2 of _train_one_epoch (model_interface.py:1073-1075).
 * Bit-identical to the reference's numpy/torch arithmetic on float32 data without NaNs.
 * ms2_data (n_windows, RT, MZ), ms1_data (n_windows, ms1_per_window) fp32 on the device; idx_dev: 2*B device int64
 * [idx1 (B) | idx2 (B)] -- a pair with an index outside [0, n_windows) is never dereferenced, its outputs are NaN.
 * Outputs (device): ms2_1, ms2_2, ms2_cond (B, RT, MZ); ms1_1, ms1_2 (B, ms1_per_window); ms2_2 / ms1_2 / ms2_cond nullable.
 * scratch: dq_pair_batch_scratch_bytes(B) bytes.  Asynchronous on stream, no host synchronisation. */
int64_t dq_pair_batch_scratch_bytes(int B);
int dq_pair_batch(const float* ms2_data, const float* ms1_data, int64_t n_windows, const int64_t* idx_dev, int B, int RT, int MZ,
                  int64_t ms1_per_window, float w1, float w2, float* ms2_1, float* ms1_1, float* ms2_2, float* ms1_2,
                  float* ms2_cond, void* scratch, int64_t scratch_bytes, void* stream);

/* ---- CustomTransformer (dquartic/model/building_blocks.py:179-260) ----------------------------------
 * The reference's alternative noise predictor: forward(x_t (B,S1,input_dim), t (B) int64, x_cond (B,S2)) -> (B,S1,input_dim).
 * The handle fixes the parameter layout: one flat fp32 buffer, tensors under the reference's state_dict keys in its
 * registration order (dq_tfm_param_info).  Needs input_dim % 4 == 0, hidden_dim % 8 == 0, (hidden_dim / num_heads) % 4 == 0;
 * NULL otherwise (dq_last_error).  Calls that share a handle must not run concurrently. */
typedef struct dq_tfm dq_tfm;
dq_tfm* dq_tfm_create(int input_dim, int hidden_dim, int num_heads, int num_layers);
void dq_tfm_destroy(dq_tfm* tfm);
int dq_tfm_num_params(const dq_tfm* tfm);
int64_t dq_tfm_param_floats(const dq_tfm* tfm);
/* name_cap bytes of name; shape: 2 entries (a vector has shape[1] = 1). */
int dq_tfm_param_info(const dq_tfm* tfm, int i, char* name, int name_cap, int64_t* offset, int* ndim, int64_t* shape);
int64_t dq_tfm_workspace_bytes(const dq_tfm* tfm, int B, int S1, int S2, int training);
/* Forward (building_blocks.py:224-260).  rope_sin / rope_cos: (max(S1,S2), hidden_dim/2) device tables of apply_rope's angles
 * (:31-49) and time_freqs: (hidden_dim/2) of TimeEmbedding (:104-106) -- formed by the caller with the reference's own torch
 * expressions so that they are bit-identical.  save_for_bwd != 0 keeps every layer's activations in the workspace. */
int dq_tfm_fwd(dq_tfm* tfm, const float* params, const float* rope_sin, const float* rope_cos, const float* time_freqs,
               const float* x_t, const int64_t* t, const float* x_cond, float* out, int save_for_bwd, void* workspace,
               int64_t workspace_bytes, int B, int S1, int S2, void* stream);
/* Backward of the last dq_tfm_fwd(save_for_bwd = 1) on the same workspace.  grads: flat, same layout as params; accumulate != 0:
 * += (autograd's convention); accumulate == 0: plain stores -- every parameter gradient is produced exactly once per backward,
 * so a training step needs no zeroing pass over the 764 MB buffer of the reference configuration.
 * dx_t (B,S1,input_dim) and dx_cond (B,S2) are plain stores and may be NULL. */
int dq_tfm_bwd(dq_tfm* tfm, const float* params, const float* rope_sin, const float* rope_cos, const float* x_t,
               const float* x_cond, const float* dout, float* grads, int accumulate, float* dx_t, float* dx_cond,
               void* workspace, int64_t workspace_bytes, int B, int S1, int S2, void* stream);
/* The same backward for data-parallel training (the reference wraps the model in DistributedDataParallel, model_interface.py
 * :953-957, whose reducer all-reduces gradient buckets while the backward is still running): the flat gradient buffer is cut into
 * dq_tfm_num_buckets() = num_layers + 1 contiguous slices -- bucket i < num_layers is layer num_layers-1-i, the last bucket is
 * everything registered before the layers -- and on_bucket(user, i, offset, count) is called ON THE CALLING THREAD as soon as every
 * kernel writing grads[offset .. offset+count) has been enqueued on `stream`, in bucket order.  The callback typically records an
 * event on `stream` and starts that slice's all-reduce on a communication stream; it must not touch other slices. */
typedef void (*dq_tfm_bucket_fn)(void* user, int bucket, int64_t offset, int64_t count);
int dq_tfm_bwd_buckets(dq_tfm* tfm, const float* params, const float* rope_sin, const float* rope_cos, const float* x_t,
                       const float* x_cond, const float* dout, float* grads, int accumulate, float* dx_t, float* dx_cond,
                       void* workspace, int64_t workspace_bytes, int B, int S1, int S2, void* stream,
                       dq_tfm_bucket_fn on_bucket, void* user);
int dq_tfm_num_buckets(const dq_tfm* tfm);
int dq_tfm_bucket_info(const dq_tfm* tfm, int i, int64_t* offset, int64_t* count); /* floats, into the flat buffer */
/* Arithmetic of the transformer's dense products: DQ_PRECISION_FP32 (default) = exact fp32 on v_mfma_f32_32x32x2_f32, the precision
 * every parity statement of this library is made in; DQ_PRECISION_BF16X3 = three bf16 matrix-core passes over operands split into
 * hi + lo bf16 halves with fp32 accumulation (~16 mantissa bits per operand, relative error ~1e-5 per product term): a separate,
 * faster mode with its own stated tolerance, never the default. */
enum { DQ_PRECISION_FP32 = 0, DQ_PRECISION_BF16X3 = 1 };
int dq_tfm_set_precision(dq_tfm* tfm, int precision);
/* The fp32 matrix-core GEMM underneath (exported for the parity tests and the roofline measurement):
 * C (M,N; ldc) = A B (+ bias[n]) with A(m,k) = a_kmajor ? A[m*lda+k] : A[k*lda+m] and B(k,n) = b_kmajor ? B[n*ldb+k] :
 * B[k*ldb+n]; splits = 0 lets the library choose a split-K factor; scratch: dq_gemm_scratch_floats(M,N,K) floats. */
int64_t dq_gemm_scratch_floats(int M, int N, int K);
int dq_gemm(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int64_t lda, int64_t ldb,
            int64_t ldc, int a_kmajor, int b_kmajor, int accumulate, int splits, float* scratch, int64_t scratch_floats,
            void* stream);
/* The same product in the DQ_PRECISION_BF16X3 arithmetic. */
int dq_gemm_bf16x3(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int64_t lda, int64_t ldb,
                   int64_t ldc, int a_kmajor, int b_kmajor, int accumulate, int splits, float* scratch, int64_t scratch_floats,
                   void* stream);

/* ---- building blocks exported for the per-block parity tests (tests/test_blocks_gpu.py) ------------------------
 * Each runs the SAME kernels / dispatch the network uses, on caller-supplied tensors, so that the reference's per-block
 * fixtures (tests/golden/blocks.npz) reach the HIP code directly.
 * Residual(PreNorm(LinearAttention)) (unet1d.py:446-496, 1017) on (rows, C, n). */
int dq_linattn_fwd(const float* x, float* y, float* ypre /* nullable: pre-norm output saved for the backward */,
                   const float* w_qkv, const float* w_out, const float* b_out, const float* g_pre, const float* g_out, int C,
                   int rows, int n, void* stream);
/* The same block the way the network runs it: the layer's derived weights (W2 = Wo Wv per head, the MFMA operand images of Wq | Wk --
 * fp32 and, for 4 / 8 channels, split-bf16 -- and the bounded-logit flag) are formed ONCE per parameter state by dq_linattn_prepare into
 * `prep` (dq_linattn_prep_floats() floats, 16-byte aligned) and every forward launch copies them instead of deriving them per workgroup.
 * n: a power of two <= 64. */
int64_t dq_linattn_prep_floats(void);
int dq_linattn_prepare(const float* w_qkv, const float* w_out, const float* g_pre, int C, float* prep, void* stream);
int dq_linattn_fwd_prepared(const float* x, float* y, float* ypre, const float* w_qkv, const float* w_out, const float* b_out,
                            const float* g_pre, const float* g_out, const float* prep, int C, int rows, int n, void* stream);
/* Backward: dx += d/dx, parameter gradients +=.  ypre from the forward; scratch: 2*rows*C*n + 2048*512*C floats. */
int dq_linattn_bwd(const float* x, const float* ypre, const float* dy, float* dx, const float* w_qkv, const float* w_out,
                   const float* b_out, const float* g_pre, const float* g_out, float* dw_qkv, float* dw_out, float* db_out,
                   float* dg_pre, float* dg_out, float* scratch, int C, int rows, int n, void* stream);
/* RMSNorm (unet1d.py:113-140): y = x / max(||x||_2 over C, 1e-12) * g * sqrt(C) on (rows, C, n); C in {4, 8, 12, 16, 32}. */
int dq_rmsnorm_fwd(const float* x, const float* g, float* y, int C, int rows, int n, void* stream);
/* SinusoidalPosEmb(4) -> Linear(4,16) -> GELU -> Linear(16,16) (unet1d.py:196-218, 956-960): t (B) int64 -> sinu_out (B,4),
 * temb_out (B,16) (either nullable).  scratch: 100 * B floats. */
int dq_time_mlp_fwd(const float* w1, const float* b1, const float* w2, const float* b2, const int64_t* t, float* sinu_out,
                    float* temb_out, float* scratch, int B, void* stream);
/* SiLU -> Linear(16, m) head hanging off the time embedding (ResnetBlock.mlp unet1d.py:292-296; ConditionalScaleShift
 * :662-678): temb (B,16), w (m,16), b (m) -> ss (B,m). */
int dq_scale_shift_fwd(const float* temb, const float* w, const float* b, float* ss, int B, int m, void* stream);
/* The first layer's inputs (unet1d.py:1107-1115, 1122-1124): cat0 (B*RT, 2, MZ) = [ (cond*cond_mul+cond_add) * (scale_b + 1) +
 * shift_b , x ], ms1n (B, RT) = ms1*cond_mul+cond_add; ss (B, 2) = [scale, shift] per sample. */
int dq_prep_inputs_fwd(const float* x, const float* cond, const float* ms1, const float* ss, float cond_mul, float cond_add, float* cat0,
                       float* ms1n, int B, int RT, int MZ, void* stream);
/* Conv1d (+ optional fused RMSNorm with gain norm_g, + optional activation) on (rows, cin, n_in) -> (rows, cout, n_out):
 * mode 0 = stride 1 'same' (K in {1,3,7}), 1 = Downsample k4 s2 p1 (unet1d.py:99-110), 2 = Upsample nearest x2 then k3 p1
 * (:82-96).  act: 0 none, 1 SiLU, 2 GELU.  bias / norm_g nullable. */
int dq_conv_fwd(const float* x, const float* w, const float* bias, const float* norm_g, int act, float* y, int cout, int cin, int K, int mode,
                int rows, int n_in, int n_out, void* stream);
/* ResnetBlock (unet1d.py:271-323) on input cat(xA (rows,cinA,n), xB (rows,cinB,n)) (xB nullable with cinB = 0) with the time
 * embedding temb (rows / rows_per_sample, 16).  params: the block's tensors in state_dict order as ONE flat buffer
 * [mlp.1.weight, mlp.1.bias, block1.proj.weight, block1.proj.bias, block1.norm.g, block2.proj.weight, block2.proj.bias,
 * block2.norm.g (, res_conv.weight, res_conv.bias iff cinA+cinB != cout)].  rows_per_sample: RT at the m/z levels, 1 at the
 * bottleneck.  workspace: dq_resblock_workspace_floats floats, shared by the forward (save_for_bwd = 1) and its backward. */
int64_t dq_resblock_workspace_floats(int cin, int cout, int rows, int n, int rows_per_sample);
int dq_resblock_fwd(const float* params, const float* xA, int cinA, const float* xB, int cinB, const float* temb, float* out, int cout,
                    int rows, int n, int rows_per_sample, int save_for_bwd, float* workspace, int64_t workspace_floats, void* stream);
/* Backward of the call above: dxA / dxB += (zero them first; nullable), grads (same layout as params) += for the conv / norm
 * tensors; dss (rows / rows_per_sample, 2*cout) = d loss / d [scale | shift] (the mlp's gradients follow from it:
 * d mlp.1.bias = sum_b dss_b, d mlp.1.weight = sum_b dss_b (x) SiLU(temb_b); the network does that in its time-embedding backward). */
/* dout == NULL (benchmarks): the gradient of the block output is already in the workspace at dq_resblock_dout_offset(...) floats;
 * dxA / dxB are then plain stores, dss is not copied out, and the call consists of the backward launches only. */
int64_t dq_resblock_dout_offset(int cin, int cout, int rows, int n, int rows_per_sample);
int dq_resblock_bwd(const float* params, const float* xA, int cinA, const float* xB, int cinB, const float* dout, float* dxA, float* dxB,
                    float* grads, float* dss, int cout, int rows, int n, int rows_per_sample, float* workspace, int64_t workspace_floats,
                    void* stream);
/* The convolutional part of a U-Net level in ONE launch (unet1d.py:1134-1142, 1150-1158, 1160-1163; k_level.hip):
 *   out_i = ResnetBlock_i(cat(h, skip_i)),  h = stage(x) for i = 0, h = out_0 for i = 1
 * pre: 0 none (x is (rows, C, n)), 1 Downsample k4 s2 (x is (rows, cp, 2n)), 2 Upsample nearest x2 + k3 (x is (rows, cp, n/2)), 3 k3 conv
 * (x is (rows, cp, n)); skip_i (rows, cs, n), nullable with cs = 0 (then the residual is the identity); temb (rows / rows_per_sample, 16).
 * params: [stage weight (C, cp, K) | stage bias (C)] (pre != 0), then nblocks (1 or 2) blocks in the dq_resblock_fwd layout with
 * cin = C + cs (dq_level_param_floats floats in all).  out0 nullable when nblocks = 2 (inference on the way up keeps only out1).
 * n: a power of two <= 64; C in {4, 8, 12, 16}.  workspace: 2 * (rows / rows_per_sample) * 2 * C floats; with 8256 floats more (and 16-byte
 * aligned) the MFMA operand image of the weights is built there by one launch of its own and the workgroups copy it, as in the network
 * path, instead of each gathering it from the parameter tensors. */
int64_t dq_level_param_floats(int pre, int C, int cp, int cs, int nblocks);
int dq_level_fwd(const float* params, int pre, const float* x, int cp, const float* skip0, const float* skip1, int cs, const float* temb,
                 float* out0, float* out1, int C, int nblocks, int rows, int n, int rows_per_sample, float* workspace,
                 int64_t workspace_floats, void* stream);
/* RoPE of the bottleneck attention (rotary_embedding_torch 'lang' mode as restated in DESIGN.md section 5; unet1d.py:529,
 * 560-561), in place on the 4 heads x 32 channels at the start of each sample of qk (B, >=128, RT); batch_stride in floats;
 * sign +1 forward, -1 the transposed rotation (backward). */
int dq_rope(float* qk, const float* freqs, int B, int64_t batch_stride, int RT, float sign, void* stream);
/* softmax(q k^T 32^-0.5) v over RT (unet1d.py:428-443) for q, k, v, o (B, 128, RT) = 4 heads x 32 channels, RT contiguous;
 * lse (B*4*RT) receives the log-sum-exp the backward needs. */
int dq_attn_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int RT, void* stream);
/* Backward: dq, dk, dv (B, 128, RT) are plain stores; delta: B*4*RT floats of scratch. */
int dq_attn_bwd(const float* q, const float* k, const float* v, const float* o, const float* d_o, const float* lse, float* delta,
                float* dq, float* dk, float* dv, int B, int RT, void* stream);

/* Test hook: offset (in floats) of a named activation inside the workspace laid out by the last call on this plan
 * ("h0", "ms1f", "down3", "down3.la", "mid1", "attn_out", "up0", "fin", ...), or -1. */
int64_t dq_debug_tensor_offset(dq_plan* plan, const char* name);

/* Test hook: from now on the backward's side stream ends with a store of `value` to `addr`, delayed by delay_us microseconds, right in
 * front of the join with the caller's stream (addr = NULL switches it off).  A launch the caller makes on its own stream after
 * dq_train_step / dq_unet_bwd returned -- the flat gradient all-reduce of data-parallel training -- must see the value: that is the
 * ordering the all-reduce relies on (tests/test_dp_gloo.py). */
int dq_debug_side_tail_store(dq_plan* plan, float* addr, float value, int delay_us);

#ifdef __cplusplus
}
#endif
#endif /* DQ_HIP_H */
