// CustomTransformer (reference: dquartic/model/building_blocks.py:179-260) forward and backward on the device: the parameter
// layout (flat fp32 buffer, the reference's state_dict keys in registration order), the workspace layout and the launch
// sequence.  Every dense product is one launch_gemm (k_gemm.hip, exact fp32 on the matrix cores); everything between them is
// k_tfm.hip.  C ABI: dq_tfm_* in include/dq_hip.h.
#include "../../include/dq_hip.h"
#include "dq_common.h"
#include "dq_tfm.h"
#include "dq_kernels.h"
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

namespace dq {

struct TfmParam { std::string name; int64_t offset; int ndim; int64_t shape[2]; int64_t numel; };
struct TfmLayer { int64_t in_w, in_b, out_w, out_b, n1_g, n1_b, f0_w, f0_b, f2_w, f2_b, n2_g, n2_b; };

}  // namespace dq

struct dq_tfm {
  int D = 0, H = 0, heads = 0, layers = 0;
  std::vector<dq::TfmParam> params;
  int64_t total = 0;
  int64_t in_w, in_b, out_w, out_b, c_w, c_b, t1_w, t1_b, t2_w, t2_b;
  std::vector<dq::TfmLayer> L;
  // which training workspaces hold a forward's saved activations (one entry per workspace; several forwards may be in flight before
  // their backwards run: micro-batches whose losses are summed).  An entry stays until the same workspace takes another forward.
  struct Saved { const void* ws; int B, S1, S2; };
  std::vector<Saved> saved;
  int precision = dq::GEMM_FP32;  // arithmetic of the dense products (dq_tfm_set_precision)
};

namespace dq {
namespace {

int64_t add(dq_tfm& p, const std::string& name, int64_t a, int64_t b = 0) {
  TfmParam pi;
  pi.name = name; pi.offset = p.total; pi.ndim = b ? 2 : 1; pi.shape[0] = a; pi.shape[1] = b ? b : 1; pi.numel = a * (b ? b : 1);
  p.total += pi.numel;
  p.params.push_back(pi);
  return pi.offset;
}

constexpr int64_t PARTIAL_FLOATS = (int64_t)(768 + 256) * 128 * 128;  // bound of launch_gemm's split-K scratch (k_gemm.hip: choose())

// workspace: carved in a fixed order; `training` keeps one set of layer buffers per layer (read by the backward)
struct Ws {
  float *cp, *tfeat, *th, *tg, *temb, *x0, *tmp, *partial, *colscr, *lnscr;
  struct Layer { float *comb, *q, *kv, *prob, *ao, *y1, *st1, *x1, *hpre, *hact, *y2, *st2, *xo; };
  std::vector<Layer> L;
  // backward only
  float *dxa, *dxb, *dh, *dq, *dkv, *dprob, *dao, *dcomb, *dcp, *dtemb, *dtg;
  int64_t floats = 0;
};
inline int64_t up4(int64_t v) { return (v + 3) & ~(int64_t)3; }

Ws carve(const dq_tfm& p, float* base, int B, int S1, int S2, bool training) {
  Ws w;
  int64_t off = 0;
  auto take = [&](int64_t n) { float* r = base ? base + off : nullptr; off += up4(n); return r; };
  const int64_t H = p.H, R1 = (int64_t)B * S1, R2 = (int64_t)B * S2, Sk = S1 + S2, ldp = up4(Sk);
  w.cp = take(R2 * H); w.tfeat = take(B * H); w.th = take(B * 4 * H); w.tg = take(B * 4 * H); w.temb = take(B * H);
  w.x0 = take(R1 * H); w.tmp = take(R1 * H);
  w.partial = take(PARTIAL_FLOATS);
  w.colscr = take(std::max<int64_t>((int64_t)COLSUM_BLOCKS * std::max<int64_t>(p.D, 4 * H), 2 * H * 64));
  w.lnscr = take((int64_t)2 * H * LN_BWD_BLOCKS);
  const int nl = training ? p.layers : 1;
  for (int l = 0; l < nl; ++l) {
    Ws::Layer a;
    a.comb = take(B * Sk * H); a.q = take(R1 * H); a.kv = take(B * Sk * 2 * H); a.prob = take((int64_t)B * p.heads * S1 * ldp);
    a.ao = take(R1 * H); a.y1 = take(R1 * H); a.st1 = take(R1 * 2); a.x1 = take(R1 * H); a.hpre = take(R1 * 4 * H);
    a.hact = take(R1 * 4 * H); a.y2 = take(R1 * H); a.st2 = take(R1 * 2); a.xo = take(R1 * H);
    w.L.push_back(a);
  }
  if (training) {
    w.dxa = take(R1 * H); w.dxb = take(R1 * H); w.dh = take(R1 * 4 * H); w.dq = take(R1 * H); w.dkv = take(B * Sk * 2 * H);
    w.dprob = take((int64_t)B * p.heads * S1 * ldp); w.dao = take(R1 * H); w.dcomb = take(B * Sk * H); w.dcp = take(R2 * H);
    w.dtemb = take(B * H); w.dtg = take(B * 4 * H);
  }
  w.floats = off;
  return w;
}

// y (M, N) = x (M, K) W^T + b   with W an nn.Linear weight (N, K)
int linear_fwd(const float* x, const float* w, const float* b, float* y, int M, int N, int K, const Ws& ws, hipStream_t s) {
  Gemm g;
  g.A = x; g.B = w; g.C = y; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.bias = b;
  g.partial = ws.partial; g.partial_floats = PARTIAL_FLOATS;
  return launch_gemm(g, s);
}
// dW (N, K) += dy^T x ; db (N) += column sums of dy ; dx (M, K) (+)= dy W  (dx nullable)
int linear_bwd(const float* x, const float* w, const float* dy, float* dw, float* db, float* dx, int dx_accumulate, int M, int N, int K,
               const Ws& ws, hipStream_t s, int acc) {
  Gemm g;
  g.A = dy; g.a_kmajor = 0; g.lda = N; g.B = x; g.b_kmajor = 0; g.ldb = K; g.C = dw; g.ldc = K; g.M = N; g.N = K; g.K = M; g.accumulate = acc;
  g.partial = ws.partial; g.partial_floats = PARTIAL_FLOATS;
  if (int rc = launch_gemm(g, s)) return rc;
  if (db)
    if (int rc = launch_colsum(dy, M, N, N, db, ws.colscr, s, acc)) return rc;
  if (dx) {
    Gemm h;
    h.A = dy; h.lda = N; h.B = w; h.b_kmajor = 0; h.ldb = K; h.C = dx; h.ldc = K; h.M = M; h.N = K; h.K = N; h.accumulate = dx_accumulate;
    h.partial = ws.partial; h.partial_floats = PARTIAL_FLOATS;
    if (int rc = launch_gemm(h, s)) return rc;
  }
  return 0;
}

__global__ void __launch_bounds__(256) k_build_comb(const float* __restrict__ cp, const float* __restrict__ x, float* __restrict__ comb, int B, int S1,
                                                    int S2, int H) {
  const int64_t per = (int64_t)(S1 + S2) * H, total = per * B;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / per, r = e - b * per;  // building_blocks.py:161: cat([x_cond, x_t], dim=1)
    comb[e] = r < (int64_t)S2 * H ? cp[b * S2 * H + r] : x[b * S1 * H + (r - (int64_t)S2 * H)];
  }
}
// dcp += d comb[:, :S2] ; dx += d comb[:, S2:]
__global__ void __launch_bounds__(256) k_split_comb(const float* __restrict__ dcomb, float* __restrict__ dcp, float* __restrict__ dx, int B, int S1,
                                                    int S2, int H) {
  const int64_t per = (int64_t)(S1 + S2) * H, total = per * B;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t b = e / per, r = e - b * per;
    if (r < (int64_t)S2 * H) dcp[b * S2 * H + r] += dcomb[e];
    else dx[b * S1 * H + (r - (int64_t)S2 * H)] += dcomb[e];
  }
}
inline unsigned grid_for(int64_t n) { return (unsigned)std::min<int64_t>(cdiv(n, 256), 8192); }

struct AttnDims { int B, S1, Sk, H, heads, dh; int64_t ldp; };
// batched (sample, head) products of the attention; `which`: 0 scores = Q K^T, 1 O = P V, 2 dP = dO V^T, 3 dV = P^T dO,
// 4 dQ = dS K, 5 dK = dS^T Q
int attn_gemm(int which, const AttnDims& d, const float* q, const float* kv, float* prob, float* o, const Ws& ws, hipStream_t s) {
  Gemm g;
  g.batch = d.B * d.heads; g.inner = d.heads;
  const int64_t qs = (int64_t)d.S1 * d.H, kvs = (int64_t)d.Sk * 2 * d.H, ps = (int64_t)d.S1 * d.ldp;
  g.partial = ws.partial; g.partial_floats = PARTIAL_FLOATS; g.splits = 1;
  switch (which) {
    case 0: case 2:  // (S1 x dh) (Sk x dh)^T -> (S1 x Sk): A = q or dO, B = K (which 0) or V (which 2) rows
      g.A = q; g.lda = d.H; g.sAo = qs; g.sAi = d.dh;
      g.B = kv + (which == 2 ? d.H : 0); g.ldb = 2 * d.H; g.sBo = kvs; g.sBi = d.dh;
      g.C = prob; g.ldc = d.ldp; g.sCo = ps * d.heads; g.sCi = ps; g.M = d.S1; g.N = d.Sk; g.K = d.dh;
      break;
    case 1: case 4:  // (S1 x Sk) (Sk x dh) -> (S1 x dh): A = P or dS, B = V (1) or K (4) as a row-major (Sk, dh) block
      g.A = prob; g.lda = d.ldp; g.sAo = ps * d.heads; g.sAi = ps;
      g.B = kv + (which == 1 ? d.H : 0); g.b_kmajor = 0; g.ldb = 2 * d.H; g.sBo = kvs; g.sBi = d.dh;
      g.C = o; g.ldc = d.H; g.sCo = qs; g.sCi = d.dh; g.M = d.S1; g.N = d.dh; g.K = d.Sk;
      break;
    default:         // 3, 5: (S1 x Sk)^T (S1 x dh) -> (Sk x dh) into the V (3) or K (5) half of d kv: A = P or dS, B = dO or Q
      g.A = prob; g.a_kmajor = 0; g.lda = d.ldp; g.sAo = ps * d.heads; g.sAi = ps;
      g.B = q; g.b_kmajor = 0; g.ldb = d.H; g.sBo = qs; g.sBi = d.dh;
      g.C = o + (which == 3 ? d.H : 0); g.ldc = 2 * d.H; g.sCo = kvs; g.sCi = d.dh; g.M = d.Sk; g.N = d.dh; g.K = d.S1;
      break;
  }
  return launch_gemm(g, s);
}

// the handle's precision as the thread's GEMM default for the duration of one call
struct PrecisionScope {
  int old;
  explicit PrecisionScope(int p) : old(set_gemm_precision(p)) {}
  ~PrecisionScope() { set_gemm_precision(old); }
};

int check_shapes(const dq_tfm* p, int B, int S1, int S2) {
  DQ_REQUIRE(p, "tfm: null handle");
  DQ_REQUIRE(B > 0 && S1 > 0 && S2 > 0, "tfm: batch and both sequence lengths must be positive");
  return 0;
}

}  // namespace
}  // namespace dq

using namespace dq;

extern "C" {

int64_t dq_gemm_scratch_floats(int M, int N, int K) { return std::max<int64_t>(gemm_partial_floats(M, N, K, 1), 4); }
int dq_gemm(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
            int a_kmajor, int b_kmajor, int accumulate, int splits, float* scratch, int64_t scratch_floats, void* stream) {
  Gemm g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.a_kmajor = a_kmajor; g.b_kmajor = b_kmajor; g.accumulate = accumulate; g.splits = splits; g.partial = scratch; g.partial_floats = scratch_floats;
  return launch_gemm(g, (hipStream_t)stream);
}

int dq_gemm_bf16x3(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, int64_t lda, int64_t ldb, int64_t ldc,
                   int a_kmajor, int b_kmajor, int accumulate, int splits, float* scratch, int64_t scratch_floats, void* stream) {
  Gemm g;
  g.A = A; g.B = B; g.C = C; g.bias = bias; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.a_kmajor = a_kmajor; g.b_kmajor = b_kmajor; g.accumulate = accumulate; g.splits = splits; g.partial = scratch; g.partial_floats = scratch_floats;
  g.precision = GEMM_BF16X3;
  return launch_gemm(g, (hipStream_t)stream);
}

int dq_tfm_set_precision(dq_tfm* tfm, int precision) {
  DQ_REQUIRE(tfm, "dq_tfm_set_precision: null handle");
  DQ_REQUIRE(precision == DQ_PRECISION_FP32 || precision == DQ_PRECISION_BF16X3, "dq_tfm_set_precision: precision must be DQ_PRECISION_FP32 or DQ_PRECISION_BF16X3");
  tfm->precision = precision == DQ_PRECISION_BF16X3 ? GEMM_BF16X3 : GEMM_FP32;
  return 0;
}

dq_tfm* dq_tfm_create(int input_dim, int hidden_dim, int num_heads, int num_layers) {
  if (input_dim <= 0 || input_dim % 4 || hidden_dim < 8 || hidden_dim % 8 || num_heads <= 0 || hidden_dim % num_heads ||
      (hidden_dim / num_heads) % 4 || num_layers <= 0) {
    set_error("dq_tfm_create: need input_dim % 4 == 0, hidden_dim % 8 == 0, hidden_dim divisible by num_heads with a head width "
              "that is a multiple of 4, num_layers >= 1");
    return nullptr;
  }
  dq_tfm* p = new dq_tfm;
  p->D = input_dim; p->H = hidden_dim; p->heads = num_heads; p->layers = num_layers;
  const int64_t D = input_dim, H = hidden_dim;
  // registration order of the reference module (building_blocks.py:211-222; TimeEmbedding :86-88; layer :136-145)
  p->in_w = add(*p, "input_projection.weight", H, D); p->in_b = add(*p, "input_projection.bias", H);
  p->out_w = add(*p, "output_projection.weight", D, H); p->out_b = add(*p, "output_projection.bias", D);
  p->c_w = add(*p, "conditional_projection.weight", H, 1); p->c_b = add(*p, "conditional_projection.bias", H);
  p->t1_w = add(*p, "time_embedding.linear1.weight", 4 * H, H); p->t1_b = add(*p, "time_embedding.linear1.bias", 4 * H);
  p->t2_w = add(*p, "time_embedding.linear2.weight", H, 4 * H); p->t2_b = add(*p, "time_embedding.linear2.bias", H);
  for (int l = 0; l < num_layers; ++l) {
    const std::string pre = "layers." + std::to_string(l) + ".";
    TfmLayer a;
    a.in_w = add(*p, pre + "attention.in_proj_weight", 3 * H, H); a.in_b = add(*p, pre + "attention.in_proj_bias", 3 * H);
    a.out_w = add(*p, pre + "attention.out_proj.weight", H, H); a.out_b = add(*p, pre + "attention.out_proj.bias", H);
    a.n1_g = add(*p, pre + "norm1.weight", H); a.n1_b = add(*p, pre + "norm1.bias", H);
    a.f0_w = add(*p, pre + "ff.0.weight", 4 * H, H); a.f0_b = add(*p, pre + "ff.0.bias", 4 * H);
    a.f2_w = add(*p, pre + "ff.2.weight", H, 4 * H); a.f2_b = add(*p, pre + "ff.2.bias", H);
    a.n2_g = add(*p, pre + "norm2.weight", H); a.n2_b = add(*p, pre + "norm2.bias", H);
    p->L.push_back(a);
  }
  return p;
}
void dq_tfm_destroy(dq_tfm* p) { delete p; }
int dq_tfm_num_params(const dq_tfm* p) { return p ? (int)p->params.size() : 0; }
int64_t dq_tfm_param_floats(const dq_tfm* p) { return p ? p->total : 0; }
int dq_tfm_param_info(const dq_tfm* p, int i, char* name, int name_cap, int64_t* offset, int* ndim, int64_t* shape) {
  DQ_REQUIRE(p && i >= 0 && i < (int)p->params.size(), "dq_tfm_param_info: index out of range");
  const TfmParam& pi = p->params[i];
  if (name && name_cap > 0) { std::strncpy(name, pi.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (offset) *offset = pi.offset;
  if (ndim) *ndim = pi.ndim;
  if (shape) { shape[0] = pi.shape[0]; shape[1] = pi.shape[1]; }
  return 0;
}
int64_t dq_tfm_workspace_bytes(const dq_tfm* p, int B, int S1, int S2, int training) {
  if (!p || B <= 0 || S1 <= 0 || S2 <= 0) return 0;
  return carve(*p, nullptr, B, S1, S2, training != 0).floats * (int64_t)sizeof(float);
}

int dq_tfm_fwd(dq_tfm* p, const float* params, const float* rope_sin, const float* rope_cos, const float* time_freqs, const float* x_t,
               const int64_t* t, const float* x_cond, float* out, int save_for_bwd, void* workspace, int64_t workspace_bytes, int B, int S1,
               int S2, void* stream) {
  if (int rc = check_shapes(p, B, S1, S2)) return rc;
  DQ_REQUIRE(params && rope_sin && rope_cos && time_freqs && x_t && t && x_cond && out && workspace, "dq_tfm_fwd: missing operand");
  DQ_REQUIRE(((uintptr_t)params & 15) == 0 && ((uintptr_t)workspace & 15) == 0 && ((uintptr_t)x_t & 15) == 0, "dq_tfm_fwd: params, x_t and workspace must be 16-byte aligned");
  PrecisionScope prec(p->precision);
  const bool training = save_for_bwd != 0;
  Ws w = carve(*p, (float*)workspace, B, S1, S2, training);
  DQ_REQUIRE(workspace_bytes >= w.floats * (int64_t)sizeof(float), "dq_tfm_fwd: workspace too small (dq_tfm_workspace_bytes)");
  hipStream_t s = (hipStream_t)stream;
  const int H = p->H, D = p->D, R1 = B * S1, Sk = S1 + S2;
  const float* P = params;
  // time embedding (building_blocks.py:92-112)
  if (int rc = launch_time_features(t, time_freqs, w.tfeat, B, H, s)) return rc;
  if (int rc = linear_fwd(w.tfeat, P + p->t1_w, P + p->t1_b, w.th, B, 4 * H, H, w, s)) return rc;
  if (int rc = launch_gelu(w.th, w.tg, (int64_t)B * 4 * H, s)) return rc;
  if (int rc = linear_fwd(w.tg, P + p->t2_w, P + p->t2_b, w.temb, B, H, 4 * H, w, s)) return rc;
  // projections + RoPE (+ time embedding on the x_t side) (:238-253)
  if (int rc = linear_fwd(x_t, P + p->in_w, P + p->in_b, w.x0, R1, H, D, w, s)) return rc;
  if (int rc = launch_rope_add(w.x0, rope_sin, rope_cos, w.temb, B, S1, H, 0, s)) return rc;
  if (int rc = launch_cond_embed(x_cond, P + p->c_w, P + p->c_b, rope_sin, rope_cos, w.cp, B, S2, H, s)) return rc;
  const AttnDims ad{B, S1, Sk, H, p->heads, H / p->heads, up4(Sk)};
  const float* x = w.x0;
  for (int l = 0; l < p->layers; ++l) {
    const TfmLayer& a = p->L[l];
    const Ws::Layer& b = w.L[training ? l : 0];
    hipLaunchKernelGGL(k_build_comb, dim3(grid_for((int64_t)B * Sk * H)), dim3(256), 0, s, w.cp, x, b.comb, B, S1, S2, H);
    DQ_LAUNCH_CHECK();
    // nn.MultiheadAttention (:164-166): q from x_t, k | v from [x_cond ; x_t]
    if (int rc = linear_fwd(x, P + a.in_w, P + a.in_b, b.q, R1, H, H, w, s)) return rc;
    if (int rc = linear_fwd(b.comb, P + a.in_w + (int64_t)H * H, P + a.in_b + H, b.kv, B * Sk, 2 * H, H, w, s)) return rc;
    if (int rc = attn_gemm(0, ad, b.q, b.kv, b.prob, nullptr, w, s)) return rc;
    if (int rc = launch_softmax_rows(b.prob, (int64_t)B * p->heads * S1, Sk, (int)ad.ldp, 1.0f / sqrtf((float)ad.dh), s)) return rc;
    if (int rc = attn_gemm(1, ad, nullptr, b.kv, b.prob, b.ao, w, s)) return rc;
    if (int rc = linear_fwd(b.ao, P + a.out_w, P + a.out_b, w.tmp, R1, H, H, w, s)) return rc;
    if (int rc = launch_layernorm_fwd(x, w.tmp, P + a.n1_g, P + a.n1_b, b.y1, b.x1, b.st1, R1, H, s)) return rc;  // :168
    if (int rc = linear_fwd(b.x1, P + a.f0_w, P + a.f0_b, b.hpre, R1, 4 * H, H, w, s)) return rc;                    // :171
    if (int rc = launch_gelu(b.hpre, b.hact, (int64_t)R1 * 4 * H, s)) return rc;
    if (int rc = linear_fwd(b.hact, P + a.f2_w, P + a.f2_b, w.tmp, R1, H, 4 * H, w, s)) return rc;
    if (int rc = launch_layernorm_fwd(b.x1, w.tmp, P + a.n2_g, P + a.n2_b, b.y2, b.xo, b.st2, R1, H, s)) return rc;  // :172
    x = b.xo;
  }
  if (int rc = linear_fwd(x, P + p->out_w, P + p->out_b, out, R1, D, H, w, s)) return rc;  // :258
  for (size_t i = 0; i < p->saved.size(); ++i)
    if (p->saved[i].ws == workspace) { p->saved.erase(p->saved.begin() + i); break; }  // (an inference forward overwrites what was saved there)
  if (training) {
    if (p->saved.size() >= 64) p->saved.erase(p->saved.begin());
    p->saved.push_back({workspace, B, S1, S2});
  }
  return 0;
}

}  // extern "C"

// Gradient buckets, in the order the backward completes them: layers L-1 .. 0 (each layer's parameters are one contiguous slice of
// the flat buffer), then everything registered before the layers (input / output / conditional projections, time MLP), whose last
// gradients are written at the very end.
static void tfm_bucket(const dq_tfm& p, int i, int64_t* off, int64_t* count) {
  if (i < p.layers) {
    const int l = p.layers - 1 - i;
    *off = p.L[l].in_w;
    *count = (l + 1 < p.layers ? p.L[l + 1].in_w : p.total) - *off;
  } else {
    *off = 0;
    *count = p.L[0].in_w;
  }
}

static int tfm_bwd_impl(dq_tfm* p, const float* params, const float* rope_sin, const float* rope_cos, const float* x_t, const float* x_cond,
                        const float* dout, float* grads, int accumulate, float* dx_t, float* dx_cond, void* workspace,
                        int64_t workspace_bytes, int B, int S1, int S2, void* stream, dq_tfm_bucket_fn on_bucket, void* user) {
  if (int rc = check_shapes(p, B, S1, S2)) return rc;
  DQ_REQUIRE(params && rope_sin && rope_cos && x_t && x_cond && dout && grads && workspace, "dq_tfm_bwd: missing operand");
  {
    bool found = false;
    for (const auto& e : p->saved) found = found || (e.ws == workspace && e.B == B && e.S1 == S1 && e.S2 == S2);
    DQ_REQUIRE(found, "dq_tfm_bwd: no matching dq_tfm_fwd(save_for_bwd = 1) on this workspace");
  }
  DQ_REQUIRE(((uintptr_t)grads & 15) == 0 && ((uintptr_t)dout & 15) == 0, "dq_tfm_bwd: grads and dout must be 16-byte aligned");
  PrecisionScope prec(p->precision);
  Ws w = carve(*p, (float*)workspace, B, S1, S2, true);
  DQ_REQUIRE(workspace_bytes >= w.floats * (int64_t)sizeof(float), "dq_tfm_bwd: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int acc = accumulate ? 1 : 0;  // 0: every parameter gradient is written exactly once by this backward -> plain stores
  const int H = p->H, D = p->D, R1 = B * S1, R2 = B * S2, Sk = S1 + S2;
  const float* P = params;
  float* G = grads;
  const AttnDims ad{B, S1, Sk, H, p->heads, H / p->heads, up4(Sk)};
  if (int rc = launch_zero(w.dcp, (int64_t)R2 * H, s)) return rc;
  // output projection
  float* dx = w.dxa;
  float* other = w.dxb;
  if (int rc = linear_bwd(w.L[p->layers - 1].xo, P + p->out_w, dout, G + p->out_w, G + p->out_b, dx, 0, R1, D, H, w, s, acc)) return rc;
  for (int l = p->layers - 1; l >= 0; --l) {
    const TfmLayer& a = p->L[l];
    const Ws::Layer& b = w.L[l];
    const float* xin = l == 0 ? w.x0 : w.L[l - 1].xo;
    // norm2 + feed-forward: d1 = d y2 (= d x1 through the residual) ...
    float* d1 = other;
    if (int rc = launch_layernorm_bwd(b.y2, b.st2, P + a.n2_g, dx, d1, G + a.n2_g, G + a.n2_b, w.lnscr, R1, H, s, acc)) return rc;
    if (int rc = linear_bwd(b.hact, P + a.f2_w, d1, G + a.f2_w, G + a.f2_b, w.dh, 0, R1, H, 4 * H, w, s, acc)) return rc;
    if (int rc = launch_gelu_bwd(b.hpre, w.dh, w.dh, (int64_t)R1 * 4 * H, s)) return rc;
    if (int rc = linear_bwd(b.x1, P + a.f0_w, w.dh, G + a.f0_w, G + a.f0_b, d1, 1, R1, 4 * H, H, w, s, acc)) return rc;  // ... + through ff
    // norm1 + attention: d2 = d y1 (= d x_in through the residual) ...
    float* d2 = dx;
    if (int rc = launch_layernorm_bwd(b.y1, b.st1, P + a.n1_g, d1, d2, G + a.n1_g, G + a.n1_b, w.lnscr, R1, H, s, acc)) return rc;
    if (int rc = linear_bwd(b.ao, P + a.out_w, d2, G + a.out_w, G + a.out_b, w.dao, 0, R1, H, H, w, s, acc)) return rc;
    if (int rc = attn_gemm(2, ad, w.dao, b.kv, w.dprob, nullptr, w, s)) return rc;            // dP = dO V^T
    if (int rc = attn_gemm(3, ad, w.dao, nullptr, b.prob, w.dkv, w, s)) return rc;             // dV = P^T dO
    if (int rc = launch_softmax_rows_bwd(b.prob, w.dprob, (int64_t)B * p->heads * S1, Sk, (int)ad.ldp, 1.0f / sqrtf((float)ad.dh), s)) return rc;
    if (int rc = attn_gemm(4, ad, nullptr, b.kv, w.dprob, w.dq, w, s)) return rc;              // dQ = dS K
    if (int rc = attn_gemm(5, ad, b.q, nullptr, w.dprob, w.dkv, w, s)) return rc;              // dK = dS^T Q
    if (int rc = linear_bwd(xin, P + a.in_w, w.dq, G + a.in_w, G + a.in_b, d2, 1, R1, H, H, w, s, acc)) return rc;  // ... + through q
    if (int rc = linear_bwd(b.comb, P + a.in_w + (int64_t)H * H, w.dkv, G + a.in_w + (int64_t)H * H, G + a.in_b + H, w.dcomb, 0, B * Sk, 2 * H, H,
                            w, s, acc))
      return rc;
    hipLaunchKernelGGL(k_split_comb, dim3(grid_for((int64_t)B * Sk * H)), dim3(256), 0, s, w.dcomb, w.dcp, d2, B, S1, S2, H);
    DQ_LAUNCH_CHECK();
    dx = d2; other = d1;
    if (on_bucket) {  // every kernel that writes this layer's gradient slice is enqueued
      int64_t off, cnt;
      tfm_bucket(*p, p->layers - 1 - l, &off, &cnt);
      on_bucket(user, p->layers - 1 - l, off, cnt);
    }
  }
  // x0 = rope(x_t Win^T + b) + temb
  if (int rc = launch_seqsum(dx, B, S1, H, w.dtemb, s)) return rc;
  if (int rc = launch_rope_add(dx, rope_sin, rope_cos, nullptr, B, S1, H, 1, s)) return rc;
  if (int rc = linear_bwd(x_t, P + p->in_w, dx, G + p->in_w, G + p->in_b, dx_t, 0, R1, H, D, w, s, acc)) return rc;
  if (int rc = launch_cond_embed_bwd(w.dcp, x_cond, P + p->c_w, rope_sin, rope_cos, G + p->c_w, G + p->c_b, dx_cond, w.colscr, B, S2, H, s, acc)) return rc;
  // time MLP
  if (int rc = linear_bwd(w.tg, P + p->t2_w, w.dtemb, G + p->t2_w, G + p->t2_b, w.dtg, 0, B, H, 4 * H, w, s, acc)) return rc;
  if (int rc = launch_gelu_bwd(w.th, w.dtg, w.dtg, (int64_t)B * 4 * H, s)) return rc;
  if (int rc = linear_bwd(w.tfeat, P + p->t1_w, w.dtg, G + p->t1_w, G + p->t1_b, nullptr, 0, B, 4 * H, H, w, s, acc)) return rc;
  if (on_bucket) {
    int64_t off, cnt;
    tfm_bucket(*p, p->layers, &off, &cnt);
    on_bucket(user, p->layers, off, cnt);
  }
  return 0;
}

extern "C" {

int dq_tfm_bwd(dq_tfm* p, const float* params, const float* rope_sin, const float* rope_cos, const float* x_t, const float* x_cond,
               const float* dout, float* grads, int accumulate, float* dx_t, float* dx_cond, void* workspace, int64_t workspace_bytes, int B,
               int S1, int S2, void* stream) {
  return tfm_bwd_impl(p, params, rope_sin, rope_cos, x_t, x_cond, dout, grads, accumulate, dx_t, dx_cond, workspace, workspace_bytes, B, S1,
                      S2, stream, nullptr, nullptr);
}

int dq_tfm_bwd_buckets(dq_tfm* p, const float* params, const float* rope_sin, const float* rope_cos, const float* x_t,
                       const float* x_cond, const float* dout, float* grads, int accumulate, float* dx_t, float* dx_cond, void* workspace,
                       int64_t workspace_bytes, int B, int S1, int S2, void* stream, dq_tfm_bucket_fn on_bucket, void* user) {
  return tfm_bwd_impl(p, params, rope_sin, rope_cos, x_t, x_cond, dout, grads, accumulate, dx_t, dx_cond, workspace, workspace_bytes, B, S1,
                      S2, stream, on_bucket, user);
}

int dq_tfm_num_buckets(const dq_tfm* p) { return p ? p->layers + 1 : 0; }

int dq_tfm_bucket_info(const dq_tfm* p, int i, int64_t* offset, int64_t* count) {
  DQ_REQUIRE(p && offset && count && i >= 0 && i <= p->layers, "dq_tfm_bucket_info: bucket index out of range");
  tfm_bucket(*p, i, offset, count);
  return 0;
}

}  // extern "C"
