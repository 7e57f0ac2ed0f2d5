// Internal launch API of the CustomTransformer path of libdq_hip.so (reference: dquartic/model/building_blocks.py).
// Every launcher is asynchronous on the given stream, allocates nothing, returns 0 on success (non-zero after set_error()).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dq {

// ---- k_gemm.hip: C[z] (+)= alpha * op(A[z]) op(B[z]) (+ bias) in exact fp32 on v_mfma_f32_32x32x2_f32.
// A is the (M x K) operand, B the (K x N) operand, both addressed through a layout flag:
//   a_kmajor = 1: A(m, k) = A[m * lda + k]   (rows of A contiguous along the reduction; torch activations x of y = x W^T)
//   a_kmajor = 0: A(m, k) = A[k * lda + m]   (A stored transposed: the dY of dW = dY^T X)
//   b_kmajor = 1: B(k, n) = B[n * ldb + k]   (an nn.Linear weight (N, K) used as W^T)
//   b_kmajor = 0: B(k, n) = B[k * ldb + n]   (a row-major (K, N) matrix: the W of dX = dY W, the V of P V)
// C(m, n) = C[m * ldc + n].  Batched over z = zo * inner + zi with element offsets zo * s?o + zi * s?i per operand
// (attention: zo = sample, zi = head).  K beyond the split / M, N beyond the tile are zero-filled / masked.
// Requirements (checked): lda, ldb, K multiples of 4 and 16-byte aligned bases (vector loads along the contiguous axis).
// arithmetic of a product: exact fp32 on v_mfma_f32_32x32x2_f32, or three bf16 passes over split operands (k_gemm.hip: k_gemm_s3)
enum GemmPrecision { GEMM_FP32 = 0, GEMM_BF16X3 = 1 };
struct Gemm {
  const float* A = nullptr; const float* B = nullptr; float* C = nullptr;
  int M = 0, N = 0, K = 0;
  int64_t lda = 0, ldb = 0, ldc = 0;
  int a_kmajor = 1, b_kmajor = 1;
  int batch = 1, inner = 1;
  int64_t sAo = 0, sAi = 0, sBo = 0, sBi = 0, sCo = 0, sCi = 0;
  // C = sum over kbatch blocks of A_b B_b^T (A_b = A + b sAk, B_b = B + b sBk; each with reduction length K): a weight gradient summed
  // over the samples of a batch in ONE pass over C instead of one read-modify-write of C per sample
  int kbatch = 1; int64_t sAk = 0, sBk = 0;
  const float* bias = nullptr;  // (N): added to every row
  const float* bias_m = nullptr;  // (M): added to every column (a conv bias: rows are output channels); unsplit launches only
  float alpha = 1.f;
  int accumulate = 0;           // C += instead of C =
  const float* add = nullptr;   // C = add + (...): a residual read from another tensor laid out like C (unsplit launches only)
  // split-K: the reduction is cut into `splits` ranges whose partial products go to `partial` ([split][z][M][N] floats) and
  // are summed in a fixed order by a second kernel (deterministic, no atomics).  splits = 0: chosen by the launcher.
  int splits = 0; float* partial = nullptr; int64_t partial_floats = 0;
  int precision = -1;  // GemmPrecision, or -1 = the calling thread's default (set_gemm_precision; GEMM_FP32 unless changed)
};
int launch_gemm(const Gemm& g, hipStream_t s);
int set_gemm_precision(int precision);  // thread-local default of launch_gemm; returns the previous one
int64_t gemm_partial_floats(int M, int N, int K, int batch);  // upper bound of what launch_gemm will ask of `partial`

// ---- k_tfm.hip: the pointwise / row-wise kernels around the GEMMs
// x (rows = B*S, H) in place: RoPE over adjacent channel pairs with the host tables sin/cos (S, H/2), then += temb[b] (nullable)
int launch_rope_add(float* x, const float* sin_t, const float* cos_t, const float* temb, int B, int S, int H, int inverse, hipStream_t s);
// c[b][s][:] = rope(x_cond[b][s] * w + bias)
int launch_cond_embed(const float* x_cond, const float* w, const float* bias, const float* sin_t, const float* cos_t, float* c, int B, int S,
                      int H, hipStream_t s);
// backward of launch_cond_embed: dw, db (+=, H each) and dx_cond (nullable, (B, S)) from dc (B, S, H); scratch: 2 * H * 64 floats
int launch_cond_embed_bwd(const float* dc, const float* x_cond, const float* w, const float* sin_t, const float* cos_t, float* dw, float* db,
                          float* dx_cond, float* scratch, int B, int S, int H, hipStream_t s, int accumulate = 1);
// e[b][:] = [sin(t_b f) | cos(t_b f)], f: (H/2) host table
int launch_time_features(const int64_t* t, const float* freqs, float* e, int B, int H, hipStream_t s);
int launch_gelu(const float* x, float* y, int64_t n, hipStream_t s);
int launch_gelu_bwd(const float* x, const float* dy, float* dx, int64_t n, hipStream_t s);  // dx = dy * gelu'(x)
// y = x + r (saved, nullable r), out = LayerNorm(y) * g + b ; stats (rows, 2) = mean, rstd
int launch_layernorm_fwd(const float* x, const float* r, const float* g, const float* b, float* y, float* out, float* stats, int rows, int H,
                         hipStream_t s);
// dy (+)= d LayerNorm ; dg, db += ; scratch: 2 * H * LN_BWD_BLOCKS floats
constexpr int LN_BWD_BLOCKS = 256;
int launch_layernorm_bwd(const float* y, const float* stats, const float* g, const float* dout, float* dy, float* dg, float* db, float* scratch,
                         int rows, int H, hipStream_t s, int accumulate = 1);
// rows of length n (row stride ld): p = softmax(scale * p) in place ; ds = p * (dp - sum(p dp)) * scale in place of dp
int launch_softmax_rows(float* p, int64_t rows, int n, int ld, float scale, hipStream_t s);
int launch_softmax_rows_bwd(const float* p, float* dp, int64_t rows, int n, int ld, float scale, hipStream_t s);
// out[n] (+)= sum_m x[m * ld + n]  (bias gradients; accumulate = 0: plain store); scratch: COLSUM_BLOCKS * N floats
constexpr int COLSUM_BLOCKS = 64;
int launch_colsum(const float* x, int M, int N, int64_t ld, float* out, float* scratch, hipStream_t s, int accumulate = 1);
// out[b][n] = sum_s x[(b * S + s) * N + n]  (time-embedding gradient: plain store)
int launch_seqsum(const float* x, int B, int S, int N, float* out, hipStream_t s);

}  // namespace dq
