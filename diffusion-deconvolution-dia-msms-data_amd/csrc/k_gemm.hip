// Exact-fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32, 256 FLOP/cycle/CU = 157 TFLOP/s dense peak) for the dense
// layers of the reference's CustomTransformer (building_blocks.py: nn.Linear at :86-88, 141-145, 211-213; nn.MultiheadAttention's
// projections and its two batched products at :136-138) and for their gradients.  See dq_tfm.h for the operand conventions.
//
// Tiling: a block of 4 waves owns a (BM x BN) tile of C and walks the reduction in steps of BK = 32.  Each wave keeps
// TM x TN accumulator tiles of 32x32 (16 registers each).  Global -> registers -> LDS with the next step's global loads in
// flight during this step's MFMAs (two LDS buffers, one barrier per step).  The reduction order inside a step is free, so a
// lane half h takes the four consecutive k = 8j + 4h .. + 3: for an operand whose rows are contiguous along k, the LDS tile
// keeps the global layout ([row][k], row stride 36 floats -> conflict-free 16-byte reads) and one ds_read_b128 feeds four MFMAs;
// an operand stored the other way round is kept as [k][col] and read one value per MFMA (conflict-free across the lanes).
// Split-K (small M x N against a long reduction: the 40000-wide input projection, the dX of the output projection) writes
// per-split partial tiles that a second kernel sums in a fixed order -- no atomics, bitwise repeatable.
#include "dq_common.h"
#include "dq_mfma.h"
#include "dq_tfm.h"
#include <algorithm>
#include <cstdlib>

namespace dq {

namespace {
constexpr int BK = 32;
constexpr int LDK = BK + 4;

struct GemmK {
  const float* A; const float* B; float* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int batch, inner;
  int64_t sAo, sAi, sBo, sBi, sCo, sCi;
  const float* bias; float alpha; int accumulate;
  int splits, k_per_split; float* partial;
};

// one (R rows/cols x 32 k) operand tile: global -> registers (float4 units), registers -> LDS
template <bool KMAJOR, int R, int NT>
struct TileIO {
  static constexpr int U = R * 8 / NT;  // float4 units per thread
  static_assert(R * 8 % NT == 0, "tile does not divide over the block");
  float4 v[U];
  __device__ __forceinline__ void load(const float* __restrict__ base, int64_t ld, int r0, int rmax, int k0, int kend, int tid) {
    if (r0 + R <= rmax && k0 + BK <= kend) {  // interior tile (block-uniform): straight vector loads, no per-unit branches
#pragma unroll
      for (int i = 0; i < U; ++i) {
        const int u = tid + NT * i;
        const float* p = KMAJOR ? base + (int64_t)(r0 + (u >> 3)) * ld + k0 + 4 * (u & 7)
                                : base + (int64_t)(k0 + u / (R / 4)) * ld + r0 + 4 * (u % (R / 4));
        v[i] = *reinterpret_cast<const float4*>(p);
      }
      return;
    }
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const int u = tid + NT * i;
      float4 x = {0.f, 0.f, 0.f, 0.f};
      if (KMAJOR) {
        const int row = r0 + (u >> 3), k = k0 + 4 * (u & 7);
        if (row < rmax) {
          const float* p = base + (int64_t)row * ld + k;
          if (k + 3 < kend) x = *reinterpret_cast<const float4*>(p);
          else {
            if (k < kend) x.x = p[0];
            if (k + 1 < kend) x.y = p[1];
            if (k + 2 < kend) x.z = p[2];
          }
        }
      } else {
        const int k = k0 + u / (R / 4), c = r0 + 4 * (u % (R / 4));
        if (k < kend) {
          const float* p = base + (int64_t)k * ld + c;
          if (c + 3 < rmax) x = *reinterpret_cast<const float4*>(p);
          else {
            if (c < rmax) x.x = p[0];
            if (c + 1 < rmax) x.y = p[1];
            if (c + 2 < rmax) x.z = p[2];
          }
        }
      }
      v[i] = x;
    }
  }
  __device__ __forceinline__ void store(float* lds, int tid) const {
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const int u = tid + NT * i;
      if (KMAJOR) *reinterpret_cast<float4*>(lds + (u >> 3) * LDK + 4 * (u & 7)) = v[i];
      else *reinterpret_cast<float4*>(lds + (u / (R / 4)) * (R + 4) + 4 * (u % (R / 4))) = v[i];
    }
  }
};

template <bool A_K, bool B_K, int TM, int TN, int WM, int WN>
__global__ void __launch_bounds__(WM * WN * 64) k_gemm(GemmK g) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  __shared__ __attribute__((aligned(16))) float as[2][BM * LDK];
  __shared__ __attribute__((aligned(16))) float bs[2][BN * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, half = lane >> 5;
  const int wm0 = (wv / WN) * TM * 32, wn0 = (wv % WN) * TN * 32;
  // XCD-aware tile order: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own L2), so id -> (id % 8) * (T / 8)
  // + id / 8 gives every XCD a contiguous run of tiles (m fastest: neighbours share the B panel, then the A panel) instead of
  // every 8th one
  int tile_id = blockIdx.x + gridDim.x * blockIdx.y;
  {
    const int T = gridDim.x * gridDim.y, per = T >> 3;
    if (tile_id < per * 8) tile_id = (tile_id & 7) * per + (tile_id >> 3);
  }
  const int m0 = (tile_id % (int)gridDim.x) * BM, n0 = (tile_id / (int)gridDim.x) * BN;
  const int z = blockIdx.z % g.batch, sp = blockIdx.z / g.batch;
  const int zo = z / g.inner, zi = z % g.inner;
  const float* A = g.A + zo * g.sAo + zi * g.sAi;
  const float* B = g.B + zo * g.sBo + zi * g.sBi;
  const int kbeg = sp * g.k_per_split, kend = min(g.K, kbeg + g.k_per_split);
  const int nkt = (kend - kbeg + BK - 1) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x16{0};

  TileIO<A_K, BM, NT> ta;
  TileIO<B_K, BN, NT> tb;
  if (nkt > 0) {
    ta.load(A, g.lda, m0, g.M, kbeg, kend, tid);
    tb.load(B, g.ldb, n0, g.N, kbeg, kend, tid);
    ta.store(as[0], tid);
    tb.store(bs[0], tid);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = kt + 1 < nkt;
    if (more) {
      ta.load(A, g.lda, m0, g.M, kbeg + (kt + 1) * BK, kend, tid);
      tb.load(B, g.ldb, n0, g.N, kbeg + (kt + 1) * BK, kend, tid);
    }
    const float* al = as[kt & 1];
    const float* bl = bs[kt & 1];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float av[TM][4], bv[TN][4];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (A_K) {
          const float4 t = *reinterpret_cast<const float4*>(al + (wm0 + 32 * i + col) * LDK + 8 * j + 4 * half);
          av[i][0] = t.x; av[i][1] = t.y; av[i][2] = t.z; av[i][3] = t.w;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) av[i][q] = al[(8 * j + 4 * half + q) * (BM + 4) + wm0 + 32 * i + col];
        }
      }
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        if (B_K) {
          const float4 t = *reinterpret_cast<const float4*>(bl + (wn0 + 32 * i + col) * LDK + 8 * j + 4 * half);
          bv[i][0] = t.x; bv[i][1] = t.y; bv[i][2] = t.z; bv[i][3] = t.w;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) bv[i][q] = bl[(8 * j + 4 * half + q) * (BN + 4) + wn0 + 32 * i + col];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) acc[i][jn] = mfma_f32(av[i][q], bv[jn][q], acc[i][jn]);
    }
    if (more) {
      ta.store(as[(kt + 1) & 1], tid);
      tb.store(bs[(kt + 1) & 1], tid);
    }
    __syncthreads();
  }

  // register r of lane (col, half) holds C[row = rmap(r, half)][col] of its 32x32 tile
  float* C = g.splits > 1 ? g.partial + ((int64_t)sp * g.batch + z) * g.M * g.N : g.C + zo * g.sCo + zi * g.sCi;
  const int64_t ldc = g.splits > 1 ? g.N : g.ldc;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      const int n = n0 + wn0 + 32 * jn + col;
      if (n >= g.N) continue;
      const float bias = (g.splits == 1 && g.bias) ? g.bias[n] : 0.f;
      const bool rmw = g.splits == 1 && g.accumulate;
      // all 16 reads of a += tile are issued before the first store (a load behind a store to a pointer the compiler cannot
      // tell apart would wait for it: 64 serial round trips per lane)
      float old[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + 32 * i + rmap(r, half);
        old[r] = (rmw && m < g.M) ? C[(int64_t)m * ldc + n] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + 32 * i + rmap(r, half);
        if (m >= g.M) continue;
        float* dst = C + (int64_t)m * ldc + n;
        if (g.splits > 1) *dst = acc[i][jn][r];
        else *dst = old[r] + (g.alpha * acc[i][jn][r] + bias);
      }
    }
}

// C = alpha * (sum over the splits, in order) + bias (+ C)
__global__ void __launch_bounds__(256) k_gemm_split_reduce(GemmK g) {
  const int64_t per = (int64_t)g.M * g.N, total = per * g.batch;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    float s = 0.f;
    int sp = 0;
    for (; sp + 8 <= g.splits; sp += 8) {  // eight loads in flight, summed in split order
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = g.partial[(int64_t)(sp + k) * total + e];
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; sp < g.splits; ++sp) s += g.partial[(int64_t)sp * total + e];
    const int z = (int)(e / per);
    const int64_t mn = e - (int64_t)z * per;
    const int m = (int)(mn / g.N), n = (int)(mn - (int64_t)m * g.N);
    float* dst = g.C + (z / g.inner) * g.sCo + (z % g.inner) * g.sCi + (int64_t)m * g.ldc + n;
    const float v = g.alpha * s + (g.bias ? g.bias[n] : 0.f);
    *dst = g.accumulate ? *dst + v : v;
  }
}

struct Shape { int bm, bn, splits, k_per_split; };
// Tile height and split-K factor by a small cost model of the 256-CU machine: blocks are dealt to the CUs in rounds, a block
// costs (k-steps x MFMA cycles of its tile + a fixed prologue / epilogue), a split adds the second kernel and its traffic.
// What it buys over "largest tile, no split": 272 or 288 tiles of 128 x 128 (the transformer's 1088- and 2176-row layers) are
// two rounds with the second one almost empty; 64-row tiles or a split fill it.  Splits are only considered while
// tiles x splits <= 1024, which bounds the scratch at 1024 tiles of 128 x 128 floats.
Shape choose(int M, int N, int K, int batch, int forced_splits) {
  constexpr int CUS = 256;
  const int kt_all = cdiv(K, BK);
  Shape best{};
  double best_cost = 1e30;
  static const int force_bm = [] { const char* e = std::getenv("DQ_GEMM_BM"); return e ? std::atoi(e) : 0; }();
  const int bms[4] = {256, 128, 64, 32};
  const double tile_cycles[4] = {7000.0, 4096.0, 2400.0, 1500.0};  // per k-step of 32: 128 / 64 / 32 / 16 MFMAs per wave
  for (int c = 0; c < 4; ++c) {
    if (force_bm ? bms[c] != force_bm : c == 0) continue;
    const int bm = bms[c];
    if (!force_bm && c < 3 && M <= bms[c + 1]) continue;  // a smaller tile covers all rows: the larger one only adds padding
    const int64_t tiles = (int64_t)cdiv(M, bm) * cdiv(N, 128) * batch;
    const int smax = forced_splits > 0 ? forced_splits : (int)std::min<int64_t>(std::min<int64_t>(64, kt_all), std::max<int64_t>(1, 1024 / tiles));
    for (int sp = forced_splits > 0 ? forced_splits : 1; sp <= smax; ++sp) {
      const int kps = cdiv(cdiv(K, sp), BK) * BK;
      const int real = cdiv(K, kps);
      if (real != sp && forced_splits <= 0) continue;  // same partition as a smaller factor
      const double rounds = (double)cdiv(tiles * real, CUS);
      double cost = rounds * ((kps / BK) * tile_cycles[c] + 2500.0);
      if (real > 1) cost += 12000.0 + (double)real * M * N * batch * 4.0 / 1250.0;
      if (cost < best_cost) { best_cost = cost; best.bm = bm; best.bn = 128; best.splits = real; best.k_per_split = kps; }
    }
  }
  return best;
}

template <bool A_K, bool B_K>
int launch_layout(const GemmK& k, const Shape& sh, hipStream_t s) {
  const dim3 grid(cdiv(k.M, sh.bm), cdiv(k.N, sh.bn), k.batch * k.splits);
  if (sh.bm == 32) hipLaunchKernelGGL((k_gemm<A_K, B_K, 1, 1, 1, 4>), grid, dim3(256), 0, s, k);
  else if (sh.bm == 64) hipLaunchKernelGGL((k_gemm<A_K, B_K, 1, 2, 2, 2>), grid, dim3(256), 0, s, k);
  else if (sh.bm == 256) hipLaunchKernelGGL((k_gemm<A_K, B_K, 4, 2, 2, 2>), grid, dim3(256), 0, s, k);
  else hipLaunchKernelGGL((k_gemm<A_K, B_K, 2, 2, 2, 2>), grid, dim3(256), 0, s, k);
  DQ_LAUNCH_CHECK();
  return 0;
}
}  // namespace

int64_t gemm_partial_floats(int M, int N, int K, int batch) {
  const Shape sh = choose(M, N, K, batch, 0);
  return sh.splits > 1 ? (int64_t)sh.splits * batch * M * N : 0;
}

int launch_gemm(const Gemm& g, hipStream_t s) {
  DQ_REQUIRE(g.A && g.B && g.C, "gemm: missing operand");
  if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return 0;
  DQ_REQUIRE(g.K > 0 && g.inner > 0, "gemm: bad reduction length / batch split");
  DQ_REQUIRE(g.lda % 4 == 0 && g.ldb % 4 == 0, "gemm: leading dimensions of A and B must be multiples of 4 floats");
  DQ_REQUIRE(((uintptr_t)g.A & 15) == 0 && ((uintptr_t)g.B & 15) == 0, "gemm: A and B must be 16-byte aligned");
  DQ_REQUIRE((g.sAo % 4 == 0) && (g.sAi % 4 == 0) && (g.sBo % 4 == 0) && (g.sBi % 4 == 0), "gemm: batch strides of A and B must be multiples of 4");
  const Shape sh = choose(g.M, g.N, g.K, g.batch, g.splits);
  GemmK k;
  k.A = g.A; k.B = g.B; k.C = g.C; k.M = g.M; k.N = g.N; k.K = g.K; k.lda = g.lda; k.ldb = g.ldb; k.ldc = g.ldc;
  k.batch = g.batch; k.inner = g.inner; k.sAo = g.sAo; k.sAi = g.sAi; k.sBo = g.sBo; k.sBi = g.sBi; k.sCo = g.sCo; k.sCi = g.sCi;
  k.bias = g.bias; k.alpha = g.alpha; k.accumulate = g.accumulate;
  k.splits = sh.splits; k.k_per_split = sh.k_per_split; k.partial = g.partial;
  if (sh.splits > 1)
    DQ_REQUIRE(g.partial && g.partial_floats >= (int64_t)sh.splits * g.batch * g.M * g.N, "gemm: split-K scratch missing or too small");
  DQ_REQUIRE((int64_t)k.batch * k.splits <= 65535, "gemm: batch x splits exceeds the grid");
  int rc;
  if (g.a_kmajor && g.b_kmajor) rc = launch_layout<true, true>(k, sh, s);
  else if (g.a_kmajor && !g.b_kmajor) rc = launch_layout<true, false>(k, sh, s);
  else if (!g.a_kmajor && !g.b_kmajor) rc = launch_layout<false, false>(k, sh, s);
  else { set_error("gemm: the (A transposed, B k-major) layout is not built"); return 2; }
  if (rc) return rc;
  if (sh.splits > 1) {
    const int64_t total = (int64_t)g.M * g.N * g.batch;
    hipLaunchKernelGGL(k_gemm_split_reduce, dim3((unsigned)std::min<int64_t>(cdiv(total, 256), 4096)), dim3(256), 0, s, k);
    DQ_LAUNCH_CHECK();
  }
  return 0;
}

}  // namespace dq
