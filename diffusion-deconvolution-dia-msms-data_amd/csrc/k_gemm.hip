// Exact-fp32 GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32, 256 FLOP/cycle/CU = 157 TFLOP/s dense peak) for the dense
// layers of the reference's CustomTransformer (building_blocks.py: nn.Linear at :86-88, 141-145, 211-213; nn.MultiheadAttention's
// projections and its two batched products at :136-138) and for their gradients.  See dq_tfm.h for the operand conventions.
//
// Tiling: a block owns a (BM x 128) tile of C -- 128 rows on 8 waves, 64 / 32 rows on 4 -- and walks the reduction in steps of
// BK = 32.  Each wave keeps TM x TN accumulator tiles of 32x32 (16 registers each).  Global -> registers -> LDS with the next
// step's global loads in flight during this step's MFMAs (two LDS buffers, one barrier per step).  A product runs as the whole
// rounds of 256 tiles unsplit plus the remaining tiles with a split reduction (choose(), below).  The reduction order inside a step is free, so a
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
#ifndef DQ_GEMM_BK
#define DQ_GEMM_BK 32
#endif
constexpr int BK = DQ_GEMM_BK;  // reduction step (floats); Q4 = float4 units per tile row
constexpr int Q4 = BK / 4;
constexpr int LDK = BK + 4;

struct GemmK {
  const float* A; const float* B; float* C;
  int M, N, K;
  int64_t lda, ldb, ldc;
  int batch, inner;
  int64_t sAo, sAi, sBo, sBi, sCo, sCi;
  const float* bias; const float* bias_m; float alpha; int accumulate;
  const float* add;  // nullable: C = add + (...) with add laid out like C (unsplit launches)
  int splits, k_per_split; float* partial;
  int tile_base, mt;  // first tile of this launch; m-tiles of the whole product (tile id = m-tile + mt * n-tile)
  // reduction over kb blocks of K (C = sum_b A_b B_b^T with A_b = A + b sAk, B_b = B + b sBk): the k axis the splits cut is the virtual
  // axis of kb * kp elements, kp = K rounded up to the k-tile, so that a k-tile never straddles two blocks
  int kb, kp; int64_t sAk, sBk;
};

// one (R rows/cols x 32 k) operand tile: global -> registers (float4 units), registers -> LDS
template <bool KMAJOR, int R, int NT>
struct TileIO {
  static constexpr int U = R * Q4 / NT;  // float4 units per thread
  static_assert(R * Q4 % NT == 0, "tile does not divide over the block");
  float4 v[U];
  __device__ __forceinline__ void load(const float* __restrict__ base, int64_t ld, int r0, int rmax, int k0, int kend, int tid) {
    if (r0 + R <= rmax && k0 + BK <= kend) {  // interior tile (block-uniform): straight vector loads, no per-unit branches
#pragma unroll
      for (int i = 0; i < U; ++i) {
        const int u = tid + NT * i;
        const float* p = KMAJOR ? base + (int64_t)(r0 + (u / Q4)) * ld + k0 + 4 * (u % Q4)
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
        const int row = r0 + (u / Q4), k = k0 + 4 * (u % Q4);
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
      if (KMAJOR) *reinterpret_cast<float4*>(lds + (u / Q4) * LDK + 4 * (u % Q4)) = v[i];
      else *reinterpret_cast<float4*>(lds + (u / (R / 4)) * (R + 4) + 4 * (u % (R / 4))) = v[i];
    }
  }
};

// the accumulator tiles of one wave -> C (or the split's partial tile).  Register r of lane (col, half) holds
// C[row = rmap(r, half)][col] of its 32x32 tile.
template <int TM, int TN, int BM, int BN>
__device__ __forceinline__ void gemm_epilogue(const GemmK& g, f32x16 (&acc)[TM][TN], int tile_local, int m0, int n0, int wm0, int wn0, int z,
                                              int zo, int zi, int sp, int col, int half) {
  if (g.splits > 1) {  // partial tile of this split, tile-local layout [split][z][tile of the launch][BM][BN]
    float* P = g.partial + (((int64_t)sp * g.batch + z) * gridDim.x + tile_local) * (BM * BN);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn)
#pragma unroll
        for (int r = 0; r < 16; ++r) P[(wm0 + 32 * i + rmap(r, half)) * BN + wn0 + 32 * jn + col] = acc[i][jn][r];
    return;
  }
  float* C = g.C + zo * g.sCo + zi * g.sCi;
  const int64_t ldc = g.ldc;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      const int n = n0 + wn0 + 32 * jn + col;
      if (n >= g.N) continue;
      const float bias = g.bias ? g.bias[n] : 0.f;
      const bool rmw = g.accumulate != 0 || g.add != nullptr;
      const float* Cold = g.add ? g.add + zo * g.sCo + zi * g.sCi : C;  // (a residual read from another tensor instead of a copy + "+=")
      // all 16 reads of a += tile are issued before the first store (a load behind a store to a pointer the compiler cannot
      // tell apart would wait for it: 64 serial round trips per lane)
      float old[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + 32 * i + rmap(r, half);
        old[r] = (rmw && m < g.M) ? Cold[(int64_t)m * ldc + n] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm0 + 32 * i + rmap(r, half);
        if (m >= g.M) continue;
        C[(int64_t)m * ldc + n] = old[r] + (g.alpha * acc[i][jn][r] + bias + (g.bias_m ? g.bias_m[m] : 0.f));
      }
    }
}

template <bool A_K, bool B_K, int TM, int TN, int WM, int WN>
__global__ void __launch_bounds__(WM * WN * 64) k_gemm(GemmK g) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  // either layout fits: [rows][BK + 4] or [BK][rows + 4].  (BK = 64 was measured: one block per CU instead of two, slower.)
  __shared__ __attribute__((aligned(16))) float as[2][BM * LDK > BK * (BM + 4) ? BM * LDK : BK * (BM + 4)];
  __shared__ __attribute__((aligned(16))) float bs[2][BN * LDK > BK * (BN + 4) ? BN * LDK : BK * (BN + 4)];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, half = lane >> 5;
  const int wm0 = (wv / WN) * TM * 32, wn0 = (wv % WN) * TN * 32;
  // XCD-aware tile order: consecutive workgroup ids go round-robin to the 8 XCDs (each with its own L2), so id -> (id % 8) * (T / 8)
  // + id / 8 gives every XCD a contiguous run of tiles (m fastest: neighbours share the B panel, then the A panel) instead of
  // every 8th one
  // A launch covers the tiles [tile_base, tile_base + gridDim.x) of the (mt x nt) tile grid.
  int tile_local = blockIdx.x;
  {
    const int T = gridDim.x, per = T >> 3;
    if (tile_local < per * 8) tile_local = (tile_local & 7) * per + (tile_local >> 3);
  }
  const int tile_id = g.tile_base + tile_local;
  const int m0 = (tile_id % g.mt) * BM, n0 = (tile_id / g.mt) * BN;
  const int z = blockIdx.y % g.batch, sp = blockIdx.y / g.batch;
  const int zo = z / g.inner, zi = z % g.inner;
  const float* A = g.A + zo * g.sAo + zi * g.sAi;
  const float* B = g.B + zo * g.sBo + zi * g.sBi;
  const int kvlen = g.kb > 1 ? g.kb * g.kp : g.K;
  const int kbeg = sp * g.k_per_split, kend = min(kvlen, kbeg + g.k_per_split);
  const int nkt = (kend - kbeg + BK - 1) / BK;
  // k-tile at virtual position kv -> (operand block, k inside it, the bound its loads are masked at)
  auto kt_block = [&](int kv) { return g.kb > 1 ? kv / g.kp : 0; };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x16{0};

  TileIO<A_K, BM, NT> ta;
  TileIO<B_K, BN, NT> tb;
  if (nkt > 0) {
    const int bq = kt_block(kbeg);
    ta.load(A + bq * g.sAk, g.lda, m0, g.M, kbeg - bq * g.kp, min(g.K, kend - bq * g.kp), tid);
    tb.load(B + bq * g.sBk, g.ldb, n0, g.N, kbeg - bq * g.kp, min(g.K, kend - bq * g.kp), tid);
    ta.store(as[0], tid);
    tb.store(bs[0], tid);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = kt + 1 < nkt;
    if (more) {
      const int kv = kbeg + (kt + 1) * BK, bq = kt_block(kv);
      ta.load(A + bq * g.sAk, g.lda, m0, g.M, kv - bq * g.kp, min(g.K, kend - bq * g.kp), tid);
      tb.load(B + bq * g.sBk, g.ldb, n0, g.N, kv - bq * g.kp, min(g.K, kend - bq * g.kp), tid);
    }
    const float* al = as[kt & 1];
    const float* bl = bs[kt & 1];
#pragma unroll
    for (int j = 0; j < BK / 8; ++j) {
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

  gemm_epilogue<TM, TN, BM, BN>(g, acc, tile_local, m0, n0, wm0, wn0, z, zo, zi, sp, col, half);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same product on the bf16 matrix path, THREE passes over split operands ("bf16x3"): every fp32 operand value x is split into
// hi = bf16(x) and lo = bf16(x - hi) as its tile is written to LDS, and a (32 x 32 x 16) step runs hi.hi + hi.lo + lo.hi on
// v_mfma_f32_32x32x16_bf16 with fp32 accumulation (lo.lo, ~2^-16 of a term, is dropped).  An operand keeps ~16 mantissa bits
// instead of fp32's 24, so a product term carries a relative error of ~1e-5 -- NOT the exact-fp32 arithmetic of k_gemm: this is a
// separate precision mode with its own stated tolerance, selected per transformer handle
// (dq_tfm_set_precision) or through dq_gemm_bf16x3.  Cost per 32-deep step and 32 x 32 tile: 6 MFMAs of 32 cycles against 16 of 64.
// LDS: both operands as [row][k] bf16 planes (hi and lo) whatever their layout in memory, row pitch 40 bf16 = 80 B, so that a
// lane's fragment (8 consecutive k of its row: A[row r][8h + j], B[8h + j][col r]) is ONE ds_read_b128 per plane.
// ---------------------------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
constexpr int BKP = BK + 8;

// One (R rows x 32 k) operand tile of the split kernel: global -> registers -> the two bf16 planes [row][k].  A thread always ends up
// with FOUR CONSECUTIVE k of ONE row (8-byte plane stores): from one float4 when the rows are contiguous along k, from four dword
// loads (each coalesced over the wave: consecutive lanes = consecutive rows) when memory runs along the rows instead -- so the
// transposition of such an operand costs narrower global loads, not 2-byte scattered LDS stores.  (Measured against float4 loads +
// a 4 x 4 DPP transpose inside each lane quad: the transformer's batch-32 train step 15.4 ms this way, 16.7 ms with the DPP form.)
template <bool KMAJOR, int R, int NT>
struct SplitIO {
  static constexpr int U = R * Q4 / NT;  // (row, k-quad) units per thread
  static_assert(R * Q4 % NT == 0, "tile does not divide over the block");
  float4 v[U];
  // unit u -> (row, k quad): k-major: rows of Q4 quads ; otherwise: quads of R rows (consecutive threads = consecutive rows)
  static __device__ __forceinline__ int row_of(int u) { return KMAJOR ? u / Q4 : u % R; }
  static __device__ __forceinline__ int quad_of(int u) { return KMAJOR ? u % Q4 : u / R; }
  __device__ __forceinline__ void load(const float* __restrict__ base, int64_t ld, int r0, int rmax, int k0, int kend, int tid) {
    const bool interior = r0 + R <= rmax && k0 + BK <= kend;  // block-uniform
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const int u = tid + NT * i;
      const int row = r0 + row_of(u), k = k0 + 4 * quad_of(u);
      float4 x = {0.f, 0.f, 0.f, 0.f};
      if (KMAJOR) {
        const float* p = base + (int64_t)row * ld + k;
        if (interior || (row < rmax && k + 3 < kend)) x = *reinterpret_cast<const float4*>(p);
        else if (row < rmax) {
          if (k < kend) x.x = p[0];
          if (k + 1 < kend) x.y = p[1];
          if (k + 2 < kend) x.z = p[2];
        }
      } else {
        const float* p = base + (int64_t)k * ld + row;
        if (interior) { x.x = p[0]; x.y = p[ld]; x.z = p[2 * ld]; x.w = p[3 * ld]; }
        else if (row < rmax) {
          if (k < kend) x.x = p[0];
          if (k + 1 < kend) x.y = p[ld];
          if (k + 2 < kend) x.z = p[2 * ld];
          if (k + 3 < kend) x.w = p[3 * ld];
        }
      }
      v[i] = x;
    }
  }
  __device__ __forceinline__ void store(__bf16* __restrict__ hi, __bf16* __restrict__ lo, int tid) const {
#pragma unroll
    for (int i = 0; i < U; ++i) {
      const int u = tid + NT * i;
      const float x[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
      __bf16 h[4], l[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        h[q] = (__bf16)x[q];
        l[q] = (__bf16)(x[q] - (float)h[q]);
      }
      const int o = row_of(u) * BKP + 4 * quad_of(u);
      *reinterpret_cast<bf16x4*>(hi + o) = bf16x4{h[0], h[1], h[2], h[3]};
      *reinterpret_cast<bf16x4*>(lo + o) = bf16x4{l[0], l[1], l[2], l[3]};
    }
  }
};

template <bool A_K, bool B_K, int TM, int TN, int WM, int WN>
__global__ void __launch_bounds__(WM * WN * 64) k_gemm_s3(GemmK g) {
  constexpr int BM = WM * TM * 32, BN = WN * TN * 32, NT = WM * WN * 64;
  __shared__ __attribute__((aligned(16))) __bf16 ah[2][BM * BKP], al[2][BM * BKP], bh[2][BN * BKP], bl[2][BN * BKP];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, col = lane & 31, half = lane >> 5;
  const int wm0 = (wv / WN) * TM * 32, wn0 = (wv % WN) * TN * 32;
  int tile_local = blockIdx.x;
  {
    const int T = gridDim.x, per = T >> 3;
    if (tile_local < per * 8) tile_local = (tile_local & 7) * per + (tile_local >> 3);
  }
  const int tile_id = g.tile_base + tile_local;
  const int m0 = (tile_id % g.mt) * BM, n0 = (tile_id / g.mt) * BN;
  const int z = blockIdx.y % g.batch, sp = blockIdx.y / g.batch;
  const int zo = z / g.inner, zi = z % g.inner;
  const float* A = g.A + zo * g.sAo + zi * g.sAi;
  const float* B = g.B + zo * g.sBo + zi * g.sBi;
  const int kvlen = g.kb > 1 ? g.kb * g.kp : g.K;
  const int kbeg = sp * g.k_per_split, kend = min(kvlen, kbeg + g.k_per_split);
  const int nkt = (kend - kbeg + BK - 1) / BK;
  // k-tile at virtual position kv -> (operand block, k inside it, the bound its loads are masked at)
  auto kt_block = [&](int kv) { return g.kb > 1 ? kv / g.kp : 0; };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = f32x16{0};

  SplitIO<A_K, BM, NT> ta;
  SplitIO<B_K, BN, NT> tb;
  if (nkt > 0) {
    const int bq = kt_block(kbeg);
    ta.load(A + bq * g.sAk, g.lda, m0, g.M, kbeg - bq * g.kp, min(g.K, kend - bq * g.kp), tid);
    tb.load(B + bq * g.sBk, g.ldb, n0, g.N, kbeg - bq * g.kp, min(g.K, kend - bq * g.kp), tid);
    ta.store(ah[0], al[0], tid);
    tb.store(bh[0], bl[0], tid);
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    const bool more = kt + 1 < nkt;
    if (more) {
      const int kv = kbeg + (kt + 1) * BK, bq = kt_block(kv);
      ta.load(A + bq * g.sAk, g.lda, m0, g.M, kv - bq * g.kp, min(g.K, kend - bq * g.kp), tid);
      tb.load(B + bq * g.sBk, g.ldb, n0, g.N, kv - bq * g.kp, min(g.K, kend - bq * g.kp), tid);
    }
    const int cur = kt & 1;
#pragma unroll
    for (int jj = 0; jj < BK / 16; ++jj) {
      bf16x8 a_h[TM], a_l[TM], b_h[TN], b_l[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int o = (wm0 + 32 * i + col) * BKP + 16 * jj + 8 * half;
        a_h[i] = *reinterpret_cast<const bf16x8*>(ah[cur] + o);
        a_l[i] = *reinterpret_cast<const bf16x8*>(al[cur] + o);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int o = (wn0 + 32 * j + col) * BKP + 16 * jj + 8 * half;
        b_h[j] = *reinterpret_cast<const bf16x8*>(bh[cur] + o);
        b_l[j] = *reinterpret_cast<const bf16x8*>(bl[cur] + o);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_l[i], b_h[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h[i], b_l[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_h[i], b_h[j], acc[i][j], 0, 0, 0);
        }
    }
    if (more) {
      ta.store(ah[(kt + 1) & 1], al[(kt + 1) & 1], tid);
      tb.store(bh[(kt + 1) & 1], bl[(kt + 1) & 1], tid);
    }
    __syncthreads();
  }
  gemm_epilogue<TM, TN, BM, BN>(g, acc, tile_local, m0, n0, wm0, wn0, z, zo, zi, sp, col, half);
}

// C tile = alpha * (sum over the splits, in order) + bias (+ C); one block per (tile of the launch, z)
__global__ void __launch_bounds__(256) k_gemm_split_reduce(GemmK g, int ntiles, int bm) {
  const int t = blockIdx.x % ntiles, z = blockIdx.x / ntiles;
  const int tile_id = g.tile_base + t;
  const int m0 = (tile_id % g.mt) * bm, n0 = (tile_id / g.mt) * 128;
  const int64_t tile_floats = (int64_t)bm * 128, split_stride = (int64_t)g.batch * ntiles * tile_floats;
  const float* P = g.partial + ((int64_t)z * ntiles + t) * tile_floats;
  float* C = g.C + (z / g.inner) * g.sCo + (z % g.inner) * g.sCi;
  {
    const int e = blockIdx.y * 256 + threadIdx.x;  // one element per thread: (tiles x bm / 2) blocks keep the machine busy
    const int m = m0 + (e >> 7), n = n0 + (e & 127);
    if (m >= g.M || n >= g.N) return;
    float s = 0.f;
    int sp = 0;
    for (; sp + 8 <= g.splits; sp += 8) {  // eight loads in flight, summed in split order
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = P[(int64_t)(sp + k) * split_stride + e];
#pragma unroll
      for (int k = 0; k < 8; ++k) s += v[k];
    }
    for (; sp < g.splits; ++sp) s += P[(int64_t)sp * split_stride + e];
    float* dst = C + (int64_t)m * g.ldc + n;
    const float v = g.alpha * s + (g.bias ? g.bias[n] : 0.f) + (g.bias_m ? g.bias_m[m] : 0.f);
    *dst = g.accumulate ? *dst + v : v;
  }
}

// A product is run as up to two launches over disjoint tile ranges: the whole rounds of 256 tiles (one per CU) unsplit, and the
// remaining tiles -- or all of them when there are fewer than 256 -- with the reduction split so that they, too, cover the machine.
struct Part { int tile_base = 0, ntiles = 0, splits = 1, k_per_split = 0; };
struct Shape { int bm = 128; Part full, rest; };
constexpr int MAX_SPLIT_TILES = 1024;  // tiles x splits of a split launch: bounds the scratch at 1024 x 128 x 128 floats
// Tile height and the split of the remainder by a small cost model of the 256-CU machine: blocks are dealt to the CUs in rounds,
// a block costs (k-steps x MFMA cycles of its tile + a fixed prologue / epilogue), a split adds the second kernel and its
// traffic.  What it buys over "largest tile, no split": 272 or 288 tiles of 128 x 128 (the transformer's 2176- and 1088-row
// layers) would be two rounds with the second one almost empty.
Shape choose(int M, int N, int K, int batch, int forced_splits) {
  constexpr int CUS = 256;
  const int kt_all = cdiv(K, BK);
  Shape best;
  double best_cost = 1e30;
  // (256 x 128 tiles were measured too, on 4 and on 8 waves: never faster than 128 x 128 on this machine, not built)
  const int bms[3] = {128, 64, 32};
  const double tile_cycles[3] = {4096.0, 2400.0, 1500.0};  // per k-step of 32: 64 / 32 / 16 MFMAs per SIMD, smaller tiles less efficient
  for (int c = 0; c < 3; ++c) {
    const int bm = bms[c];
    if (c < 2 && M <= bms[c + 1]) continue;  // a smaller tile covers all rows: the larger one only adds padding
    const int tiles = cdiv(M, bm) * cdiv(N, 128);
    auto block = [&](int kps) { return (kps / BK) * tile_cycles[c] + 2500.0; };
    Shape sh;
    sh.bm = bm;
    double cost;
    if (forced_splits > 0 || batch > 1) {  // one launch over everything (batched products have short reductions)
      const int sp = forced_splits > 0 ? std::min(forced_splits, kt_all) : 1;
      const int kps = cdiv(cdiv(K, sp), BK) * BK;
      sh.rest.ntiles = tiles; sh.rest.k_per_split = kps; sh.rest.splits = cdiv(K, kps);
      cost = (double)cdiv((int64_t)tiles * batch * sh.rest.splits, CUS) * block(kps);
    } else {
      const int full = tiles / CUS * CUS, rem = tiles - full;
      sh.full.ntiles = full; sh.full.k_per_split = kt_all * BK;
      cost = (full / CUS) * block(kt_all * BK);
      if (rem) {
        double best_rem = 1e30;
        const int smax = std::max(1, std::min(std::min(64, kt_all), MAX_SPLIT_TILES / rem));
        for (int sp = 1; sp <= smax; ++sp) {
          const int kps = cdiv(cdiv(K, sp), BK) * BK, real = cdiv(K, kps);
          if (real != sp) continue;  // same partition as a smaller factor
          double cr = (double)cdiv(rem * real, CUS) * block(kps) + (full ? 2000.0 : 0.0);
          if (real > 1) cr += 12000.0 + (double)real * rem * bm * 128 * 4.0 / 1250.0;
          if (cr < best_rem) { best_rem = cr; sh.rest.tile_base = full; sh.rest.ntiles = rem; sh.rest.splits = real; sh.rest.k_per_split = kps; }
        }
        cost += best_rem;
      }
    }
    if (cost < best_cost) { best_cost = cost; best = sh; }
  }
  return best;
}

template <bool A_K, bool B_K>
int launch_part(GemmK k, const Part& p, int bm, hipStream_t s, int precision) {
  if (p.ntiles == 0) return 0;
  k.tile_base = p.tile_base; k.splits = p.splits; k.k_per_split = p.k_per_split;
  const dim3 grid(p.ntiles, k.batch * p.splits);
  if (precision == GEMM_BF16X3) {
    if (bm == 32) hipLaunchKernelGGL((k_gemm_s3<A_K, B_K, 1, 1, 1, 4>), grid, dim3(256), 0, s, k);
    else if (bm == 64) hipLaunchKernelGGL((k_gemm_s3<A_K, B_K, 1, 2, 2, 2>), grid, dim3(256), 0, s, k);
    else hipLaunchKernelGGL((k_gemm_s3<A_K, B_K, 1, 2, 4, 2>), grid, dim3(512), 0, s, k);
  } else if (bm == 32) hipLaunchKernelGGL((k_gemm<A_K, B_K, 1, 1, 1, 4>), grid, dim3(256), 0, s, k);
  else if (bm == 64) hipLaunchKernelGGL((k_gemm<A_K, B_K, 1, 2, 2, 2>), grid, dim3(256), 0, s, k);
  // 128 x 128 on EIGHT waves (32 x 64 each): two waves per SIMD from one block cover each other's LDS / barrier waits, which a
  // lone 4-wave block per CU (a round of 256 tiles) cannot -- 3-6 % over the 4-wave form on every shape measured
  else hipLaunchKernelGGL((k_gemm<A_K, B_K, 1, 2, 4, 2>), grid, dim3(512), 0, s, k);
  DQ_LAUNCH_CHECK();
  if (p.splits > 1) {
    hipLaunchKernelGGL(k_gemm_split_reduce, dim3(p.ntiles * k.batch, bm / 2), dim3(256), 0, s, k, p.ntiles, bm);
    DQ_LAUNCH_CHECK();
  }
  return 0;
}
int64_t part_scratch(const Part& p, int bm, int batch) { return p.splits > 1 ? (int64_t)p.splits * batch * p.ntiles * bm * 128 : 0; }
}  // namespace

int64_t gemm_partial_floats(int M, int N, int K, int batch) {
  const Shape sh = choose(M, N, K, batch, 0);
  return std::max(part_scratch(sh.full, sh.bm, batch), part_scratch(sh.rest, sh.bm, batch));
}

static thread_local int g_default_precision = GEMM_FP32;
int set_gemm_precision(int precision) {
  const int old = g_default_precision;
  g_default_precision = precision == GEMM_BF16X3 ? GEMM_BF16X3 : GEMM_FP32;
  return old;
}

int launch_gemm(const Gemm& g, hipStream_t s) {
  DQ_REQUIRE(g.A && g.B && g.C, "gemm: missing operand");
  const int precision = g.precision >= 0 ? g.precision : g_default_precision;
  if (g.M <= 0 || g.N <= 0 || g.batch <= 0) return 0;
  DQ_REQUIRE(g.K > 0 && g.inner > 0, "gemm: bad reduction length / batch split");
  DQ_REQUIRE(g.lda % 4 == 0 && g.ldb % 4 == 0, "gemm: leading dimensions of A and B must be multiples of 4 floats");
  DQ_REQUIRE(((uintptr_t)g.A & 15) == 0 && ((uintptr_t)g.B & 15) == 0, "gemm: A and B must be 16-byte aligned");
  DQ_REQUIRE((g.sAo % 4 == 0) && (g.sAi % 4 == 0) && (g.sBo % 4 == 0) && (g.sBi % 4 == 0), "gemm: batch strides of A and B must be multiples of 4");
  DQ_REQUIRE(g.kbatch >= 1 && (g.kbatch == 1 || (g.sAk % 4 == 0 && g.sBk % 4 == 0)), "gemm: bad k-batch count / strides");
  const int kp = cdiv(g.K, BK) * BK;
  const Shape sh = choose(g.M, g.N, g.kbatch > 1 ? g.kbatch * kp : g.K, g.batch, g.splits);
  GemmK k;
  k.A = g.A; k.B = g.B; k.C = g.C; k.M = g.M; k.N = g.N; k.K = g.K; k.lda = g.lda; k.ldb = g.ldb; k.ldc = g.ldc;
  k.batch = g.batch; k.inner = g.inner; k.sAo = g.sAo; k.sAi = g.sAi; k.sBo = g.sBo; k.sBi = g.sBi; k.sCo = g.sCo; k.sCi = g.sCi;
  k.bias = g.bias; k.bias_m = g.bias_m; k.alpha = g.alpha; k.accumulate = g.accumulate; k.add = g.add;
  DQ_REQUIRE(!g.add || (!g.accumulate && std::max(sh.full.splits, sh.rest.splits) <= 1), "gemm: `add` needs an unsplit, non-accumulating product");
  k.splits = 1; k.k_per_split = 0; k.partial = g.partial; k.tile_base = 0; k.mt = cdiv(g.M, sh.bm);
  k.kb = g.kbatch; k.kp = kp; k.sAk = g.sAk; k.sBk = g.sBk;
  const int64_t need = std::max(part_scratch(sh.full, sh.bm, g.batch), part_scratch(sh.rest, sh.bm, g.batch));
  if (need > 0) DQ_REQUIRE(g.partial && g.partial_floats >= need, "gemm: split-K scratch missing or too small");
  DQ_REQUIRE((int64_t)k.batch * std::max(sh.full.splits, sh.rest.splits) <= 65535, "gemm: batch x splits exceeds the grid");
  for (const Part* p : {&sh.full, &sh.rest}) {
    int rc;
    if (g.a_kmajor && g.b_kmajor) rc = launch_part<true, true>(k, *p, sh.bm, s, precision);
    else if (g.a_kmajor && !g.b_kmajor) rc = launch_part<true, false>(k, *p, sh.bm, s, precision);
    else if (!g.a_kmajor && !g.b_kmajor) rc = launch_part<false, false>(k, *p, sh.bm, s, precision);
    else { set_error("gemm: the (A transposed, B k-major) layout is not built"); return 2; }
    if (rc) return rc;
  }
  return 0;
}

}  // namespace dq
