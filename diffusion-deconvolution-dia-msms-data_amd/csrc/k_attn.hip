// K7: the bottleneck's RT x RT softmax attention (queries/values from the folded U-Net state, keys from the MS1
// chromatogram features), with RoPE on q and k.  Forward and backward.
// Reference arithmetic: dquartic/model/unet1d.py:552-567 (Attention.forward, x-attn branch), :428-443 (Attend:
// softmax(q k^T * 32^-0.5) v).  RoPE comes from the third-party rotary_embedding_torch (unet1d.py:529, 560-561):
// default 'lang' mode, adjacent-pair rotation of the first 16 of the 32 head channels -- restated, parity unpinned.
//
// q, k, v, o live in conv layout (B, heads*32, RT), RT contiguous.  Flash-style: the RT x RT score matrix never exists in
// memory.  One WAVE owns a block of 32 queries (forward, dQ) or 32 keys (dK, dV) of one (sample, head) and sweeps the other
// side in blocks of 32; every product is a 32x32 tile on v_mfma_f32_32x32x2_f32 (exact fp32), with the orientations chosen
// so that (a) every operand tile is loaded straight from global memory in one of two register layouts -- "rows = channel,
// col = position" (16 coalesced row loads) or "rows = position, col = channel" (4 x 16-B loads per lane) -- and (b) each
// accumulator is directly the next product's operand (sum_r mfma(X.r, Y.r) = X^T Y, dq_mfma.h).  The softmax runs over the
// 16 registers of a lane + its partner lane^32.  No LDS, no barriers; several waves per SIMD hide the load latency.
//   forward   (queries on lanes):  S^T = K^T.. xty(Kt, Qt) ; online softmax over rows ; O^T = xty(Vx, P^T)
//   dQ kernel (queries on lanes):  S^T, dP^T = xty(Vt, dOt), dS^T = P^T o (dP^T - delta) ; dQ = xty(Kx, dS^T)
//   dK/dV     (keys on lanes):     S = xty(Qt, Kt), dP = xty(dOt, Vt) ; dV = xty(dOx, P) ; dK = xty(Qx, dS)
// 3.4 % of the network FLOPs at 64x400.
#include "dq_common.h"
#include "dq_dev.h"
#include "dq_kernels.h"
#include "dq_mfma.h"
#include <cstdlib>

namespace dq {


constexpr float ATT_SCALE = 0.17677669529663687f;  // 32^-0.5
constexpr float LOG2E = 1.4426950408889634f;

// ---- RoPE, in place.  sign = +1 forward, -1 backward (transpose of the rotation).
// (blockIdx.y selects one of up to two tensors: q and k of the same attention in ONE launch)
__global__ void __launch_bounds__(256) k_rope(float* __restrict__ t0, int64_t bs0, float* __restrict__ t1, int64_t bs1, const float* __restrict__ freqs,
                                              int RT, float sign, int64_t total) {
  float* __restrict__ t = blockIdx.y ? t1 : t0;
  const int64_t batch_stride = blockIdx.y ? bs1 : bs0;
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int pos = (int)(i % RT);
  const int pr = (int)((i / RT) % 8);
  const int64_t bh = i / ((int64_t)RT * 8);  // b * 4 + head
  const int64_t b = bh >> 2, h = bh & 3;
  const float ang = (float)pos * freqs[pr];
  float sn, cs;
  sincosf(ang, &sn, &cs);
  sn *= sign;
  float* pa = t + b * batch_stride + (h * 32 + 2 * pr) * RT + pos;
  const float xa = pa[0], xb = pa[RT];
  pa[0] = xa * cs - xb * sn;
  pa[RT] = xb * cs + xa * sn;
}

int launch_rope(float* qk, const float* freqs, int B, int64_t batch_stride, int RT, float sign, hipStream_t s) {
  // rotates the 4 heads x 32 channels that start at each sample's base (B, >=128, RT); batch_stride in floats
  const int64_t total = (int64_t)B * 4 * 8 * RT;
  if (total == 0) return 0;
  hipLaunchKernelGGL(k_rope, dim3(cdiv(total, 256)), dim3(256), 0, s, qk, batch_stride, (float*)nullptr, (int64_t)0, freqs, RT, sign, total);
  DQ_LAUNCH_CHECK();
  return 0;
}
// the same rotation on two tensors of the same batch and length (q inside the q | v buffer, k) in one launch
int launch_rope2(float* q, int64_t q_bs, float* k, int64_t k_bs, const float* freqs, int B, int RT, float sign, hipStream_t s) {
  const int64_t total = (int64_t)B * 4 * 8 * RT;
  if (total == 0) return 0;
  hipLaunchKernelGGL(k_rope, dim3(cdiv(total, 256), 2), dim3(256), 0, s, q, q_bs, k, k_bs, freqs, RT, sign, total);
  DQ_LAUNCH_CHECK();
  return 0;
}

// ---- operand tiles (32 channels of one head x 32 positions starting at p0; zero beyond RT)
// No load is predicated: positions beyond RT read position RT - 1 and are zeroed by a select afterwards (a predicated load is a branch
// around the load, and the loops below keep the NEXT block's tiles in flight while the current block multiplies).
// rows = channel (registers: rmap(r, half)), col = position (lane & 31): 16 row loads, each coalesced over the lanes.
// A block that lies wholly inside the sequence (p0 + 32 <= RT: wave-uniform, all but the last block of a sweep) takes the FAST form: no clamp,
// no select, no multiply when mul == 1, and the address of row r is a wave-uniform base (src + c_r RT, scalar registers) + ONE 32-bit byte offset
// per lane -- the general form spent ~10 vector instructions per row on min / compare / select / 64-bit adds, ~170 per key block of the forward
// against its 32 MFMAs (ISA count, round 4).
// (FAST = false: the forward keeps the general form for every block -- measured at batch 512: forward 579 -> 610 us with the fast form,
// backward 2,000 -> 1,843 us; batch 32: backward 166 -> 153 us)
template <bool FAST = true>
__device__ __forceinline__ f32x16 tile_ch_rows(const float* __restrict__ src, int RT, int p0, int col, int half, float mul) {
  f32x16 t;
  if (FAST && p0 + 32 <= RT) {  // (wave-uniform)
    const unsigned voff = (unsigned)(p0 + col + 4 * half * RT) * 4u;  // a (sample, head) slice is 32 RT floats: far below 2^32 bytes
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(src + (size_t)((r & 3) + 8 * (r >> 2)) * RT) + voff);
      t[r] = mul == 1.f ? v : v * mul;
    }
    return t;
  }
  const int p = p0 + col;
  const bool ok = p < RT;
  const float* s0 = src + (ok ? p : RT - 1);
  const float m = ok ? mul : 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) t[r] = s0[(int64_t)rmap(r, half) * RT] * m;
  return t;
}
// rows = position (registers: p0 + rmap(r, half)), col = channel (lane & 31): 4 x 16-B loads per lane when RT % 4 == 0
__device__ __forceinline__ f32x16 tile_pos_rows(const float* __restrict__ src, int RT, int p0, int col, int half, float mul) {
  f32x16 t;
  const float* row = src + (int64_t)col * RT;
  if ((RT & 3) == 0 && p0 + 32 <= RT) {  // (wave-uniform)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const float4 v = *reinterpret_cast<const float4*>(row + p0 + 8 * q4 + 4 * half);
      if (mul == 1.f) { t[4 * q4 + 0] = v.x; t[4 * q4 + 1] = v.y; t[4 * q4 + 2] = v.z; t[4 * q4 + 3] = v.w; }
      else { t[4 * q4 + 0] = v.x * mul; t[4 * q4 + 1] = v.y * mul; t[4 * q4 + 2] = v.z * mul; t[4 * q4 + 3] = v.w * mul; }
    }
  } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int p = p0 + rmap(r, half);
      t[r] = row[p < RT ? p : RT - 1] * (p < RT ? mul : 0.f);
    }
  }
  return t;
}
// per-position scalars by register row: v[p0 + rmap(r, half)], ``fill`` beyond RT
__device__ __forceinline__ f32x16 rows_scalar(const float* __restrict__ v, int RT, int p0, int half, float fill) {
  f32x16 t;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int p = p0 + rmap(r, half);
    const float x = v[p < RT ? p : RT - 1];
    t[r] = p < RT ? x : fill;
  }
  return t;
}

// How many waves share one block of 32 queries (forward, dQ), each taking every NW-th key block: the sweep over the other side is a
// serial chain of RT / 32 steps per wave, and at a training batch (32 samples x 4 heads x 13 blocks = 1664 waves on 1024 SIMDs) nothing
// hides its latencies.  The next block's tiles are in flight while the current one multiplies (forward / dQ / dK,dV at batch 32: 48 / 70 /
// 157 -> 44 / 62 / 130 us; the products themselves -- dependent chains of sixteen 64-cycle fp32 MFMAs -- bound the three launches at
// 18 / 27 / 36 us).  With a sampling batch the grid fills the SIMDs many times over and one wave per block (no merge) is the form.
// ---- forward: NW waves = the same 32 queries of one (sample, head); their online-softmax partials (m, l, O^T) meet in LDS and are
// merged by wave 0 in wave order (fixed order: repeatable)
// (Round-4 measurements, batch 512: with every key block reading block 0's tiles -- L1-hot -- the launch takes 512 instead of 602 us, so the
// tile loads are 15 % of it; two accumulation chains per product change nothing forward and spill backward; launch bounds of 4 / 5 waves per
// SIMD spill 41 / 80 registers: 1,012 / 1,441 us.  The sweep runs at ~3,500 cycles per key block and SIMD against 2,048 cycles of MFMA.)
template <int NW>
__global__ void __launch_bounds__(64 * NW) k_attn_fwd(const float* __restrict__ q, int64_t q_bs, const float* __restrict__ k, int64_t k_bs,
                                                      const float* __restrict__ v, int64_t v_bs, float* __restrict__ o,
                                                      float* __restrict__ lse, int RT) {
  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bh = blockIdx.y, b = bh >> 2, h = bh & 3;
  const int i0 = blockIdx.x * 32, i = i0 + col;
  const float* qb = q + b * q_bs + (int64_t)h * 32 * RT;
  const float* kb = k + b * k_bs + (int64_t)h * 32 * RT;
  const float* vb = v + b * v_bs + (int64_t)h * 32 * RT;
  const f32x16 Qt = tile_ch_rows<false>(qb, RT, i0, col, half, ATT_SCALE);  // rows d, col i
  f32x16 Oa = {0};                                                   // rows e, col i
  float m = -INFINITY, l = 0.f;
  const int jlast = (RT - 1) / 32 * 32;  // (prefetches past the end re-read the last block and are not used)
  f32x16 Kn = tile_ch_rows<false>(kb, RT, min(wv * 32, jlast), col, half, 1.f);   // rows d, col j
  f32x16 Vn = tile_pos_rows(vb, RT, min(wv * 32, jlast), col, half, 1.f);  // rows j, col e
  for (int j0 = wv * 32; j0 < RT; j0 += 32 * NW) {
    const f32x16 Kt = Kn, Vx = Vn;
    Kn = tile_ch_rows<false>(kb, RT, min(j0 + 32 * NW, jlast), col, half, 1.f);
    Vn = tile_pos_rows(vb, RT, min(j0 + 32 * NW, jlast), col, half, 1.f);
    f32x16 St = xty(Kt, Qt, f32x16{0});                           // rows j, col i
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (j0 + rmap(r, half) >= RT) St[r] = -INFINITY;
      mx = fmaxf(mx, St[r]);
    }
    mx = fmaxf(mx, swap_half(mx));
    const float mn = fmaxf(m, mx);  // finite: every block holds at least one valid key
    const float al = __builtin_amdgcn_exp2f((m - mn) * LOG2E);
    float ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      St[r] = __builtin_amdgcn_exp2f((St[r] - mn) * LOG2E);
      ls += St[r];
    }
    ls += swap_half(ls);
    l = fmaf(l, al, ls);
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) Oa[r] *= al;
    Oa = xty(Vx, St, Oa);  // rows e, col i
  }
  if constexpr (NW > 1) {
    __shared__ float part[NW - 1][18][64];
    if (wv > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) part[wv - 1][r][lane] = Oa[r];
      part[wv - 1][16][lane] = m;
      part[wv - 1][17][lane] = l;
    }
    __syncthreads();
    if (wv > 0) return;
    float mt = m;  // (wave 0 always has key block 0: finite)
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) mt = fmaxf(mt, part[w][16][lane]);
    const float s0 = __builtin_amdgcn_exp2f((m - mt) * LOG2E);
    l *= s0;
#pragma unroll
    for (int r = 0; r < 16; ++r) Oa[r] *= s0;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w) {
      const float sw = __builtin_amdgcn_exp2f((part[w][16][lane] - mt) * LOG2E);  // (a wave without key blocks: m = -inf, weight 0)
      l = fmaf(part[w][17][lane], sw, l);
#pragma unroll
      for (int r = 0; r < 16; ++r) Oa[r] = fmaf(part[w][r][lane], sw, Oa[r]);
    }
    m = mt;
  }
  if (i < RT) {
    const float rl = 1.0f / l;
    float* ob = o + (int64_t)b * 128 * RT + (int64_t)h * 32 * RT;
#pragma unroll
    for (int r = 0; r < 16; ++r) ob[(int64_t)rmap(r, half) * RT + i] = Oa[r] * rl;
    if (lse && half == 0) lse[(int64_t)bh * RT + i] = m + __logf(l);
  }
}

// waves per query / key block: four while the grid would leave SIMDs short of work, one otherwise
// waves per query block: ONE.  Four (each wave every fourth key block, partials merged in LDS) were measured at batch 32: forward 44 us
// either way, dQ 62 -> 89 us (its 253 registers leave two waves per SIMD either way, and the merge comes on top).  The four-wave forms
// stay selectable (DQ_ATTN_NW_F / DQ_ATTN_NW_Q = 4; tests/test_blocks_gpu.py is run with them).
#ifdef DQ_DEV_SWITCHES
static int attn_split(int B, int RT, bool query_side) {
  const int env_f = DQ_DEV_FLAG("DQ_ATTN_NW_F", '4') ? 4 : 0, env_q = DQ_DEV_FLAG("DQ_ATTN_NW_Q", '4') ? 4 : 0;  // (dev switches)
  (void)B; (void)RT;
  // (not chosen by grid size: a window's result must not depend on the batch it is computed in -- the four-wave merge adds the
  // partial softmaxes in another order, and tests/test_scale_parity.py holds batch 2 against batch 512 bit for bit)
  return query_side ? (env_q ? env_q : 1) : (env_f ? env_f : 1);
}
#endif

int launch_attn_fwd(const float* q, int64_t q_bs, const float* k, int64_t k_bs, const float* v, int64_t v_bs, float* o, float* lse,
                    int B, int RT, hipStream_t s) {
  if (B == 0 || RT == 0) return 0;
#ifdef DQ_DEV_SWITCHES
  if (attn_split(B, RT, false) == 4) hipLaunchKernelGGL(k_attn_fwd<4>, dim3(cdiv(RT, 32), B * 4), dim3(256), 0, s, q, q_bs, k, k_bs, v, v_bs, o, lse, RT);
  else
#endif
  hipLaunchKernelGGL(k_attn_fwd<1>, dim3(cdiv(RT, 32), B * 4), dim3(64), 0, s, q, q_bs, k, k_bs, v, v_bs, o, lse, RT);
  DQ_LAUNCH_CHECK();
  return 0;
}

// ---- backward, query side: delta_i = dO_i . O_i ; dQ_i = 32^-0.5 * sum_j P_ij (dP_ij - delta_i) K_j
template <int NW>
__global__ void __launch_bounds__(64 * NW) k_attn_bwd_q(const float* __restrict__ q, int64_t q_bs, const float* __restrict__ k,
                                                        int64_t k_bs, const float* __restrict__ v, int64_t v_bs,
                                                        const float* __restrict__ o, const float* __restrict__ d_o,
                                                        const float* __restrict__ lse, float* __restrict__ delta,
                                                        float* __restrict__ dq, int64_t dq_bs, int RT) {
  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bh = blockIdx.y, b = bh >> 2, h = bh & 3;
  const int i0 = blockIdx.x * 32, i = i0 + col;
  const float* qb = q + b * q_bs + (int64_t)h * 32 * RT;
  const float* kb = k + b * k_bs + (int64_t)h * 32 * RT;
  const float* vb = v + b * v_bs + (int64_t)h * 32 * RT;
  const float* ob = o + (int64_t)b * 128 * RT + (int64_t)h * 32 * RT;
  const float* dob = d_o + (int64_t)b * 128 * RT + (int64_t)h * 32 * RT;
  const int jlast = (RT - 1) / 32 * 32;
  f32x16 Kn = tile_ch_rows(kb, RT, min(wv * 32, jlast), col, half, 1.f);   // rows d, col j
  f32x16 Vn = tile_ch_rows(vb, RT, min(wv * 32, jlast), col, half, 1.f);   // rows e, col j
  f32x16 Xn = tile_pos_rows(kb, RT, min(wv * 32, jlast), col, half, 1.f);  // rows j, col d
  const f32x16 Qt = tile_ch_rows(qb, RT, i0, col, half, ATT_SCALE);  // rows d, col i
  const f32x16 dOt = tile_ch_rows(dob, RT, i0, col, half, 1.f);      // rows e, col i
  float dl = 0.f;
  {
    const f32x16 Ot = tile_ch_rows(ob, RT, i0, col, half, 1.f);
#pragma unroll
    for (int r = 0; r < 16; ++r) dl = fmaf(dOt[r], Ot[r], dl);
    dl += swap_half(dl);
  }
  const float ls = i < RT ? lse[(int64_t)bh * RT + i] : INFINITY;
  f32x16 dQa = {0};  // rows d, col i
  for (int j0 = wv * 32; j0 < RT; j0 += 32 * NW) {
    const f32x16 Kt = Kn, Vt = Vn, Kx = Xn;
    Kn = tile_ch_rows(kb, RT, min(j0 + 32 * NW, jlast), col, half, 1.f);
    Vn = tile_ch_rows(vb, RT, min(j0 + 32 * NW, jlast), col, half, 1.f);
    Xn = tile_pos_rows(kb, RT, min(j0 + 32 * NW, jlast), col, half, 1.f);
    f32x16 St = xty(Kt, Qt, f32x16{0});                           // rows j, col i
    const f32x16 dPt = xty(Vt, dOt, f32x16{0});                   // rows j, col i
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = (j0 + rmap(r, half) < RT) ? __builtin_amdgcn_exp2f((St[r] - ls) * LOG2E) : 0.f;
      St[r] = p * (dPt[r] - dl);  // dS^T
    }
    dQa = xty(Kx, St, dQa);
  }
  if constexpr (NW > 1) {
    __shared__ float part[NW - 1][16][64];
    if (wv > 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) part[wv - 1][r][lane] = dQa[r];
    }
    __syncthreads();
    if (wv > 0) return;
#pragma unroll
    for (int w = 0; w < NW - 1; ++w)
#pragma unroll
      for (int r = 0; r < 16; ++r) dQa[r] += part[w][r][lane];
  }
  if (i < RT) {
    float* dqb = dq + b * dq_bs + (int64_t)h * 32 * RT;
#pragma unroll
    for (int r = 0; r < 16; ++r) dqb[(int64_t)rmap(r, half) * RT + i] = dQa[r] * ATT_SCALE;
    if (half == 0) delta[(int64_t)bh * RT + i] = dl;
  }
}

// ---- backward, key side: dV_j = sum_i P_ij dO_i ; dK_j = 32^-0.5 * sum_i P_ij (dP_ij - delta_i) Q_i
// Four waves per key block, each taking every fourth query block with the next block's tiles in flight; the four partial (dK, dV) tiles
// meet in LDS and are summed in wave order by wave 0 (fixed order: repeatable).
__global__ void __launch_bounds__(256, 2) k_attn_bwd_kv(const float* __restrict__ q, int64_t q_bs, const float* __restrict__ k,
                                                     int64_t k_bs, const float* __restrict__ v, int64_t v_bs,
                                                     const float* __restrict__ d_o, const float* __restrict__ lse,
                                                     const float* __restrict__ delta, float* __restrict__ dk, int64_t dk_bs,
                                                     float* __restrict__ dv, int64_t dv_bs, int RT) {
  const int lane = threadIdx.x & 63, col = lane & 31, half = lane >> 5;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int bh = blockIdx.y, b = bh >> 2, h = bh & 3;
  const int j0 = blockIdx.x * 32, j = j0 + col;
  const float* qb = q + b * q_bs + (int64_t)h * 32 * RT;
  const float* kb = k + b * k_bs + (int64_t)h * 32 * RT;
  const float* vb = v + b * v_bs + (int64_t)h * 32 * RT;
  const float* dob = d_o + (int64_t)b * 128 * RT + (int64_t)h * 32 * RT;
  const float* lsb = lse + (int64_t)bh * RT;
  const float* dlb = delta + (int64_t)bh * RT;
  const int ilast = (RT - 1) / 32 * 32;
  f32x16 Qn = tile_ch_rows(qb, RT, min(wv * 32, ilast), col, half, ATT_SCALE), dOn = tile_ch_rows(dob, RT, min(wv * 32, ilast), col, half, 1.f);
  const f32x16 Kt = tile_ch_rows(kb, RT, j0, col, half, 1.f);  // rows d, col j
  const f32x16 Vt = tile_ch_rows(vb, RT, j0, col, half, 1.f);  // rows e, col j
  f32x16 dKa = {0}, dVa = {0};                                 // rows d / e, col j
  // The (rows i, col d) orientations of Q and dO come from the (rows d, col i) tiles through a wave-private LDS transpose.
  __shared__ float ttiles[4][32 * 33];
  __shared__ float acc_lds[3][2][16][64];  // partial (dK, dV) of waves 1..3
  float* ttile = ttiles[wv];
  for (int i0 = wv * 32; i0 < RT; i0 += 128) {
    const f32x16 Qt = Qn, dOt = dOn;  // rows d / e, col i
    // (the per-query scalars are requested here and first used behind the two score products: registers for their prefetch are not there)
    const f32x16 lsr = rows_scalar(lsb, RT, i0, half, INFINITY);  // exp(s - inf) = 0 masks the tail
    const f32x16 dlr = rows_scalar(dlb, RT, i0, half, 0.f);
    Qn = tile_ch_rows(qb, RT, min(i0 + 128, ilast), col, half, ATT_SCALE);
    dOn = tile_ch_rows(dob, RT, min(i0 + 128, ilast), col, half, 1.f);
    // The P side first (S, P, dV), then the dS side (dP, dS, dK): each transposed tile is formed right in front of its product, so that
    // Q / Q^T and dO / dO^T are not all live at once (the kernel held 6 registers more than its 256 and spilled them).
    f32x16 S = xty(Qt, Kt, f32x16{0});              // rows i, col j
#pragma unroll
    for (int r = 0; r < 16; ++r) S[r] = __builtin_amdgcn_exp2f((S[r] - lsr[r]) * LOG2E);  // P
    {
      const f32x16 dOx = transpose_tile(dOt, ttile, col, half);  // rows i, col e
      dVa = xty(dOx, S, dVa);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 dP = xty(dOt, Vt, f32x16{0});            // rows i, col j
#pragma unroll
    for (int r = 0; r < 16; ++r) dP[r] = S[r] * (dP[r] - dlr[r]);  // dS
    {
      const f32x16 Qx = transpose_tile(Qt, ttile, col, half);    // rows i, col d
      dKa = xty(Qx, dP, dKa);
    }
  }
  if (wv > 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc_lds[wv - 1][0][r][lane] = dKa[r]; acc_lds[wv - 1][1][r][lane] = dVa[r]; }
  }
  __syncthreads();
  if (wv == 0 && j < RT) {
    float* dkb = dk + b * dk_bs + (int64_t)h * 32 * RT;
    float* dvb = dv + b * dv_bs + (int64_t)h * 32 * RT;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float sk = ((dKa[r] + acc_lds[0][0][r][lane]) + acc_lds[1][0][r][lane]) + acc_lds[2][0][r][lane];
      const float sv = ((dVa[r] + acc_lds[0][1][r][lane]) + acc_lds[1][1][r][lane]) + acc_lds[2][1][r][lane];
      dkb[(int64_t)rmap(r, half) * RT + j] = sk;  // Qx already carries the 32^-0.5
      dvb[(int64_t)rmap(r, half) * RT + j] = sv;
    }
  }
}

int launch_attn_bwd(const float* q, int64_t q_bs, const float* k, int64_t k_bs, const float* v, int64_t v_bs, const float* o,
                    const float* d_o, const float* lse, float* delta, float* dq, int64_t dq_bs, float* dk, int64_t dk_bs, float* dv,
                    int64_t dv_bs, int B, int RT, hipStream_t s) {
  if (B == 0 || RT == 0) return 0;
  dim3 grid(cdiv(RT, 32), B * 4);
#ifdef DQ_DEV_SWITCHES
  if (attn_split(B, RT, true) == 4) hipLaunchKernelGGL(k_attn_bwd_q<4>, grid, dim3(256), 0, s, q, q_bs, k, k_bs, v, v_bs, o, d_o, lse, delta, dq, dq_bs, RT);
  else
#endif
  hipLaunchKernelGGL(k_attn_bwd_q<1>, grid, dim3(64), 0, s, q, q_bs, k, k_bs, v, v_bs, o, d_o, lse, delta, dq, dq_bs, RT);
  DQ_LAUNCH_CHECK();
  hipLaunchKernelGGL(k_attn_bwd_kv, grid, dim3(256), 0, s, q, q_bs, k, k_bs, v, v_bs, d_o, lse, delta, dk, dk_bs, dv, dv_bs, RT);
  DQ_LAUNCH_CHECK();
  return 0;
}

}  // namespace dq
