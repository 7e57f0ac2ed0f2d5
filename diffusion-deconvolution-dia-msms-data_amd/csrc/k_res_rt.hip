// The bottleneck's ResnetBlocks (reference dquartic/model/unet1d.py:271-323 at unet1d.py:1144-1148): 16 channels over a sample's RT axis
// (one "row" of RT positions per sample, k = 3 convolutions ALONG RT, identity residual, no skip input), forward and backward data path.
//
// k_res.hip gives such a block one workgroup per sample, thread = position with all 16 channels in registers: 1,536 dependent FMAs per
// conv pair with every weight a broadcast LDS read -- 20 us per launch whatever the batch (one workgroup's latency chain; 32 workgroups on
// 256 CUs at batch 32).  Here the RT POSITION is the lane column of v_mfma_f32_16x16x4_f32 (the layout of k_la_rows_bwd.hip / k_wgrad_rows
// with "row" := position):
//   lane = (g = lane / 16, j = lane % 16), a wave = a tile of 16 consecutive positions, register r of lane (g, j) = channel 4 g + r of
//   position j -- the accumulator layout of an M = 16 product AND the B operand of the next K = 16 x 3 product: K-step (r, tap) takes
//   register r, shifted by one lane inside the 16-lane row for the outer taps (DPP row_shr / row_shl: one VALU move, zeros shifted in).
//   A operand of step (r, tap) = W[co = j][ci = 4 g + r][tap]: 12 registers per conv and lane, loaded once per wave.
//   RMSNorm's channel sums = 4 registers in the lane + a sum over the four lane groups (v_permlane16_swap / v_permlane32_swap).
// A tile's edge lanes have no neighbour in the wave: tiles overlap -- forward: 14 own positions (lanes 1..14) + one halo lane each side
// (conv1 reads x straight from memory for every tap, so u1 / a1 are valid on all 16 lanes and conv2 on the inner 14); backward: 12 own
// positions (d a1 and dU1 valid on lanes 1..14, d x on 2..13).  Only own positions are stored or counted in the norm-gain / scale-shift sums.
// A workgroup = four waves = four consecutive tiles of one sample; its [d g2 | d g1 | d scale | d shift] sums go to slot
// (sample, workgroup) of gpart for launch_part_reduce (ordered, bitwise repeatable), as k_res_bwd leaves them.
// QKV (the first block only): the front of Residual(PreNorm(Attention)) (unet1d.py:552-561) rides behind the block -- its output is in
// registers in exactly the B-operand layout: xn = RMSNorm(out) g, qv = W_qv xn (256 x 16: sixteen M-tiles of four K-steps), RoPE on the
// first 16 channels of each of q's heads (adjacent pairs = registers (0, 1) and (2, 3) of a lane), and k = W_k ms1f (128 x 8) with its RoPE
// when the caller has not prepared it (sampling computes k once in front of the loop).  Four launches (k_rmsnorm_fwd, two GEMMs, k_rope) less.
// PRE (the first block's backward): the back of that front runs as the prologue -- RoPE^T on d q, d xn = W_qv^T d qv (K = 256: the K-slot order
// m = 8 q + 2 g + h puts a RoPE pair into one lane's consecutive steps), the PreNorm backward and the residual branch's gradient give the
// block's d out in registers, in the layout the backward wants (three launches less: k_rope, a GEMM, k_block_bwd; d mid1.out never exists).
// OUT (the second block): the back of the attention rides in FRONT of the block: attn_out = mid1.out + W_o o + b_o (K = 128) is formed on the
// tile's 16 lanes, conv1's outer taps come from the neighbouring lanes (12 own positions then), and in the block's BACKWARD d o = W_o^T d attn_out
// follows d x as an epilogue (two GEMM launches less).
// Any RT length (k_res.hip: <= 512; longer axes took the unfused five-launch path).  Weight gradients: unchanged (k_conv_wgrad_multi reads
// the dU1 / dU2 / a1 tensors written here).
#include "dq_common.h"
#include "dq_kernels.h"
#include <algorithm>

namespace dq {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// sum over the four lane groups (lanes j, j + 16, j + 32, j + 48): every lane receives the total
__device__ __forceinline__ float gsum4(float t) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_int(t), __float_as_int(t), false, false);
  t = __int_as_float(a[0]) + __int_as_float(a[1]);
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_int(t), __float_as_int(t), false, false);
  return __int_as_float(b[0]) + __int_as_float(b[1]);
}
// the value of lane - 1 / lane + 1 inside the 16-lane row (0 at the row's first / last lane); every lane takes part
__device__ __forceinline__ float from_left(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xF, 0xF, true)); }   // row_shr:1
__device__ __forceinline__ float from_right(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x101, 0xF, 0xF, true)); }  // row_shl:1
// sum over the 16 lanes of a row: every lane receives it
__device__ __forceinline__ float row_sum16(float v) {
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));   // quad_perm [1,0,3,2]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));   // quad_perm [2,3,0,1]
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));  // row_half_mirror
  v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));  // row_mirror
  return v;
}

constexpr int C = 16;
constexpr int FWD_OWN = 14, BWD_OWN = 12;  // own positions of a 16-lane tile
constexpr float SQC = 4.0f;                // sqrt(C)

template <bool QKV, bool OUT>
__global__ void __launch_bounds__(256) k_res_rt_fwd(ResFwd a, ResRtQkv q, ResRtOut ao, int tiles_per_wave) {
  constexpr int HALO = OUT ? 2 : 1, OWN = 16 - 2 * HALO;  // OUT: the block's input is formed in the lanes, its neighbours come from them too
  // QKV: W_qv (256 x 16) and W_k (128 x 8) staged once per workgroup -- read from memory tile by tile inside the epilogue they were sixteen
  // memory round trips in a row (21 us per launch against 6 without the epilogue); the copy overlaps the block itself
  __shared__ __attribute__((aligned(16))) float wqv_l[QKV ? 256 * C : 4];
  __shared__ __attribute__((aligned(16))) float wk_l[QKV ? 128 * 8 : 4];
  if constexpr (QKV) {
    float v[16], vk[4];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = q.wqv[u * 256 + (int)threadIdx.x];
    if (q.kk) {
#pragma unroll
      for (int u = 0; u < 4; ++u) vk[u] = q.wk[u * 256 + (int)threadIdx.x];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) wqv_l[u * 256 + (int)threadIdx.x] = v[u];
    if (q.kk) {
#pragma unroll
      for (int u = 0; u < 4; ++u) wk_l[u * 256 + (int)threadIdx.x] = vk[u];
    }
    __syncthreads();  // (in front of the tile loop: every wave reaches it, whatever its tile)
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
  const int b = blockIdx.y, n = a.n;
  float w1[4][3], w2[4][3], b1[4], g1[4], b2[4], g2[4], sc[4], sh[4];
  {
    const float* ss = a.ss + (int64_t)b * a.ss_stride;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 4 * g + r;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        w1[r][t] = a.w1[(j * C + c) * 3 + t];
        w2[r][t] = a.w2[(j * C + c) * 3 + t];
      }
      b1[r] = a.b1[c]; g1[r] = a.g1[c]; b2[r] = a.b2[c]; g2[r] = a.g2[c];
      sc[r] = ss[c] + 1.0f; sh[r] = ss[C + c];
    }
  }
  const int64_t base = (int64_t)b * C * n + (int64_t)(4 * g) * n;  // (sample, channel 4 g, position 0)
#pragma unroll 1
  for (int t = 0; t < tiles_per_wave; ++t) {
    const int p0 = (((int)blockIdx.x * 4 + wv) * tiles_per_wave + t) * OWN;
    if (p0 >= n) break;  // (wave-uniform)
    const int p = p0 - HALO + j;
    const bool inr = p >= 0 && p < n, own = j >= HALO && j < HALO + OWN && p < n;
    const bool okl = p - 1 >= 0 && p - 1 < n, okr = p + 1 >= 0 && p + 1 < n;
    float x[4][3];
    if constexpr (OUT) {
      // ---- the block's input = attn_out = res + W_o o + b_o (unet1d.py:563-567 + the Residual): K-step s takes o channel 4 s + g
      const float* ob = ao.o + (int64_t)b * 128 * n + p;
      float ov[32];
#pragma unroll
      for (int st = 0; st < 32; ++st) ov[st] = inr ? ob[(int64_t)(4 * st + g) * n] : 0.f;
      f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int st = 0; st < 32; st += 2) {
        acc0 = mfma16(ao.w[j * 128 + 4 * st + g], ov[st], acc0);
        acc1 = mfma16(ao.w[j * 128 + 4 * (st + 1) + g], ov[st + 1], acc1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = inr ? (acc0[r] + acc1[r]) + ao.b[4 * g + r] + ao.res[base + (int64_t)r * n + p] : 0.f;
        x[r][1] = v;
        if (own) ao.out[base + (int64_t)r * n + p] = v;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { x[r][0] = from_left(x[r][1]); x[r][2] = from_right(x[r][1]); }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* src = a.inA + base + (int64_t)r * n;
        x[r][0] = okl ? src[p - 1] : 0.f;
        x[r][1] = inr ? src[p] : 0.f;
        x[r][2] = okr ? src[p + 1] : 0.f;
      }
    }
    // ---- conv1 -> u1 ; RMSNorm, (scale + 1, shift), SiLU -> a1 (zero outside the axis: conv2's padding)
    f32x4 u = {b1[0], b1[1], b1[2], b1[3]};
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int k = 0; k < 3; ++k) u = mfma16(w1[r][k], x[r][k], u);
    if (a.u1 && own) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a.u1[base + (int64_t)r * n + p] = u[r];
    }
    float a1[4];
    {
      float ssq = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) ssq = fmaf(u[r], u[r], ssq);
      const float inv = rms_inv(gsum4(ssq), SQC);
#pragma unroll
      for (int r = 0; r < 4; ++r) a1[r] = inr ? silu_f(fmaf(u[r] * inv * g1[r], sc[r], sh[r])) : 0.f;
    }
    if (a.a1 && own) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a.a1[base + (int64_t)r * n + p] = a1[r];
    }
    // ---- conv2 -> u2 ; RMSNorm, SiLU ; + x
    f32x4 o = {b2[0], b2[1], b2[2], b2[3]};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float l = from_left(a1[r]), rt = from_right(a1[r]);
      o = mfma16(w2[r][0], l, o);
      o = mfma16(w2[r][1], a1[r], o);
      o = mfma16(w2[r][2], rt, o);
    }
    if (a.u2 && own) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a.u2[base + (int64_t)r * n + p] = o[r];
    }
    {
      float ssq = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) ssq = fmaf(o[r], o[r], ssq);
      const float inv = rms_inv(gsum4(ssq), SQC);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[r] = silu_f(o[r] * inv * g2[r]) + x[r][1];
      if (own) {
#pragma unroll
        for (int r = 0; r < 4; ++r) a.out[base + (int64_t)r * n + p] = o[r];
      }
    }
    if constexpr (QKV) {
      float xn[4];
      {
        float ssq = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) ssq = fmaf(o[r], o[r], ssq);
        const float inv = rms_inv(gsum4(ssq), SQC);
#pragma unroll
        for (int r = 0; r < 4; ++r) xn[r] = o[r] * inv * q.gn[4 * g + r];
      }
      if (q.xn && own) {
#pragma unroll
        for (int r = 0; r < 4; ++r) q.xn[base + (int64_t)r * n + p] = xn[r];
      }
      // RoPE (k_attn.hip: k_rope): pair pr = 2 g + (r / 2) of a head's first 16 channels, angle = position * freqs[pr]
      float cs[2] = {1.f, 1.f}, sn[2] = {0.f, 0.f};
      if (q.rope) {
#pragma unroll
        for (int h = 0; h < 2; ++h) sincosf((float)p * q.rope[2 * g + h], &sn[h], &cs[h]);
      }
      auto rotate = [&](f32x4& v) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float xa = v[2 * h], xb = v[2 * h + 1];
          v[2 * h] = xa * cs[h] - xb * sn[h];
          v[2 * h + 1] = xb * cs[h] + xa * sn[h];
        }
      };
      float* qvb = q.qv + (int64_t)b * 256 * n + (int64_t)(4 * g) * n + p;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const float4 w4 = *reinterpret_cast<const float4*>(wqv_l + (16 * t + j) * C + 4 * g);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        acc = mfma16(w4.x, xn[0], acc);
        acc = mfma16(w4.y, xn[1], acc);
        acc = mfma16(w4.z, xn[2], acc);
        acc = mfma16(w4.w, xn[3], acc);
        if (t < 8 && (t & 1) == 0) rotate(acc);  // q, channels 0..15 of head t / 2
        if (own) {
#pragma unroll
          for (int r = 0; r < 4; ++r) qvb[(int64_t)(16 * t + r) * n] = acc[r];
        }
      }
      if (q.kk) {  // (wave-uniform)
        const float* mb = q.ms1f + (int64_t)b * 8 * n;
        const float m0 = inr ? mb[(int64_t)g * n + p] : 0.f, m1 = inr ? mb[(int64_t)(4 + g) * n + p] : 0.f;
        float* kb = q.kk + (int64_t)b * 128 * n + (int64_t)(4 * g) * n + p;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const float* wr = wk_l + (16 * t + j) * 8 + g;
          f32x4 acc = {0.f, 0.f, 0.f, 0.f};
          acc = mfma16(wr[0], m0, acc);
          acc = mfma16(wr[4], m1, acc);
          if ((t & 1) == 0) rotate(acc);
          if (own) {
#pragma unroll
            for (int r = 0; r < 4; ++r) kb[(int64_t)(16 * t + r) * n] = acc[r];
          }
        }
      }
    }
  }
}

// pointwise backward of RMSNorm -> (scale + 1, shift) -> SiLU in the lane-column layout (k_res_common.h: norm_act_bwd, same arithmetic; the
// channel sums are completed over the four lane groups).  d[] in: gradient of the activation; out: dU.  Sums only where `own`.
template <bool SS>
__device__ __forceinline__ void norm_act_bwd_lc(const float (&u)[4], float (&d)[4], const float (&gn)[4], const float (&sc)[4], const float (&sh)[4],
                                                bool own, float (&dg)[4], float (&dsc)[4], float (&dsh)[4]) {
  float ssq = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) ssq = fmaf(u[r], u[r], ssq);
  const float nrm = fast_sqrt(gsum4(ssq)), inv = fast_rcp(fmaxf(nrm, RMS_EPS));
  float uh[4];
  float dot = 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    uh[r] = u[r] * inv;
    const float z = uh[r] * gn[r] * SQC;
    const float s = SS ? sc[r] : 1.0f, h = SS ? sh[r] : 0.f;
    const float w = fmaf(z, s, h);
    const float dw = d[r] * silu_grad_f(w);
    if (SS) { dsh[r] += own ? dw : 0.f; dsc[r] = fmaf(own ? dw : 0.f, z, dsc[r]); }
    const float dz = dw * s;
    dg[r] = fmaf(own ? dz : 0.f, uh[r] * SQC, dg[r]);
    d[r] = dz * gn[r] * SQC;
    dot = fmaf(d[r], uh[r], dot);
  }
  dot = gsum4(dot);
  const bool clamped = nrm < RMS_EPS;
#pragma unroll
  for (int r = 0; r < 4; ++r) d[r] = clamped ? d[r] * inv : inv * (d[r] - uh[r] * dot);
}

template <bool PRE, bool OUT>
__global__ void __launch_bounds__(256) k_res_rt_bwd(ResBwd a, ResRtPre q, ResRtOut ao, int tiles_per_wave) {
  __shared__ float red[4][5 * C];
  // PRE: W_qv staged once per workgroup (as the forward's epilogue does): the prologue's 64 weight operands per lane are LDS reads, its 64
  // d qv operands ONE batch of loads (in four chunks of weights + gradients from memory the launch took 21 us against 6.5 without the prologue)
  __shared__ __attribute__((aligned(16))) float wqv_l[PRE ? 256 * C : 4];
  if constexpr (PRE) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = q.wqv[u * 256 + (int)threadIdx.x];
#pragma unroll
    for (int u = 0; u < 16; ++u) wqv_l[u * 256 + (int)threadIdx.x] = v[u];
    __syncthreads();
  }
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, g = lane >> 4, j = lane & 15;
  const int b = blockIdx.y, n = a.n;
  // A operands of the transposed convolutions: d in[ci][p] = sum_(co, k) W[co][ci][k] dU[co][p + 1 - k]; K-step (r, k): lane group g
  // supplies co = 4 g + r, so A[i = ci = j][slot g] = W[co = 4 g + r][ci = j][k]
  float w1t[4][3], w2t[4][3], g1[4], g2[4], sc[4], sh[4];
  {
    const float* ss = a.ss + (int64_t)b * a.ss_stride;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int c = 4 * g + r;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        w1t[r][k] = a.w1[(c * C + j) * 3 + k];
        w2t[r][k] = a.w2[(c * C + j) * 3 + k];
      }
      g1[r] = a.g1[c]; g2[r] = a.g2[c];
      sc[r] = ss[c] + 1.0f; sh[r] = ss[C + c];
    }
  }
  float dg2[4] = {0.f, 0.f, 0.f, 0.f}, dg1[4] = {0.f, 0.f, 0.f, 0.f}, dsc[4] = {0.f, 0.f, 0.f, 0.f}, dsh[4] = {0.f, 0.f, 0.f, 0.f};
  float dgn[4] = {0.f, 0.f, 0.f, 0.f};  // PRE: d (PreNorm gain)
  float gn[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (PRE) {
#pragma unroll
    for (int r = 0; r < 4; ++r) gn[r] = q.gn[4 * g + r];
  }
  float none[4] = {0.f, 0.f, 0.f, 0.f};
  const int64_t base = (int64_t)b * C * n + (int64_t)(4 * g) * n;
  const bool rmw = a.dA && !a.dA_store;
#pragma unroll 1
  for (int t = 0; t < tiles_per_wave; ++t) {
    const int p0 = (((int)blockIdx.x * 4 + wv) * tiles_per_wave + t) * BWD_OWN;
    if (p0 >= n) break;  // (wave-uniform)
    const int p = p0 - 2 + j;
    const bool inr = p >= 0 && p < n, own = j >= 2 && j < 2 + BWD_OWN && p < n;
    float dout[4], d[4], u2[4], u1[4], dold[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t o = base + (int64_t)r * n + p;
      if constexpr (!PRE) dout[r] = inr ? a.dout[o] : 0.f;
      u2[r] = inr ? a.u2[o] : 1.f;
      u1[r] = inr ? a.u1[o] : 1.f;
      dold[r] = (own && rmw) ? a.dA[o] : 0.f;
    }
    if constexpr (PRE) {
      // ---- d out of this block = d attn_out (the residual around the attention) + PreNorm^T (W_qv^T RoPE^T (d q | d v)); needed on every lane
      // of the tile (the transposed convolutions read the neighbours), per position throughout
      float xo[4], ad[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t o = base + (int64_t)r * n + p;
        xo[r] = inr ? q.x[o] : 1.f;
        ad[r] = inr ? q.add[o] : 0.f;
      }
      float cs[2] = {1.f, 1.f}, sn[2] = {0.f, 0.f};
      if (q.rope) {
#pragma unroll
        for (int h = 0; h < 2; ++h) sincosf((float)p * q.rope[4 * h + g], &sn[h], &cs[h]);  // pair 4 h + g of a head's first 16 channels
      }
      const float* db = q.dqv + (int64_t)b * 256 * n + p;
      f32x4 dn0 = {0.f, 0.f, 0.f, 0.f}, dn1 = {0.f, 0.f, 0.f, 0.f};  // (two chains)
#pragma unroll 1
      for (int hf = 0; hf < 2; ++hf) {  // d q (hf 0: the half with the rotation), then d v: 32 loads in flight each
        float dv[32];  // K-steps (2 qq, 2 qq + 1): lane group g supplies m = 8 qq + 2 g, + 1
#pragma unroll
        for (int qi = 0; qi < 16; ++qi) {
          const int m0 = 8 * (16 * hf + qi) + 2 * g;
          dv[2 * qi] = inr ? db[(int64_t)m0 * n] : 0.f;
          dv[2 * qi + 1] = inr ? db[(int64_t)(m0 + 1) * n] : 0.f;
        }
#pragma unroll
        for (int qi = 0; qi < 16; ++qi) {
          const int m0 = 8 * (16 * hf + qi) + 2 * g;
          float v0 = dv[2 * qi], v1 = dv[2 * qi + 1];
          if ((qi & 3) < 2) {  // d q, channels 0..15 of head qi / 4: the transpose of the rotation by angle p * freqs[4 (qi & 3) + g]
            const float c_ = hf == 0 ? cs[qi & 1] : 1.f, s_ = hf == 0 ? sn[qi & 1] : 0.f;  // (qi & 3 is 0 or 1 here; d v is not rotated)
            const float xa = v0, xb = v1;
            v0 = xa * c_ + xb * s_;
            v1 = xb * c_ - xa * s_;
          }
          dn0 = mfma16(wqv_l[m0 * C + j], v0, dn0);
          dn1 = mfma16(wqv_l[(m0 + 1) * C + j], v1, dn1);
        }
      }
      const f32x4 dn = dn0 + dn1;
      // PreNorm backward (RMSNorm with gain, no activation: k_block_bwd's arithmetic) ; + the residual branch
      float ssq = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) ssq = fmaf(xo[r], xo[r], ssq);
      const float nrm = fast_sqrt(gsum4(ssq)), inv = fast_rcp(fmaxf(nrm, RMS_EPS));
      float uh[4], dd[4];
      float dot = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        uh[r] = xo[r] * inv;
        dgn[r] = fmaf(own ? dn[r] : 0.f, uh[r] * SQC, dgn[r]);
        dd[r] = dn[r] * gn[r] * SQC;
        dot = fmaf(dd[r], uh[r], dot);
      }
      dot = gsum4(dot);
      const bool clamped = nrm < RMS_EPS;
#pragma unroll
      for (int r = 0; r < 4; ++r) dout[r] = inr ? (ad[r] + (clamped ? dd[r] * inv : inv * (dd[r] - uh[r] * dot))) : 0.f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = dout[r];
    // ---- block2: dU2 (zero outside the axis)
    norm_act_bwd_lc<false>(u2, d, g2, none, none, own, dg2, none, none);
#pragma unroll
    for (int r = 0; r < 4; ++r) d[r] = inr ? d[r] : 0.f;
    if (own) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a.du2[base + (int64_t)r * n + p] = d[r];
    }
    // ---- d a1 = W2^T * dU2 (tap k reads position p + 1 - k)
    f32x4 da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float l = from_left(d[r]), rt = from_right(d[r]);
      da = mfma16(w2t[r][0], rt, da);
      da = mfma16(w2t[r][1], d[r], da);
      da = mfma16(w2t[r][2], l, da);
    }
    // ---- block1: dU1
    float d1[4] = {da[0], da[1], da[2], da[3]};
    norm_act_bwd_lc<true>(u1, d1, g1, sc, sh, own, dg1, dsc, dsh);
#pragma unroll
    for (int r = 0; r < 4; ++r) d1[r] = inr ? d1[r] : 0.f;
    if (own) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a.du1[base + (int64_t)r * n + p] = d1[r];
    }
    // ---- d x = W1^T * dU1 + d out (identity residual)
    f32x4 dx = {dout[0], dout[1], dout[2], dout[3]};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float l = from_left(d1[r]), rt = from_right(d1[r]);
      dx = mfma16(w1t[r][0], rt, dx);
      dx = mfma16(w1t[r][1], d1[r], dx);
      dx = mfma16(w1t[r][2], l, dx);
    }
    if (own && a.dA) {
#pragma unroll
      for (int r = 0; r < 4; ++r) a.dA[base + (int64_t)r * n + p] = dold[r] + dx[r];
    }
    if constexpr (OUT) {
      // ---- d o = W_o^T d attn_out (M = 128: eight tiles; K-step r takes channel 4 g + r of the gradient just formed)
      float dt[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) dt[r] = dold[r] + dx[r];
      float* dob = ao.d_o + (int64_t)b * 128 * n + (int64_t)(4 * g) * n + p;
#pragma unroll
      for (int tm = 0; tm < 8; ++tm) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) acc = mfma16(ao.w[(4 * g + r) * 128 + 16 * tm + j], dt[r], acc);
        if (own) {
#pragma unroll
          for (int r = 0; r < 4; ++r) dob[(int64_t)(16 * tm + r) * n] = acc[r];
        }
      }
    }
  }
  // ---- this workgroup's [d g2 | d g1 | d scale | d shift]: over the row's 16 positions (DPP), then over the four waves (fixed order)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float s0 = row_sum16(dg2[r]), s1 = row_sum16(dg1[r]), s2 = row_sum16(dsc[r]), s3 = row_sum16(dsh[r]);
    const float s4 = PRE ? row_sum16(dgn[r]) : 0.f;
    if (j == 0) {
      const int c = 4 * g + r;
      red[wv][c] = s0; red[wv][C + c] = s1; red[wv][2 * C + c] = s2; red[wv][3 * C + c] = s3;
      if (PRE) red[wv][4 * C + c] = s4;
    }
  }
  __syncthreads();
  if (threadIdx.x < 4 * C) {
    const int i = threadIdx.x;
    a.gpart[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (4 * C) + i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  } else if (PRE && threadIdx.x < 5 * C) {
    const int i = threadIdx.x;
    q.gn_part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * C + (i - 4 * C)] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
}

// tiles per wave so that a sample has at most 64 workgroups (the gpart slot count the arena reserves per sample, dq_unet.hip)
int rt_tiles_per_wave(int n, int own) { return std::max(1, cdiv(n, 4 * own * 64)); }

}  // namespace

bool res_rt_usable(int C_, int cinA, int cinB, bool has_wr, int rows_per_sample) {
  return rows_per_sample == 1 && C_ == C && cinA == C && cinB == 0 && !has_wr;
}

int launch_res_rt_fwd(const ResFwd& a, hipStream_t s, const ResRtQkv* q, const ResRtOut* ao) {
  DQ_REQUIRE(res_rt_usable(a.C, a.cinA, a.cinB, a.wr != nullptr, a.rows_per_sample) && a.n >= 1 && (a.inA || ao) && a.out && a.ss,
             "res_rt_fwd: 16 channels, identity residual, one RT row per sample");
  DQ_REQUIRE(!(q && ao), "res_rt_fwd: the attention front and back ride with different blocks");
  const int own = ao ? 12 : FWD_OWN;
  const int tpw = rt_tiles_per_wave(a.n, own);
  const dim3 grid(cdiv(a.n, 4 * own * tpw), a.rows);
  if (q) {
    DQ_REQUIRE(q->gn && q->wqv && q->qv && (!q->kk || (q->wk && q->ms1f)), "res_rt_fwd: missing attention-front operand");
    hipLaunchKernelGGL((k_res_rt_fwd<true, false>), grid, dim3(256), 0, s, a, *q, ResRtOut{}, tpw);
  } else if (ao) {
    DQ_REQUIRE(ao->o && ao->w && ao->b && ao->res && ao->out, "res_rt_fwd: missing to_out operand");
    hipLaunchKernelGGL((k_res_rt_fwd<false, true>), grid, dim3(256), 0, s, a, ResRtQkv{}, *ao, tpw);
  } else {
    hipLaunchKernelGGL((k_res_rt_fwd<false, false>), grid, dim3(256), 0, s, a, ResRtQkv{}, ResRtOut{}, tpw);
  }
  DQ_LAUNCH_CHECK();
  return 0;
}

int launch_res_rt_bwd(const ResBwd& a, hipStream_t s, const ResRtPre* q, const ResRtOut* ao) {
  DQ_REQUIRE(res_rt_usable(a.C, a.cinA, a.cinB, a.wr != nullptr, a.rows_per_sample) && a.n >= 1 && (a.dout || q) && a.u1 && a.u2 && a.du1 && a.du2 && a.ss,
             "res_rt_bwd: 16 channels, identity residual, one RT row per sample");
  DQ_REQUIRE(!(q && ao), "res_rt_bwd: the attention front and back ride with different blocks");
  const int tpw = rt_tiles_per_wave(a.n, BWD_OWN);
  const dim3 grid(cdiv(a.n, 4 * BWD_OWN * tpw), a.rows);
  DQ_REQUIRE(a.gpart && a.gblocks && a.gpart_floats >= (int64_t)grid.x * grid.y * 4 * C, "res_rt_bwd: partial-sum slot missing or too small");
  *a.gblocks = (int)grid.x;  // workgroups per sample
  if (q) {
    DQ_REQUIRE(q->dqv && q->wqv && q->x && q->gn && q->add && q->gn_part && q->gn_part_floats >= (int64_t)grid.x * grid.y * C,
               "res_rt_bwd: missing attention-front operand");
    hipLaunchKernelGGL((k_res_rt_bwd<true, false>), grid, dim3(256), 0, s, a, *q, ResRtOut{}, tpw);
  } else if (ao) {
    DQ_REQUIRE(ao->w && ao->d_o && a.dA, "res_rt_bwd: missing to_out operand");
    hipLaunchKernelGGL((k_res_rt_bwd<false, true>), grid, dim3(256), 0, s, a, ResRtPre{}, *ao, tpw);
  } else {
    hipLaunchKernelGGL((k_res_rt_bwd<false, false>), grid, dim3(256), 0, s, a, ResRtPre{}, ResRtOut{}, tpw);
  }
  DQ_LAUNCH_CHECK();
  return 0;
}

}  // namespace dq
