// One launch for the convolutional part of a U-Net level (reference dquartic/model/unet1d.py:1134-1142, 1150-1158, 1160-1163):
//   [ the resample conv that produces the level's input: Downsample k4 s2 | Upsample nearest x2 + k3 | k3 (last levels) ]
//   -> ResnetBlock -> ResnetBlock
// with every convolution on the 4x4x1 matrix pipe in the lane = position layout of k_res_mm.hip.  The level's input never goes
// to memory in inference (the resampled tensor `rs` and, on the way down, the re-read of block 1's output disappear), and the four
// launches of a level become two (this + the LinearAttention).
//
// Input stage, lane q = output position of the level (row length n):
//   DOWN: out[q] = sum_k w[k] in[2 q - 1 + k]   -- the lane loads in[2 q], in[2 q + 1] (8 bytes); in[2 q - 1] is lane q - 1's second
//         value and in[2 q + 2] lane q + 1's first (one DPP wave shift each)
//   UP:   out[q] = sum_k w[k] up[q - 1 + k], up[j] = in[j / 2]  -- the lane loads in[q / 2] (= up[q]); its neighbours' values are up[q -+ 1]
//   S1:   plain k3
//   INIT: the network's first layer (unet1d.py:1107-1118): channel 0 = mixture * (scale + 1) + shift (ConditionalScaleShift on the normalised
//         mixture), channel 1 = x_t, then init_conv k7 p3 -- three DPP shifts either way per channel; its output (h0) is always written
//         (the final ResnetBlock concatenates it)
// Head epilogue (the launch that ends with final_res_block): eps = final_conv (1x1, C -> 1) of the block output and, while sampling, the
// DDIM update x_{t-1} = f(x_t, eps) of model.py:265-289 -- the block output, eps and the separate update launch never touch memory.
// Every global read of a tile (stage input, the skip channels of cat(x, skip), unet1d.py:1151, 1154) is requested before the first use.
#include "dq_common.h"
#include "dq_kernels.h"
#include "dq_probe.h"
#include <algorithm>

namespace dq {

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float lane_m1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x138, 0xF, 0xF, true)); }
__device__ __forceinline__ float lane_p1(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x130, 0xF, 0xF, true)); }
__host__ __device__ constexpr int pad4(int x) { return (x + 3) / 4 * 4; }

// job offsets of the weight image: [stage][block 0: conv1 | conv2 | res][block 1: ...]; a job = one MFMA weight operand
struct Jobs {
  int pre, c1[2], c2[2], rs[2], total;
};
__host__ __device__ inline Jobs level_jobs(int C, int pre_mode, int cp, int nblocks, const int* cin, const bool* wr) {
  Jobs j;
  const int G = C / 4;
  int o = 0;
  j.pre = o;
  if (pre_mode != LEVEL_PRE_NONE) o += pad4(G * cp * (pre_mode == LEVEL_PRE_DOWN ? 4 : (pre_mode == LEVEL_PRE_INIT ? 7 : 3)));
  for (int b = 0; b < 2; ++b) {
    j.c1[b] = o;
    if (b < nblocks) o += pad4(G * cin[b] * 3);
    j.c2[b] = o;
    if (b < nblocks) o += pad4(G * C * 3);
    j.rs[b] = o;
    if (b < nblocks && wr[b]) o += pad4(G * cin[b]);
  }
  j.total = o;
  return j;
}

// What the kernel receives: parameter tensors as 32-bit offsets into the flat parameter buffer P (ONE `const float* __restrict__`
// kernel argument: wave-uniform reads of biases / gains / weights become scalar loads, and a tensor costs one scalar register instead of
// two -- with 26 pointers the kernel spilled scalar registers inside the tile loop), activations as pointers.
struct LevelBlkK {
  int w1, b1, g1, w2, b2, g2, wr, br;  // wr < 0: identity residual
  int ss_off, cinB;
  const float* inB; float* u1; float* a1; float* u2; float* out;
};
struct LevelFwdK {
  const float* in; float* pre_out;
  int pw, pb, nblocks, rows_per_sample, n, ss_stride;
  LevelBlkK blk[2];
  // INIT stage: the mixture (rows, n), its normalisation cond * cm + ca, the offset of init_cond_proj's [scale, shift] in the ss vector
  const float* cond; float cm, ca; int ss_init;
  float* cat0_out;  // (TH instantiation of the INIT stage)
  const float* qs_noise; const float* qs_ab; const int64_t* qs_t; int qs_norm;  // (TH, nullable: x_t = sqrt(ab) x0 + sqrt(1 - ab) noise formed here)
  // head epilogue (ep_w >= 0): final_conv weight / bias offsets in P; eps_out nullable; DDIM update when x_t is set
  int ep_w, ep_b, pred_x0;
  float* eps_out; const float* x_t; float* x_out; const float* coef; const int* step_ptr;
  // training head: target, per-wave squared-error sums, d loss / d eps, d loss / d (block output)
  const float* loss_z; float* loss_part; float* grad_out; float* dout; float loss_gscale;
};

}  // namespace


// Source of element idx of the operand image ([job / 4][lane & 3][job % 4]) in the flat parameter buffer, or -1 (padding).  No control flow:
// called for several elements at a time with their loads in flight together.
struct LevelImgSrc {
  int pre, G, C, cp, kp, nblocks, pw;
  int w1[2], w2[2], wr[2], cin[2];  // wr < 0: identity residual
};
__device__ __forceinline__ int level_img_src(const LevelImgSrc& m, const Jobs& J, int idx) {
  const int j = (idx >> 4) * 4 + (idx & 3), li = (idx >> 2) & 3;
  const int G = m.G, C = m.C;
  int off = -1;
  if (m.pre != LEVEL_PRE_NONE) {  // stage: job = (c * G + g) * KP + k
    const int jj = j - J.pre;
    const int k = jj % m.kp, g = (jj / m.kp) % G, c = jj / (m.kp * G);
    off = (jj >= 0 && jj < G * m.cp * m.kp) ? m.pw + ((4 * g + li) * m.cp + c) * m.kp + k : off;
  }
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int cin = m.cin[b];
    const bool on = b < m.nblocks;
    {
      const int jj = j - J.c1[b], k = jj % 3, g = (jj / 3) % G, c = jj / (3 * G);          // conv1: job = (c * G + g) * 3 + k
      off = (on && jj >= 0 && jj < G * cin * 3) ? m.w1[b] + ((4 * g + li) * cin + c) * 3 + k : off;
    }
    {
      const int jj = j - J.c2[b], k = jj % 3, g = (jj / 3) % G, c = jj / (3 * G);          // conv2
      off = (on && jj >= 0 && jj < G * C * 3) ? m.w2[b] + ((4 * g + li) * C + c) * 3 + k : off;
    }
    {
      const int jj = j - J.rs[b], g = jj % G, c = jj / G;                                  // res_conv: job = c * G + g
      off = (on && m.wr[b] >= 0 && jj >= 0 && jj < G * cin) ? m.wr[b] + (4 * g + li) * cin + c : off;
    }
  }
  return off;
}

// The images of several launches, built ONCE per parameter state (launch_level_images): block (x, y) = a slice of image y.
struct LevelImgItem { LevelImgSrc m; float* dst; };
struct LevelImgMulti { LevelImgItem it[LEVEL_IMG_MAX]; };
__global__ void __launch_bounds__(256) k_level_images(LevelImgMulti mm, const float* __restrict__ P) {
  const LevelImgItem& it = mm.it[blockIdx.y];
  const bool wr[2] = {it.m.wr[0] >= 0, it.m.wr[1] >= 0};
  const Jobs J = level_jobs(it.m.C, it.m.pre, it.m.cp, it.m.nblocks, it.m.cin, wr);
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= J.total * 4) return;
  const int off = level_img_src(it.m, J, idx);
  it.dst[idx] = off < 0 ? 0.f : P[off];
}

// CP: input channels of the stage (compile time: every global read of a tile is issued up front from statically indexed registers --
// with the channel quads walked in a run-time load -> use loop a wave exposed one memory latency per quad and the launch ran at a
// third of the rate its loads in flight allow)
// N64: rows of exactly 64 positions = one row per wave: the wave shifts' zero fill at lanes 0 / 63 IS the conv's zero padding, no masks
// TH: the train step's code is compiled in -- the INIT stage that stores cat0 and may form x_t (LevelFwd::cat0_out, qs_*), the training head behind
// the final block (LevelFwd::loss_z).  Instantiations of their own, so that the inference kernels keep their registers (as a run-time branch the
// head took every C = 4 kernel from 77 to 87: five instead of six workgroups per CU)
template <int C, int PRE, int CP, bool N64, bool TH = false>
__global__ void __launch_bounds__(256) k_level_fwd(LevelFwdK a, const float* __restrict__ P, const float* __restrict__ ssb, int tiles_ps, int total_tiles,
                                                   int ln_rt, const float* __restrict__ img) {  // ssb: the per-sample scale / shift vectors; ln_rt = log2(n)
  // N64: the row length is a compile-time 64 -- a channel's plane offset (c * 256 bytes) then folds into the instructions' immediate offsets
  // instead of costing a 64-bit address add per access (69 of them in <4, 2, 8>) and scalar registers for the per-channel bases
  const int ln = N64 ? 6 : ln_rt;
  constexpr int G = C / 4;
  constexpr int KP = PRE == LEVEL_PRE_DOWN ? 4 : (PRE == LEVEL_PRE_INIT ? 7 : 3);
  extern __shared__ __attribute__((aligned(16))) float wl[];  // [job / 4][lane & 3][job % 4]
  const int cin_[2] = {C + a.blk[0].cinB, C + a.blk[1].cinB};
  const bool wr_[2] = {a.blk[0].wr >= 0, a.blk[1].wr >= 0};
  const Jobs J = level_jobs(C, PRE, CP, a.nblocks, cin_, wr_);
  DQ_PSTAMP(C * 1000 + PRE * 100 + CP, 0);
  // Staging of the operand image.  With a prepared image (launch_level_images, once per parameter state) it is a linear copy, every
  // 16-byte load of a thread in flight at once.  Without one (the stand-alone entry point) the workgroup gathers it from the parameter
  // tensors: element offsets formed without control flow, four loads in flight -- ~12 us per workgroup at 16 channels, which at training
  // batch sizes (one or two tiles per wave at the deep levels) was a third of the launch.
  constexpr int MAXJ = pad4(G * CP * KP) + 2 * (pad4(G * 2 * C * 3) + pad4(G * C * 3) + pad4(G * 2 * C));
  constexpr int NLD = (MAXJ + 255) / 256;  // rounds of 256 x 16 bytes that cover the largest image of this instantiation
  if (img) {
    // NO guards: the image slot (LEVEL_IMG_FLOATS) and the LDS region (level_img_rounds x 4 KB) hold whole rounds.  With `if (i < total4) wl[i] = v[u]`
    // the compiler sank every load under its store's branch -- `global_load_dwordx4; s_waitcnt vmcnt(0); ds_write_b128` four to six times in a
    // row, one memory round trip per 4 KB of image at the head of every workgroup (ISA of every instantiation, round 4), the opposite of what this
    // block was written for.
    float4 v[NLD];
#pragma unroll
    for (int u = 0; u < NLD; ++u) v[u] = reinterpret_cast<const float4*>(img)[u * 256 + (int)threadIdx.x];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NLD; ++u) reinterpret_cast<float4*>(wl)[u * 256 + (int)threadIdx.x] = v[u];
  } else {
    LevelImgSrc m;
    m.pre = PRE; m.G = G; m.C = C; m.cp = CP; m.kp = KP; m.nblocks = a.nblocks; m.pw = a.pw;
#pragma unroll
    for (int b = 0; b < 2; ++b) { m.w1[b] = a.blk[b].w1; m.w2[b] = a.blk[b].w2; m.wr[b] = a.blk[b].wr; m.cin[b] = cin_[b]; }
    const int total = J.total * 4;
    for (int base = threadIdx.x; base < total; base += 256 * 4) {
      int off[4];
      float v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) off[u] = base + u * 256 < total ? level_img_src(m, J, base + u * 256) : -1;
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = P[off[u] < 0 ? 0 : off[u]];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (base + u * 256 < total) wl[base + u * 256] = off[u] < 0 ? 0.f : v[u];
    }
  }
  // Per-channel parameters and THIS SAMPLE's scale / shift vectors (a workgroup works on one sample, blockIdx.y), also in LDS:
  // [stage bias C][per block: b1 | g1 | b2 | g2 | br | scale + 1 | shift] -- read back as 16-byte broadcasts.  Inside the tile loop a
  // wave then waits on nothing but its own tile's loads and LDS reads (as scalar loads from memory, re-issued per tile under
  // scalar-register pressure, they cost a full wait each: the launch time did not move with the instruction count).
  float* prm = wl + (img ? NLD * 1024 : J.total * 4);  // (the unguarded image copy fills whole rounds)
  DQ_PSTAMP(C * 1000 + PRE * 100 + CP, 1);
  {
    // ONE value per thread (C * 15 + 8 <= 248), no control flow: the source offset -- into P, or into this sample's scale / shift vector -- is a
    // chain of selects, both loads are unconditional, the store is unguarded (the table holds 256 floats).  As `for (i ...) { if (what == 0) ...
    // else if (k == 0) v = P[...] ...; prm[i] = v; }` every arm kept its own load and its own s_waitcnt vmcnt(0), and a wave -- whose 64 threads
    // span ~5 of the 15 vectors -- walked them one after the other: 5+ memory round trips in series at the head of every workgroup (ISA, round 4).
    const int b = blockIdx.y;
    const int i = threadIdx.x;
    const int what = i / C, c = i - what * C, t = i - C * 15;  // t >= 0: [final_conv weight (4) | bias | init scale + 1 | init shift | 0]
    const int bi = what >= 8 ? 1 : 0, k = (what >= 1 && what <= 14) ? (what - 1) % 7 : 7;
    const bool on = what >= 1 && what <= 14 && bi < a.nblocks;
    const int b1 = bi ? a.blk[1].b1 : a.blk[0].b1, g1 = bi ? a.blk[1].g1 : a.blk[0].g1, b2 = bi ? a.blk[1].b2 : a.blk[0].b2;
    const int g2 = bi ? a.blk[1].g2 : a.blk[0].g2, br = bi ? a.blk[1].br : a.blk[0].br, wrr = bi ? a.blk[1].wr : a.blk[0].wr;
    const int sso = bi ? a.blk[1].ss_off : a.blk[0].ss_off;
    int offP = -1, offS = 0;   // (offsets into the scale / shift vector are relative to block 0's and may be NEGATIVE: hasS says whether one is meant)
    bool plus1 = false, hasS = false;
    if (what == 0) offP = PRE != LEVEL_PRE_NONE ? a.pb + c : -1;
    offP = on ? (k == 0 ? b1 + c : k == 1 ? g1 + c : k == 2 ? b2 + c : k == 3 ? g2 + c : (k == 4 && wrr >= 0) ? br + c : -1) : offP;
    hasS = on && (k == 5 || k == 6);
    offS = hasS ? (k == 5 ? sso + c : sso + C + c) : 0;
    plus1 = on && k == 5;
    if (t >= 0) {
      offP = (C == 4 && a.ep_w >= 0 && t < 4) ? a.ep_w + t : ((C == 4 && a.ep_w >= 0 && t == 4) ? a.ep_b : -1);
      hasS = PRE == LEVEL_PRE_INIT && (t == 5 || t == 6);
      offS = hasS ? a.ss_init + (t - 5) : 0;
      plus1 = PRE == LEVEL_PRE_INIT && t == 5;
    }
    const float pv = P[offP >= 0 ? offP : 0];
    const float sv = ssb[(int64_t)b * a.ss_stride + offS];
    prm[i] = hasS ? (plus1 ? sv + 1.0f : sv) : (offP >= 0 ? pv : 0.f);
  }
  __syncthreads();
  DQ_PSTAMP(C * 1000 + PRE * 100 + CP, 2);
  auto prm4 = [&](int what, int g) -> float4 { return *reinterpret_cast<const float4*>(prm + what * C + 4 * g); };
  const int lane = threadIdx.x & 63, li = lane & 3;
  const int wid = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nwaves = gridDim.x * 4;
  const int b = blockIdx.y;
  const int n = N64 ? 64 : a.n;
  const int per_sample = a.rows_per_sample * n;
  const float sqC = sqrtf((float)C);
  const float* wlane = wl + li * 4;
  // Weight operands are consumed as a STREAM of jobs in image order (one 16-byte LDS read per four jobs) through a ring of RD reads in
  // flight: the read of jobs 4 q + 4 RD .. is issued when job 4 q is reached.  (Left to the compiler, one read was in flight -- issued
  // four MFMAs = ~34 cycles before its use against an LDS latency of 64+: with one wave per SIMD, as at the deep levels of a training
  // batch, every group of four MFMAs waited and a 16-channel block ran at ~20 cycles per MFMA instead of ~8.5.)
  constexpr int RD = 3;
  struct WRing { float4 r[RD]; float4 cur; };
  auto ldw = [&](int base, int q) -> float4 { return *reinterpret_cast<const float4*>(wlane + (base >> 2) * 16 + q * 16); };
  auto ring_start = [&](WRing& R, int base) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < RD; ++d) R.r[d] = ldw(base, d);  // (reads past a stream's last job land in the next stream / the parameter image)
  };
  // operand of job j of the stream at `base` (j: a compile-time constant once the loops are unrolled; jobs are taken in order)
  auto wjob = [&](WRing& R, int base, int j) __attribute__((always_inline)) -> float {
    if ((j & 3) == 0) {
      R.cur = R.r[(j >> 2) % RD];
      R.r[(j >> 2) % RD] = ldw(base, (j >> 2) + RD);
      __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks the read to one group ahead again, to save registers)
    }
    return (j & 3) == 0 ? R.cur.x : (j & 3) == 1 ? R.cur.y : (j & 3) == 2 ? R.cur.z : R.cur.w;
  };

  // the grid is ONE resident round (launcher); wave w takes the 64-position tiles w, w + nwaves, ... of the (sample, tile) list
  float dsa = 1.f, dsb = 0.f, dsap = -1.f, dsbp = 0.f;  // this step's row of the DDIM coefficient table (read once, outside the tile loop)
  if (a.x_t) {
    const float* cf = a.coef + (a.step_ptr ? 4 * a.step_ptr[0] : 0);
    dsa = cf[0]; dsb = cf[1]; dsap = cf[2]; dsbp = cf[3];
  }
  // Index arithmetic per tile is kept off the critical path: n is a power of two (shifts), offsets are 32-bit element counts against
  // wave-uniform channel base pointers (the launcher checks every tensor stays below 2^31 elements), and NO load is predicated -- the
  // lanes beyond a sample's last position read its last position instead (their values reach no live lane: a sample ends at a row end)
  // -- so the only control flow is one branch around each group of stores.
  float qsa = 1.f, qsb = 0.f;  // train step with q_sample in the INIT stage: this sample's sqrt(ab), sqrt(1 - ab)
  if constexpr (TH && PRE == LEVEL_PRE_INIT) {
    if (a.qs_noise) {
      const float ab = a.qs_ab[a.qs_t[b]];
      qsa = sqrtf(ab); qsb = sqrtf(1.0f - ab);
    }
  }
  float lacc = 0.f;  // training head: this lane's sum of squared errors
#pragma unroll 1
  for (int tile = wid; tile < tiles_ps; tile += nwaves) {  // the workgroups of a sample share its tiles; the grid is one resident round
    const int it = tile * 64 + lane;
    const bool live = it < per_sample;
    const int itc = live ? it : per_sample - 1;
    const int rr = itc >> ln, p = itc & (n - 1);
    const unsigned row = (unsigned)(b * a.rows_per_sample + rr);
    const bool hasL = N64 || p > 0, hasR = N64 || p + 1 < n;  // (N64: constant true -- the selects below fold away)
    const unsigned obase = (((row * C) << ln) + p) * 4u;  // BYTE offset of (row, channel 0, p) in a (rows, C, n) tensor
    // tensor base + channel (wave-uniform, scalar registers) + the lane's 32-bit byte offset: one address register per lane
    auto ld = [&](const float* base, int c, unsigned boff) -> float {
      return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base + ((size_t)c << ln)) + boff);
    };
    auto st = [&](float* base, int c, unsigned boff, float v) {
      *reinterpret_cast<float*>(reinterpret_cast<char*>(base + ((size_t)c << ln)) + boff) = v;
    };
    float x[C];
    // skip channels of the two blocks (cinB <= C): block 0's are requested with the stage's input, block 1's before block 0 computes
    float xb[2][C];
    auto load_skip = [&](int bi) {
      const LevelBlkK& r = a.blk[bi];
      const unsigned boff = (((row * r.cinB) << ln) + p) * 4u;
#pragma unroll
      for (int c = 0; c < C; ++c) xb[bi][c] = 0.f;
      if (bi < a.nblocks && r.cinB > 0) {  // ONE wave-uniform branch around the group; inside it no load is predicated (channels beyond cinB re-read channel 0)
#pragma unroll
        for (int c = 0; c < C; ++c) xb[bi][c] = ld(r.inB, c < r.cinB ? c : 0, boff);
        // (NO masking here: the consumers below take whole quads of channels under `4 * cq < cinB`, so a value beyond cinB is never used -- and
        // `c < cinB ? v : 0` was a USE of every loaded value right behind its load: eleven s_waitcnt in a row in the middle of the tile, one
        // full memory latency per tile with block 1's skip channels supposedly "travelling while block 0 computes" (ISA, round 4))
      }
    };
    // ---------------------------------------------------------------- input stage
    if constexpr (PRE == LEVEL_PRE_NONE) {
#pragma unroll
      for (int c = 0; c < C; ++c) x[c] = ld(a.in, c, obase);
      load_skip(0);
    } else if constexpr (PRE == LEVEL_PRE_INIT) {
      // cat(conditioned mixture, x_t) -> init_conv k7 p3 (zero padding of the CONCATENATED tensor: the shifts bring zeros in)
      const unsigned ioff = ((row << ln) + p) * 4u;
      float xt = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.in) + ioff);
      if constexpr (TH) {
        if (a.qs_noise) {  // (wave-uniform) q_sample in place: a.in is x0
          const float nz = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.qs_noise) + ioff);
          if (a.qs_norm) xt = xt * 2.f - 1.f;
          xt = qsa * xt + qsb * nz;
        }
      }
      const float cd = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.cond) + ioff);
      load_skip(0);
      const float4 e4 = *reinterpret_cast<const float4*>(prm + C * 15 + 4);  // bias (unused here), scale + 1, shift
      float v[2] = {fmaf(cd, a.cm, a.ca) * e4.y + e4.z, xt};
      if constexpr (TH) {  // train step: the concatenated input is kept for the backward
        if (live) {
          const unsigned coff = (((row * 2) << ln) + p) * 4u;
          st(a.cat0_out, 0, coff, v[0]); st(a.cat0_out, 1, coff, v[1]);
        }
      }
      f32x4 acc[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      WRing R;
      ring_start(R, J.pre);
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        // taps k = 0..6 read position p + k - 3
        float m1 = lane_m1(v[c]); m1 = (N64 || p >= 1) ? m1 : 0.f;
        float m2 = lane_m1(m1);   m2 = (N64 || p >= 2) ? m2 : 0.f;
        float m3 = lane_m1(m2);   m3 = (N64 || p >= 3) ? m3 : 0.f;
        float p1 = lane_p1(v[c]); p1 = (N64 || p + 1 < n) ? p1 : 0.f;
        float p2 = lane_p1(p1);   p2 = (N64 || p + 2 < n) ? p2 : 0.f;
        float p3 = lane_p1(p2);   p3 = (N64 || p + 3 < n) ? p3 : 0.f;
        const float tap[7] = {m3, m2, m1, v[c], p1, p2, p3};
#pragma unroll
        for (int k = 0; k < 7; ++k) acc[k] = mfma4(wjob(R, J.pre, c * 7 + k), tap[k], acc[k]);
      }
      {
        const float4 pb4 = prm4(0, 0);
        const float pbv[4] = {pb4.x, pb4.y, pb4.z, pb4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) x[i] = ((acc[0][i] + acc[1][i]) + (acc[2][i] + acc[3][i])) + ((acc[4][i] + acc[5][i]) + (acc[6][i] + pbv[i]));
      }
      if (live) {
#pragma unroll
        for (int c = 0; c < C; ++c) st(a.pre_out, c, obase, x[c]);
      }
    } else {
      f32x4 acc[G][KP];
#pragma unroll
      for (int g = 0; g < G; ++g)
#pragma unroll
        for (int k = 0; k < KP; ++k) acc[g][k] = f32x4{0.f, 0.f, 0.f, 0.f};
      const int n_in = PRE == LEVEL_PRE_DOWN ? 2 * n : (PRE == LEVEL_PRE_UP ? n / 2 : n);
      const int p_in = PRE == LEVEL_PRE_DOWN ? 2 * p : (PRE == LEVEL_PRE_UP ? p >> 1 : p);
      const unsigned soff = (row * CP * n_in + p_in) * 4u;
      float v0[CP], v1[PRE == LEVEL_PRE_DOWN ? CP : 1];
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        const float* s = a.in + (size_t)c * n_in;  // wave-uniform
        if constexpr (PRE == LEVEL_PRE_DOWN) {
          const float2 v = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(s) + soff);
          v0[c] = v.x; v1[c] = v.y;
        } else {
          v0[c] = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(s) + soff);
        }
      }
      load_skip(0);  // block 0's skip channels: requested together with the stage's input
      WRing R;
      ring_start(R, J.pre);
#pragma unroll
      for (int c = 0; c < CP; ++c) {
        if constexpr (PRE == LEVEL_PRE_DOWN) {
          const float tm = lane_m1(v1[c]), tp = lane_p1(v0[c]);
          const float xm = hasL ? tm : 0.f, xp = hasR ? tp : 0.f;
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const int js = (c * G + g) * 4;
            acc[g][0] = mfma4(wjob(R, J.pre, js + 0), xm, acc[g][0]);
            acc[g][1] = mfma4(wjob(R, J.pre, js + 1), v0[c], acc[g][1]);
            acc[g][2] = mfma4(wjob(R, J.pre, js + 2), v1[c], acc[g][2]);
            acc[g][3] = mfma4(wjob(R, J.pre, js + 3), xp, acc[g][3]);
          }
        } else {
          const float tm = lane_m1(v0[c]), tp = lane_p1(v0[c]);
          const float xm = hasL ? tm : 0.f, xp = hasR ? tp : 0.f;
#pragma unroll
          for (int g = 0; g < G; ++g) {
            const int js = (c * G + g) * 3;
            acc[g][0] = mfma4(wjob(R, J.pre, js + 0), xm, acc[g][0]);
            acc[g][1] = mfma4(wjob(R, J.pre, js + 1), v0[c], acc[g][1]);
            acc[g][2] = mfma4(wjob(R, J.pre, js + 2), xp, acc[g][2]);
          }
        }
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float4 pb4 = prm4(0, g);
        const float pbv[4] = {pb4.x, pb4.y, pb4.z, pb4.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float s = (acc[g][0][i] + acc[g][1][i]) + acc[g][2][i];
          if constexpr (KP == 4) s += acc[g][3][i];
          x[4 * g + i] = s + pbv[i];
        }
      }
      if (a.pre_out && live) {
#pragma unroll
        for (int c = 0; c < C; ++c) st(a.pre_out, c, obase, x[c]);
      }
    }
    // ---------------------------------------------------------------- the level's ResnetBlocks
    DQ_PSTAMP(C * 1000 + PRE * 100 + CP, 3);
#pragma unroll
    for (int bi = 0; bi < 2; ++bi) {
      if (bi >= a.nblocks) break;
      if (bi == 1) DQ_PSTAMP(C * 1000 + PRE * 100 + CP, 4);
      const LevelBlkK& r = a.blk[bi];
      const bool wr = r.wr >= 0;
      const int pq = 1 + 7 * bi;  // this block's rows of the parameter image: b1, g1, b2, g2, br, scale + 1, shift
      constexpr int NAR = G == 1 ? 4 : 2;  // independent accumulation chains of the residual conv per output quad
      // ONE accumulation chain per output quad, started from the bias, at 4 / 8 channels: those levels always run with many waves per SIMD (rows
      // of 8 .. 64 positions: thousands of tiles even at batch 1), which cover a dependent 4x4x1 chain (14.5 against 8.5 cycles per MFMA), and
      // the three-chain form's `(a0 + a1) + (a2 + bias)` is 3 C VALU adds per conv in a kernel bound by VALU issue.  12 / 16 channels (deep
      // levels, one wave per SIMD at a training batch) keep one chain per tap.
      constexpr bool ONE = C <= 8;
      constexpr int TK = ONE ? 1 : 3;
      auto tapk = [](int k) constexpr { return ONE ? 0 : k; };
      f32x4 acc[G][TK], ar[G][NAR];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if constexpr (ONE) {
          const float4 t = prm4(pq + 0, g);
          acc[g][0] = f32x4{t.x, t.y, t.z, t.w};
        } else {
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[g][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int k = 0; k < NAR; ++k) ar[g][k] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      // conv1 over the block input held in registers, then over the skip channels: ONE job stream (c, g, k) ...
      WRing R;
      ring_start(R, J.c1[bi]);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float tm = lane_m1(x[c]), tp = lane_p1(x[c]);
        const float xm = hasL ? tm : 0.f, xp = hasR ? tp : 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int js = (c * G + g) * 3;
          acc[g][tapk(0)] = mfma4(wjob(R, J.c1[bi], js + 0), xm, acc[g][tapk(0)]);
          acc[g][tapk(1)] = mfma4(wjob(R, J.c1[bi], js + 1), x[c], acc[g][tapk(1)]);
          acc[g][tapk(2)] = mfma4(wjob(R, J.c1[bi], js + 2), xp, acc[g][tapk(2)]);
        }
      }
      if (bi == 1) DQ_PSTAMP(C * 1000 + PRE * 100 + CP, 7);
      if (bi == 0 && a.nblocks > 1) load_skip(1);  // block 1's skip channels travel while block 0 computes
      // (one wave-uniform branch per QUAD of skip channels -- cinB is a multiple of 4 -- not per channel: a branch ends the basic block and
      // the operand reads in flight are waited for there; the skip channels are a prefix, the stream is never resumed after a skipped quad)
#pragma unroll
      for (int cq = 0; cq < G; ++cq) {
        if (4 * cq < r.cinB) {
#pragma unroll
          for (int ci = 0; ci < 4; ++ci) {
            const int c = 4 * cq + ci;
            const float tm = lane_m1(xb[bi][c]), tp = lane_p1(xb[bi][c]);
            const float xm = hasL ? tm : 0.f, xp = hasR ? tp : 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) {
              const int js = ((C + c) * G + g) * 3;
              acc[g][tapk(0)] = mfma4(wjob(R, J.c1[bi], js + 0), xm, acc[g][tapk(0)]);
              acc[g][tapk(1)] = mfma4(wjob(R, J.c1[bi], js + 1), xb[bi][c], acc[g][tapk(1)]);
              acc[g][tapk(2)] = mfma4(wjob(R, J.c1[bi], js + 2), xp, acc[g][tapk(2)]);
            }
          }
        }
      }
      if (bi == 1) DQ_PSTAMP(C * 1000 + PRE * 100 + CP, 8);
      // ... and the 1x1 residual conv of cat(x, skip) as a second stream (c, g) (a block with skip channels has one)
      if (wr) {
        ring_start(R, J.rs[bi]);
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
          for (int g = 0; g < G; ++g) ar[g][c & (NAR - 1)] = mfma4(wjob(R, J.rs[bi], c * G + g), x[c], ar[g][c & (NAR - 1)]);
#pragma unroll
        for (int cq = 0; cq < G; ++cq) {
          if (4 * cq < r.cinB) {
#pragma unroll
            for (int ci = 0; ci < 4; ++ci) {
              const int c = 4 * cq + ci;
#pragma unroll
              for (int g = 0; g < G; ++g)
                ar[g][c & (NAR - 1)] = mfma4(wjob(R, J.rs[bi], (C + c) * G + g), xb[bi][c], ar[g][c & (NAR - 1)]);
            }
          }
        }
      }
      if (bi == 1) DQ_PSTAMP(C * 1000 + PRE * 100 + CP, 9);
      float u[C];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float4 t = prm4(pq + 0, g);
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) u[4 * g + i] = ONE ? acc[g][0][i] : (acc[g][0][i] + acc[g][1][i]) + (acc[g][TK - 1][i] + tv[i]);
      }
      if (r.u1 && live) {
#pragma unroll
        for (int c = 0; c < C; ++c) st(r.u1, c, obase, u[c]);
      }
      {
        float ssq = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) ssq = fmaf(u[c], u[c], ssq);
        const float inv = rms_inv(ssq, sqC);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float4 g4 = prm4(pq + 1, g), s4 = prm4(pq + 5, g), h4 = prm4(pq + 6, g);
          const float gv[4] = {g4.x, g4.y, g4.z, g4.w}, sv[4] = {s4.x, s4.y, s4.z, s4.w}, hv[4] = {h4.x, h4.y, h4.z, h4.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) u[4 * g + i] = silu_f(fmaf(u[4 * g + i] * inv * gv[i], sv[i], hv[i]));
        }
      }
      if (r.a1 && live) {
#pragma unroll
        for (int c = 0; c < C; ++c) st(r.a1, c, obase, u[c]);
      }
      if (bi == 1) DQ_PSTAMP(C * 1000 + PRE * 100 + CP, 10);
      // conv2 over the block-1 activation
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if constexpr (ONE) {
          const float4 t = prm4(pq + 2, g);
          acc[g][0] = f32x4{t.x, t.y, t.z, t.w};
        } else {
#pragma unroll
          for (int k = 0; k < 3; ++k) acc[g][k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
      }
      ring_start(R, J.c2[bi]);
#pragma unroll
      for (int c = 0; c < C; ++c) {
        const float tm = lane_m1(u[c]), tp = lane_p1(u[c]);
        const float xm = hasL ? tm : 0.f, xp = hasR ? tp : 0.f;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const int js = (c * G + g) * 3;
          acc[g][tapk(0)] = mfma4(wjob(R, J.c2[bi], js + 0), xm, acc[g][tapk(0)]);
          acc[g][tapk(1)] = mfma4(wjob(R, J.c2[bi], js + 1), u[c], acc[g][tapk(1)]);
          acc[g][tapk(2)] = mfma4(wjob(R, J.c2[bi], js + 2), xp, acc[g][tapk(2)]);
        }
      }
      if (bi == 1) DQ_PSTAMP(C * 1000 + PRE * 100 + CP, 11);
      float o[C];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const float4 t = prm4(pq + 2, g);
        const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) o[4 * g + i] = ONE ? acc[g][0][i] : (acc[g][0][i] + acc[g][1][i]) + (acc[g][TK - 1][i] + tv[i]);
      }
      if (r.u2 && live) {
#pragma unroll
        for (int c = 0; c < C; ++c) st(r.u2, c, obase, o[c]);
      }
      {
        float ssq = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) ssq = fmaf(o[c], o[c], ssq);
        const float inv = rms_inv(ssq, sqC);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float4 g4 = prm4(pq + 3, g);
          const float gv[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) o[4 * g + i] = silu_f(o[4 * g + i] * inv * gv[i]);
        }
      }
      if (wr) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const float4 t = prm4(pq + 4, g);
          const float tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float rs = ar[g][0][i] + ar[g][1][i];
            if constexpr (NAR == 4) rs += ar[g][2][i] + ar[g][3][i];
            o[4 * g + i] += rs + tv[i];
          }
        }
      } else {
#pragma unroll
        for (int c = 0; c < C; ++c) o[c] += x[c];
      }
      if (r.out && live) {
#pragma unroll
        for (int c = 0; c < C; ++c) st(r.out, c, obase, o[c]);
      }
      if constexpr (C == 4) {
        if (a.ep_w >= 0 && bi == a.nblocks - 1) {  // head: final_conv (1x1, 4 -> 1) and, while sampling, the DDIM update
          const float4 w4 = *reinterpret_cast<const float4*>(prm + C * 15);
          const float eb = prm[C * 15 + 4];
          const float ev = fmaf(w4.w, o[3], fmaf(w4.z, o[2], fmaf(w4.y, o[1], fmaf(w4.x, o[0], eb))));
          const unsigned eoff = ((row << ln) + p) * 4u;
          if (live) {
            float ep = ev;  // what eps_out receives: the network output, or -- while sampling with the x0 objective -- the derived eps
            if (a.x_t) {  // model.py:265-289, the arithmetic of k_ddim_step
              const float xv = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.x_t) + eoff);
              float x0;
              if (a.pred_x0) { x0 = ev; ep = (xv - dsa * x0) / dsb; }
              else           { x0 = (xv - dsb * ep) / dsa; }
              *reinterpret_cast<float*>(reinterpret_cast<char*>(a.x_out) + eoff) = dsap < 0.f ? x0 : dsap * x0 + dsbp * ep;
            }
            if (a.eps_out) *reinterpret_cast<float*>(reinterpret_cast<char*>(a.eps_out) + eoff) = ep;
          }
          if constexpr (TH && PRE != LEVEL_PRE_INIT) {  // training: model.py:361 and the first two steps of its backward, k_mse_fwd_bwd's arithmetic per element
            const float z = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.loss_z) + eoff);  // (not predicated: a clamped position)
            const float d = ev - z;
            if (live) {
              lacc += d * d;
              const float gv = d * a.loss_gscale;
              *reinterpret_cast<float*>(reinterpret_cast<char*>(a.grad_out) + eoff) = gv;
              st(a.dout, 0, obase, w4.x * gv); st(a.dout, 1, obase, w4.y * gv); st(a.dout, 2, obase, w4.z * gv); st(a.dout, 3, obase, w4.w * gv);
            }
          }
        }
      }
#pragma unroll
      for (int c = 0; c < C; ++c) x[c] = o[c];
    }
    DQ_PSTAMP(C * 1000 + PRE * 100 + CP, 5);
  }
  if constexpr (TH && PRE != LEVEL_PRE_INIT) {  // one partial sum per wave (fixed order inside the wave; the caller sums the waves in index order)
    const float t = wave_sum(lacc);
    if (lane == 0) a.loss_part[(blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)] = t;
  }
  DQ_PSTAMP(C * 1000 + PRE * 100 + CP, 6);
}

// stage input widths that are instantiated: a Downsample from C - 4 or C channels, an Upsample / k3 conv from C or C + 4 (channel
// widths that move by at most one step of 4 per level, as in the reference's configurations; anything else keeps the per-op path)
static bool level_cp_built(int C, int pre, int cp) {
  if (pre == LEVEL_PRE_NONE) return true;
  if (pre == LEVEL_PRE_INIT) return C == 4 && cp == 2;
  if (pre == LEVEL_PRE_DOWN) return cp == C || (cp == C - 4 && cp >= 4);
  return cp == C || (cp == C + 4 && cp <= 16);
}

static int num_cus() {
  static const int v = [] { int d = 0; hipDeviceProp_t pr; return (hipGetDevice(&d) == hipSuccess && hipGetDeviceProperties(&pr, d) == hipSuccess) ? pr.multiProcessorCount : 256; }();
  return v;
}

bool level_fwd_usable(int C, int n, int rows_per_sample, int pre_mode, int cp, int nblocks, const ResFwd* blk) {
  if (!(C == 4 || C == 8 || C == 12 || C == 16) || n < 1 || n > 64 || (n & (n - 1)) != 0 || rows_per_sample <= 1) return false;
  if (nblocks < 1 || nblocks > 2) return false;
  if (!level_cp_built(C, pre_mode, cp)) return false;
  if (pre_mode == LEVEL_PRE_UP && n < 2) return false;
  for (int b = 0; b < nblocks; ++b) {
    const ResFwd& r = blk[b];
    if (r.cinB % 4 != 0 || r.cinB < 0 || r.cinB > C || (r.cinB > 0) != (r.wr != nullptr)) return false;
  }
  return true;
}

static int level_img_item(const LevelFwd& a, LevelImgSrc* m) {
  auto poff = [&](const float* ptr) -> int { return ptr ? (int)(ptr - a.params) : -1; };
  m->pre = a.pre; m->G = a.C / 4; m->C = a.C; m->cp = a.pre == LEVEL_PRE_NONE ? 4 : a.cp;
  m->kp = a.pre == LEVEL_PRE_DOWN ? 4 : (a.pre == LEVEL_PRE_INIT ? 7 : 3);
  m->nblocks = a.nblocks; m->pw = poff(a.pw);
  for (int b = 0; b < 2; ++b) {
    const ResFwd& r = a.blk[b < a.nblocks ? b : 0];
    m->w1[b] = poff(r.w1); m->w2[b] = poff(r.w2); m->wr[b] = poff(r.wr); m->cin[b] = a.C + (b < a.nblocks ? r.cinB : 0);
  }
  return 0;
}
int64_t level_img_floats(const LevelFwd& a) {
  const int cin[2] = {a.C + a.blk[0].cinB, a.C + (a.nblocks > 1 ? a.blk[1].cinB : 0)};
  const bool wr[2] = {a.blk[0].wr != nullptr, a.nblocks > 1 && a.blk[1].wr != nullptr};
  return (int64_t)level_jobs(a.C, a.pre, a.pre == LEVEL_PRE_NONE ? 4 : a.cp, a.nblocks, cin, wr).total * 4;
}
int launch_level_images(const LevelFwd* calls, int count, hipStream_t s) {
  if (count == 0) return 0;
  DQ_REQUIRE(count <= LEVEL_IMG_MAX, "level images: too many launches");
  LevelImgMulti mm;
  int64_t mx = 0;
  for (int i = 0; i < count; ++i) {
    const LevelFwd& a = calls[i];
    DQ_REQUIRE(a.img && ((uintptr_t)a.img & 15) == 0 && a.params, "level images: missing / misaligned image buffer");
    DQ_REQUIRE(level_fwd_usable(a.C, a.n, a.rows_per_sample, a.pre, a.cp, a.nblocks, a.blk), "level images: unsupported shape");
    DQ_REQUIRE(a.params == calls[0].params, "level images: the launches must share one parameter buffer");
    level_img_item(a, &mm.it[i].m);
    mm.it[i].dst = const_cast<float*>(a.img);
    const int64_t fl = level_img_floats(a);
    DQ_REQUIRE(fl <= LEVEL_IMG_FLOATS, "level images: image larger than its slot");
    mx = std::max(mx, fl);
  }
  hipLaunchKernelGGL(k_level_images, dim3(cdiv(mx, 256), count), dim3(256), 0, s, mm, calls[0].params);
  DQ_LAUNCH_CHECK();
  return 0;
}

int launch_level_fwd(const LevelFwd& a, hipStream_t s) {
  DQ_REQUIRE(level_fwd_usable(a.C, a.n, a.rows_per_sample, a.pre, a.cp, a.nblocks, a.blk), "level_fwd: unsupported shape");
  DQ_REQUIRE(a.rows % a.rows_per_sample == 0 && a.in && a.params, "level_fwd: bad rows / missing input");
  DQ_REQUIRE(a.pre == LEVEL_PRE_NONE || (a.pw && a.pb), "level_fwd: the input stage needs its conv weight and bias");
  LevelFwdK k;
  auto poff = [&](const float* ptr) -> int { return ptr ? (int)(ptr - a.params) : -1; };
  DQ_REQUIRE(a.pre != LEVEL_PRE_INIT || (a.cond && a.ss_init && a.pre_out), "level_fwd: the first-layer stage needs the mixture, its scale / shift and h0");
  DQ_REQUIRE(!a.ew || (a.C == 4 && a.eb && (a.eps_out || a.x_t || a.loss_z) && (!a.x_t || (a.x_out && a.coef))), "level_fwd: incomplete head epilogue");
  DQ_REQUIRE(!a.loss_z || (a.ew && !a.x_t && a.loss_part && a.grad_out && a.dout && a.loss_parts_out), "level_fwd: incomplete training head");
  DQ_REQUIRE(!a.cat0_out || (a.pre == LEVEL_PRE_INIT && !a.loss_z), "level_fwd: cat0_out belongs to the first-layer stage");
  k.cat0_out = a.cat0_out;
  DQ_REQUIRE(!a.qs_noise || (a.cat0_out && a.qs_ab && a.qs_t), "level_fwd: q_sample in the first-layer stage needs the schedule, the timesteps and cat0_out");
  k.qs_noise = a.qs_noise; k.qs_ab = a.qs_ab; k.qs_t = a.qs_t; k.qs_norm = a.qs_norm;
  k.loss_z = a.loss_z; k.loss_part = a.loss_part; k.grad_out = a.grad_out; k.dout = a.dout; k.loss_gscale = a.loss_gscale;
  k.cond = a.cond; k.cm = a.cm; k.ca = a.ca; k.ss_init = a.ss_init ? (int)(a.ss_init - a.blk[0].ss) : 0;
  k.ep_w = poff(a.ew); k.ep_b = poff(a.eb); k.pred_x0 = a.pred_x0; k.eps_out = a.eps_out; k.x_t = a.x_t; k.x_out = a.x_out; k.coef = a.coef;
  k.step_ptr = a.step_ptr;
  k.in = a.in; k.pre_out = a.pre_out; k.pw = poff(a.pw); k.pb = poff(a.pb); k.nblocks = a.nblocks; k.rows_per_sample = a.rows_per_sample; k.n = a.n;
  const float* ssb = a.blk[0].ss;
  k.ss_stride = a.blk[0].ss_stride;
  for (int b = 0; b < 2; ++b) {
    const ResFwd& r = a.blk[b < a.nblocks ? b : 0];
    DQ_REQUIRE(r.w1 && r.b1 && r.g1 && r.w2 && r.b2 && r.g2 && r.ss && (r.cinB == 0 || (r.inB && r.wr && r.br)), "level_fwd: missing block operand");
    DQ_REQUIRE(r.ss_stride == k.ss_stride && r.ss >= ssb - (1 << 20) && r.ss <= ssb + (1 << 20), "level_fwd: the blocks' scale / shift vectors must share one buffer");
    for (const float* q : {r.w1, r.b1, r.g1, r.w2, r.b2, r.g2, r.wr, r.br})
      DQ_REQUIRE(!q || (q >= a.params && q - a.params < (1ll << 31)), "level_fwd: a parameter lies outside the flat parameter buffer");
    LevelBlkK& d = k.blk[b];
    d.w1 = poff(r.w1); d.b1 = poff(r.b1); d.g1 = poff(r.g1); d.w2 = poff(r.w2); d.b2 = poff(r.b2); d.g2 = poff(r.g2); d.wr = poff(r.wr); d.br = poff(r.br);
    d.ss_off = (int)(r.ss - ssb); d.cinB = b < a.nblocks ? r.cinB : 0;
    d.inB = r.inB; d.u1 = r.u1; d.a1 = r.a1; d.u2 = r.u2; d.out = r.out;
  }
  const int B = a.rows / a.rows_per_sample;
  const int tiles_ps = cdiv((int64_t)a.rows_per_sample * a.n, 64);
  const int64_t total = (int64_t)tiles_ps * B;
  DQ_REQUIRE(total < (1ll << 31), "level_fwd: too many tiles");
  DQ_REQUIRE((int64_t)a.rows * 2 * std::max(a.C, a.cp) * std::max(a.n, 2) * 4 < (1ll << 32), "level_fwd: tensors of 2^31 elements or more are not built (32-bit offsets)");
  int ln = 0;
  while ((1 << ln) < a.n) ++ln;
  const int cin[2] = {a.C + a.blk[0].cinB, a.C + a.blk[1].cinB};
  const bool wr[2] = {a.blk[0].wr != nullptr, a.blk[1].wr != nullptr};
  const int cp = a.pre == LEVEL_PRE_NONE ? 4 : a.cp;
  size_t lds = (size_t)level_jobs(a.C, a.pre, cp, a.nblocks, cin, wr).total * 16 + 256 * 4;  // + the parameter table: one value per thread
  if (a.img) {  // the kernel copies the image in whole rounds of 256 x 16 bytes (no guards): LDS region and slot must hold them
    const int G = a.C / 4, kp = a.pre == LEVEL_PRE_DOWN ? 4 : (a.pre == LEVEL_PRE_INIT ? 7 : 3);
    const int maxj = pad4(G * cp * kp) + 2 * (pad4(G * 2 * a.C * 3) + pad4(G * a.C * 3) + pad4(G * 2 * a.C));
    const int rounds = (maxj + 255) / 256;
    DQ_REQUIRE((int64_t)rounds * 1024 <= LEVEL_IMG_FLOATS, "level_fwd: image slot too small for the padded copy");
    lds = (size_t)rounds * 4096 + 256 * 4;
  }
  DQ_REQUIRE(lds <= 64 * 1024, "level_fwd: weight image too large");
  // one resident round: blocks per CU from the occupancy query, capped at 6 (at this kernel's ~106 scalar registers the hardware admits six
  // 256-thread blocks per CU where the query can say seven: MI355X_MICROARCH.md, Residency), never more blocks than tiles need
#define DQ_LVN(CC, PP, PC, NN) DQ_LVK((k_level_fwd<CC, PP, PC, NN>))
#define DQ_LVK(KERNEL)                                                                                                        \
  {                                                                                                                           \
    const int nb = occ_blocks_per_cu((const void*)KERNEL, 256, lds);  /* keyed on (instantiation, lds, device) */              \
    if (nb < 0) return 1;                                                                                                     \
    const int occ = std::min(nb, 6);                                                                                          \
    const int gx = std::max(1, std::min(occ * num_cus() / B, (tiles_ps + 3) / 4));  /* workgroups per sample */                \
    if (a.loss_z) {                                                                                                           \
      DQ_REQUIRE((int64_t)gx * B * 4 <= LEVEL_LOSS_PARTS, "level_fwd: more waves than the loss partial-sum scratch holds");      \
      *a.loss_parts_out = gx * B * 4;                                                                                         \
    }                                                                                                                         \
    hipLaunchKernelGGL(KERNEL, dim3(gx, B), dim3(256), lds, s, k, a.params, ssb, tiles_ps, (int)total, ln, a.img);             \
    DQ_LAUNCH_CHECK();                                                                                                        \
    return 0;                                                                                                                 \
  }
#define DQ_LV(CC, PP, PC)                                                                                                     \
  if (a.C == CC && a.pre == PP && cp == PC) {                                                                                 \
    if (CC == 4 && a.n == 64) DQ_LVN(CC, PP, PC, (CC == 4))                                                                    \
    DQ_LVN(CC, PP, PC, false)                                                                                                 \
  }
  if (a.loss_z) {  // the training head has its own instantiations (the network's final block: k3 stage conv from 4 channels)
    DQ_REQUIRE(a.C == 4 && a.pre == LEVEL_PRE_S1 && cp == 4, "level_fwd: the training head is built for the (4, k3 conv, 4) launch");
    if (a.n == 64) DQ_LVK((k_level_fwd<4, LEVEL_PRE_S1, 4, true, true>))
    DQ_LVK((k_level_fwd<4, LEVEL_PRE_S1, 4, false, true>))
  }
  if (a.cat0_out) {  // the first level of a train step: the INIT stage that also stores its input
    DQ_REQUIRE(a.C == 4 && cp == 2, "level_fwd: the first-layer stage is built for (4, init conv, 2)");
    if (a.n == 64) DQ_LVK((k_level_fwd<4, LEVEL_PRE_INIT, 2, true, true>))
    DQ_LVK((k_level_fwd<4, LEVEL_PRE_INIT, 2, false, true>))
  }
  DQ_LV(4, LEVEL_PRE_NONE, 4) DQ_LV(8, LEVEL_PRE_NONE, 4) DQ_LV(12, LEVEL_PRE_NONE, 4) DQ_LV(16, LEVEL_PRE_NONE, 4)
  DQ_LV(4, LEVEL_PRE_DOWN, 4) DQ_LV(8, LEVEL_PRE_DOWN, 4) DQ_LV(8, LEVEL_PRE_DOWN, 8) DQ_LV(12, LEVEL_PRE_DOWN, 8) DQ_LV(12, LEVEL_PRE_DOWN, 12)
  DQ_LV(16, LEVEL_PRE_DOWN, 12) DQ_LV(16, LEVEL_PRE_DOWN, 16)
  DQ_LV(4, LEVEL_PRE_UP, 4) DQ_LV(4, LEVEL_PRE_UP, 8) DQ_LV(8, LEVEL_PRE_UP, 8) DQ_LV(8, LEVEL_PRE_UP, 12) DQ_LV(12, LEVEL_PRE_UP, 12) DQ_LV(12, LEVEL_PRE_UP, 16)
  DQ_LV(16, LEVEL_PRE_UP, 16)
  DQ_LV(4, LEVEL_PRE_INIT, 2)
  DQ_LV(4, LEVEL_PRE_S1, 4) DQ_LV(4, LEVEL_PRE_S1, 8) DQ_LV(8, LEVEL_PRE_S1, 8) DQ_LV(8, LEVEL_PRE_S1, 12) DQ_LV(12, LEVEL_PRE_S1, 12) DQ_LV(12, LEVEL_PRE_S1, 16)
  DQ_LV(16, LEVEL_PRE_S1, 16)
#undef DQ_LV
#undef DQ_LVN
#undef DQ_LVK
  set_error("level_fwd: unsupported (C, stage, stage input width)");
  return 2;
}

}  // namespace dq
