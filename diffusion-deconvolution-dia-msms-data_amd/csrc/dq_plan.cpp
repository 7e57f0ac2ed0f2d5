// Parameter layout + op list for UNet1d(simple=True, conditional=True, channels=1, init_cond_channels=1,
// attn_cond_channels=1).  Order and names follow the reference module registration order
// (dquartic/model/unet1d.py:949-1082): init_conv, time_mlp, init_cond_proj, attn_cond_proj, downs, ups,
// mid_block1, mid_attn, mid_block2, final_res_block, final_conv.  The non-trainable RoPE frequencies
// (mid_attn.fn.fn.rotary_emb.freqs) are NOT part of the flat buffer.
#include "dq_plan.h"
#include "dq_options.h"
#include <atomic>
#include <cstring>
#include <cstdlib>

namespace dq {
namespace {

struct Builder {
  Plan& p;
  // align4: start the tensor on a 16-byte boundary of the flat buffer (the tensors the GEMM reads as an operand; the few floats
  // skipped stay zero in the parameter, gradient and moment buffers and belong to no tensor)
  bool align4 = false;
  int64_t add(const std::string& name, std::initializer_list<int64_t> shape) {
    if (align4) p.total_floats = (p.total_floats + 3) / 4 * 4;
    ParamInfo pi;
    pi.name = name;
    pi.offset = p.total_floats;
    pi.ndim = (int)shape.size();
    pi.numel = 1;
    int i = 0;
    for (auto s : shape) { pi.shape[i++] = s; pi.numel *= s; }
    for (; i < 4; ++i) pi.shape[i] = 1;
    p.total_floats += pi.numel;
    p.params.push_back(pi);
    return pi.offset;
  }
  ConvP conv(const std::string& pre, int cout, int cin, int k, bool bias = true) {
    ConvP c;
    c.cin = cin; c.cout = cout; c.k = k;
    c.w = add(pre + ".weight", {cout, cin, k});
    c.b = bias ? add(pre + ".bias", {cout}) : -1;
    return c;
  }
  ResP res(const std::string& pre, int cin, int cout) {
    ResP r;
    r.cin = cin; r.cout = cout;
    r.mlp_w = add(pre + ".mlp.1.weight", {2 * cout, p.time_dim});
    r.mlp_b = add(pre + ".mlp.1.bias", {2 * cout});
    r.c1 = conv(pre + ".block1.proj", cout, cin, 3);
    r.g1 = add(pre + ".block1.norm.g", {1, cout, 1});
    r.c2 = conv(pre + ".block2.proj", cout, cout, 3);
    r.g2 = add(pre + ".block2.norm.g", {1, cout, 1});
    if (cin != cout) r.res = conv(pre + ".res_conv", cout, cin, 1);
    r.ss_off = p.ss_total;
    p.ss_lins.push_back({r.mlp_w, r.mlp_b, 2 * cout, r.ss_off});
    p.ss_total += 2 * cout;
    return r;
  }
  LAP la(const std::string& pre, int C) {
    LAP a;
    a.C = C;
    a.qkv_w = add(pre + ".fn.fn.to_qkv.weight", {3 * HID, C, 1});
    a.out_w = add(pre + ".fn.fn.to_out.0.weight", {C, HID, 1});
    a.out_b = add(pre + ".fn.fn.to_out.0.bias", {C});
    a.g_out = add(pre + ".fn.fn.to_out.1.g", {1, C, 1});
    a.g_pre = add(pre + ".fn.norm.g", {1, C, 1});
    return a;
  }
};

}  // namespace

std::string build_plan(Plan& p, int dim, int n_mults, const int* mults, int mz, int T) {
  if (dim < 4 || dim % 4) return "dim must be a positive multiple of 4";
  if (n_mults < 1 || n_mults > 10) return "dim_mults must have 1..10 entries";
  const int L = n_mults;
  if (mz <= 0 || mz % (1 << (L - 1))) return "MZ (downsample_dim) must be divisible by 2**(len(dim_mults)-1)";
  if (T < 1) return "num_timesteps must be >= 1";
  p = Plan();
  p.dim = dim; p.levels = L; p.mz = mz; p.time_dim = 4 * dim; p.T = T;
  p.dims.push_back(dim);
  for (int i = 0; i < L; ++i) {
    if (mults[i] < 1) return "dim_mults entries must be >= 1";
    p.dims.push_back(dim * mults[i]);
  }
  for (int d : p.dims)
    if (d > 16) return "channel widths above 16 are not built (dim*mult <= 16)";
  p.mid_n = mz >> (L - 1);
  p.mid_c = p.dims[L] * p.mid_n;
  // 16 / 32 channels: the register-resident bottleneck kernels; anything else -- the 64 channels of BASELINE configs[4] (256 x 2000 windows),
  // the 10,000 of the reference's shipped downsample_dim 40000 (unet1d.py:1027-1029) -- the wide path (im2col + matrix-core GEMM + channel-axis
  // norm, k_wide.hip).  Round 4 moved 64 channels there: a thread of the register-resident kernels ran 64 x 64 x 3 dependent FMAs per position
  // (k_conv_fwd<64, 3, 0> 241 us per conv, k_block_bwd<64> with 596 spilled registers); measured at batch 8: train step 19.5 -> 18.0 ms,
  // sampling step 4.7 -> 4.0 ms.
  p.wide_mid = !(p.mid_c == 16 || p.mid_c == 32);
  if (p.wide_mid && p.mid_c % 4) return "bottleneck width dims[-1]*MZ/2**(L-1) must be a multiple of 4";
  p.cond_dim = 2 * dim;
  if (p.cond_dim != 8) return "attn_cond_init_dim (2*dim) must be 8";
  if (p.time_dim != 16) return "time_dim (4*dim) must be 16";

  Builder b{p};
  p.init_conv = b.conv("init_conv", dim, 2, 7);
  p.t1_w = b.add("time_mlp.1.weight", {p.time_dim, dim});
  p.t1_b = b.add("time_mlp.1.bias", {p.time_dim});
  p.t2_w = b.add("time_mlp.3.weight", {p.time_dim, p.time_dim});
  p.t2_b = b.add("time_mlp.3.bias", {p.time_dim});
  p.icp_w = b.add("init_cond_proj.to_scale_shift.1.weight", {2, p.time_dim});
  p.icp_b = b.add("init_cond_proj.to_scale_shift.1.bias", {2});
  p.ss_init = p.ss_total;
  p.ss_lins.push_back({p.icp_w, p.icp_b, 2, p.ss_init});
  p.ss_total += 2;
  p.ms1_c0 = b.conv("attn_cond_proj.1.0", p.cond_dim, 1, 7);
  p.ms1_c1 = b.conv("attn_cond_proj.1.2", p.cond_dim, p.cond_dim, 1);

  int n = mz;
  for (int lv = 0; lv < L; ++lv) {
    const std::string pre = "downs." + std::to_string(lv);
    const int din = p.dims[lv], dout = p.dims[lv + 1];
    LevelP l;
    l.n = n;
    l.last = lv == L - 1;
    l.r0 = b.res(pre + ".0", din, din);
    l.r1 = b.res(pre + ".1", din, din);
    l.la = b.la(pre + ".2", din);
    l.resample = b.conv(pre + ".3", dout, din, l.last ? 3 : 4);
    l.n_next = l.last ? n : n / 2;
    n = l.n_next;
    p.downs.push_back(l);
  }
  for (int ui = 0; ui < L; ++ui) {
    const std::string pre = "ups." + std::to_string(ui);
    const int lv = L - 1 - ui;
    const int din = p.dims[lv], dout = p.dims[lv + 1];
    LevelP l;
    l.n = p.downs[lv].n;
    l.last = ui == L - 1;
    l.r0 = b.res(pre + ".0", dout + din, dout);
    l.r1 = b.res(pre + ".1", dout + din, dout);
    l.la = b.la(pre + ".2", dout);
    l.resample = b.conv(l.last ? pre + ".3" : pre + ".3.1", din, dout, 3);
    l.n_next = l.last ? l.n : l.n * 2;
    p.ups.push_back(l);
  }
  b.align4 = p.wide_mid;
  p.mid1 = b.res("mid_block1", p.mid_c, p.mid_c);
  p.qv_w = b.add("mid_attn.fn.fn.to_qv.weight", {2 * HID, p.mid_c, 1});
  p.k_w = b.add("mid_attn.fn.fn.to_k.weight", {HID, p.cond_dim, 1});
  p.ao_w = b.add("mid_attn.fn.fn.to_out.weight", {p.mid_c, HID, 1});
  p.ao_b = b.add("mid_attn.fn.fn.to_out.bias", {p.mid_c});
  p.ag = b.add("mid_attn.fn.norm.g", {1, p.mid_c, 1});
  p.mid2 = b.res("mid_block2", p.mid_c, p.mid_c);
  b.align4 = false;
  p.fin = b.res("final_res_block", 2 * dim, dim);
  p.final_conv = b.conv("final_conv", 1, dim, 1);
  return "";
}

// A one-block plan for the stand-alone ResnetBlock entry points (dq_resblock_*): the block's tensors in state_dict order
// [mlp.1.weight, mlp.1.bias, block1.proj.weight, block1.proj.bias, block1.norm.g, block2.proj.weight, block2.proj.bias,
// block2.norm.g (, res_conv.weight, res_conv.bias iff cin != cout)] as one flat buffer.
void build_resblock_plan(Plan& p, ResP& r, int cin, int cout) {
  p = Plan();
  p.dim = 4; p.time_dim = 16;
  Builder b{p};
  r = b.res("block", cin, cout);
}

}  // namespace dq

namespace dq {
namespace {
std::atomic<int64_t> g_options[OPT_COUNT] = {{-1}, {-1}};
std::atomic<unsigned> g_options_epoch{0};
const char* const g_option_keys[OPT_COUNT] = {"la_small_min_rows", "la_rows_bwd_min_rows"};
}  // namespace
int64_t option(Option o) { return g_options[o].load(std::memory_order_relaxed); }
unsigned options_epoch() { return g_options_epoch.load(std::memory_order_relaxed); }
int option_index(const char* key) {
  for (int i = 0; key && i < OPT_COUNT; ++i)
    if (!std::strcmp(key, g_option_keys[i])) return i;
  return -1;
}
void set_option(int index, int64_t value) {
  g_options[index].store(value, std::memory_order_relaxed);
  g_options_epoch.fetch_add(1, std::memory_order_relaxed);
}
}  // namespace dq
