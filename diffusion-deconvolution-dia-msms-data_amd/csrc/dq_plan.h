// Network plan: the flat parameter layout (one fp32 buffer, tensors in the reference's
// state_dict order, names = state_dict keys) and the op list the host code walks.
// Reference structure: dquartic/model/unet1d.py:918-1084 (UNet1d.__init__, simple=True, conditional=True).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

namespace dq {

constexpr int HEADS = 4;      // unet1d.py:932
constexpr int DIM_HEAD = 32;  // unet1d.py:933
constexpr int HID = HEADS * DIM_HEAD;

struct ParamInfo {
  std::string name;
  int64_t offset;  // in floats, into the flat trainable buffer
  int64_t numel;
  int ndim;
  int64_t shape[4];
};

struct ConvP {  // Conv1d weight (cout, cin, k) + bias (cout) ; b < 0 => no bias
  int64_t w = -1, b = -1;
  int cin = 0, cout = 0, k = 0;
};

struct ResP {  // ResnetBlock (unet1d.py:271-323)
  int64_t mlp_w = -1, mlp_b = -1;  // Linear(time_dim, 2*cout)
  ConvP c1, c2, res;               // res.cout == 0 => identity
  int64_t g1 = -1, g2 = -1;
  int cin = 0, cout = 0;
  int ss_off = 0;  // offset of this block's [scale(cout) | shift(cout)] in the per-sample ss vector
};

struct LAP {  // Residual(PreNorm(LinearAttention)) (unet1d.py:446-496)
  int64_t qkv_w = -1, out_w = -1, out_b = -1, g_out = -1, g_pre = -1;
  int C = 0;
};

struct LevelP {
  ResP r0, r1;
  LAP la;
  ConvP resample;
  int n = 0;        // m/z length the level's blocks run at
  int n_next = 0;   // m/z length after the level's resample conv
  bool last = false;
};

struct Plan {
  int dim = 0, levels = 0, mz = 0, time_dim = 0, T = 0;
  std::vector<int> dims;  // [init_dim, dim*mult...]
  int mid_n = 0;          // downsampled_n (unet1d.py:1027)
  int mid_c = 0;          // mid_dim * downsampled_n
  bool wide_mid = false;  // mid_c not in {16, 32, 64}: the bottleneck runs on im2col + GEMM + channel-axis norm (k_wide.hip)
  int cond_dim = 0;       // attn_cond_init_dim = 2*dim (unet1d.py:970)
  int ss_total = 0;       // floats per sample in the ss vector (all ResnetBlock mlps + init_cond_proj)
  int ss_init = 0;        // offset of init_cond_proj's [scale, shift]

  std::vector<ParamInfo> params;
  int64_t total_floats = 0;

  ConvP init_conv;                  // (dim, 2, 7)
  int64_t t1_w, t1_b, t2_w, t2_b;   // time_mlp.1 / .3
  int64_t icp_w, icp_b;             // init_cond_proj.to_scale_shift.1  (2, time_dim)
  ConvP ms1_c0, ms1_c1;             // attn_cond_proj.1.0 (k7) / .1.2 (k1)
  std::vector<LevelP> downs, ups;
  ResP mid1, mid2;
  int64_t qv_w, k_w, ao_w, ao_b, ag;  // mid_attn: to_qv (256,mid_c), to_k (128,cond_dim), to_out (mid_c,128)+b, norm.g
  ResP fin;
  ConvP final_conv;

  // every Linear(time_dim -> m) that hangs off silu(temb): rows of one virtual (ss_total x time_dim) matrix
  struct SSLin { int64_t w, b; int rows; int ss_off; };
  std::vector<SSLin> ss_lins;
};

// Builds the plan; returns empty string on success, else an error message.
std::string build_plan(Plan& p, int dim, int n_mults, const int* mults, int mz, int T);

void build_resblock_plan(Plan& p, ResP& r, int cin, int cout);  // one ResnetBlock as its own flat parameter buffer

}  // namespace dq
