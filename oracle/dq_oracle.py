"""CPU oracle for the dquartic DDIM hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product path (``dquartic.*`` on top of ``libdq_hip.so``) never does and fails loudly when
the HIP library is missing.

This is a from-scratch functional restatement (plain PyTorch fp32 on the CPU) of the reference's
arithmetic for the one hot path; every function cites the reference file:line it follows
(paths relative to the reference checkout).  Parameters are passed as a flat ``dict`` that uses the
reference's ``state_dict`` key names, so reference checkpoints and the build's checkpoints are both
usable as-is.

Batched semantics (the reference network only runs at B = 1): "the B = 1 reference applied
independently to every sample" -- the time-embedding scale/shift of sample b is applied to all RT rows
of sample b, and the loss is the mean over samples of the B = 1 loss (a scalar).

Parity status: pinned against outputs of the reference itself, imported in the build container by
``oracle/make_golden.py`` (fixtures under ``tests/golden/``).  The one exception is RoPE: the reference
takes it from the un-vendored third-party package ``rotary_embedding_torch ^0.8.4``
(pyproject.toml:23; call sites unet1d.py:16,529,560-561); ``rope_rotate`` restates that package's
published default ('lang') algorithm and is "parity unpinned" -- the goldens pin everything else with
RoPE both as this stand-in and disabled.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Params = Dict[str, torch.Tensor]

HEADS = 4  # unet1d.py:932 (attn_heads default; the CLI never overrides it, cli.py:90-100)
DIM_HEAD = 32  # unet1d.py:933
RMS_EPS = 1e-12  # F.normalize default eps, unet1d.py:140


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class UNetConfig:
    """Mirror of the ``UNet1d(simple=True, conditional=True)`` constructor surface (unet1d.py:918-939)."""

    dim: int = 4
    dim_mults: Tuple[int, ...] = (1, 2, 2, 3, 3, 4, 4)
    channels: int = 1
    init_cond_channels: int = 1
    attn_cond_channels: int = 1
    downsample_dim: int = 64
    theta: float = 10000.0

    @property
    def dims(self) -> List[int]:  # unet1d.py:951
        return [self.dim] + [self.dim * m for m in self.dim_mults]

    @property
    def in_out(self) -> List[Tuple[int, int]]:  # unet1d.py:952
        d = self.dims
        return list(zip(d[:-1], d[1:]))

    @property
    def time_dim(self) -> int:  # unet1d.py:956
        return self.dim * 4

    @property
    def levels(self) -> int:
        return len(self.dim_mults)

    @property
    def downsampled_n(self) -> int:  # unet1d.py:1027
        return self.downsample_dim // (2 ** (self.levels - 1))

    @property
    def mid_channels(self) -> int:  # unet1d.py:1028-1029
        return self.dims[-1] * self.downsampled_n

    @property
    def attn_cond_dim(self) -> int:  # unet1d.py:970
        return self.dim * 2


# --------------------------------------------------------------------------------------
# schedule and helpers (model.py:14-148, 196-213)
# --------------------------------------------------------------------------------------
def linear_betas(T: int, beta_start: float = 1e-4, beta_end: float = 0.02) -> torch.Tensor:
    """model.py:14-29 -- fp64 linspace."""
    return torch.linspace(beta_start, beta_end, T, dtype=torch.float64)


def cosine_betas(T: int, s: float = 0.008) -> torch.Tensor:
    """model.py:32-54 -- fp64 cosine schedule, clipped to [0, 0.999]."""
    x = torch.linspace(0, T, T + 1, dtype=torch.float64)
    ac = torch.cos(((x / T) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - ac[1:] / ac[:-1], 0, 0.999)


def make_schedule(T: int = 1000, kind: str = "cosine") -> Dict[str, torch.Tensor]:
    """model.py:196-213: betas fp64 -> fp32; alphas = 1 - betas (fp32); alpha_bars = cumprod in fp32."""
    betas = (linear_betas(T) if kind == "linear" else cosine_betas(T)).to(torch.float32)
    alphas = (1.0 - betas).to(torch.float32)
    alpha_bars = torch.cumprod(alphas, dim=0).to(torch.float32)
    return {"betas": betas, "alphas": alphas, "alpha_bars": alpha_bars}


def sampler_timesteps(T: int, num_steps: int) -> List[int]:
    """model.py:313: ``linspace(T-1, 0, num_steps, dtype=long)``."""
    return torch.linspace(T - 1, 0, num_steps, dtype=torch.long).tolist()


def normalize(x):  # model.py:89-99
    return x * 2 - 1


def unnormalize(x):  # model.py:102-112
    return (x + 1) * 0.5


def q_sample(alpha_bars: torch.Tensor, x0: torch.Tensor, t: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """model.py:239-242 (K0).  x0/noise (B,RT,MZ), t (B,) int64."""
    a = torch.sqrt(alpha_bars[t])[:, None, None]
    b = torch.sqrt(1.0 - alpha_bars[t])[:, None, None]
    return a * x0 + b * noise


def ddim_update(alpha_bars: torch.Tensor, x_t: torch.Tensor, eps: torch.Tensor, t: int) -> torch.Tensor:
    """model.py:265-289 (K9), pred_type='eps'.  Lands on alpha_bars[t-1] whatever the stride."""
    ab = alpha_bars[t]
    sa, sb = torch.sqrt(ab), torch.sqrt(1.0 - ab)
    x0 = (x_t - sb * eps) / sa
    if t > 0:
        abp = alpha_bars[t - 1]
        return torch.sqrt(abp) * x0 + torch.sqrt(1.0 - abp) * eps
    return x0


def ddim_update_x0(alpha_bars: torch.Tensor, x_t: torch.Tensor, x0_pred: torch.Tensor, t: int):
    """model.py:274-289, pred_type='x0': the network output is x0; eps is derived from it.  Returns (x_prev, eps)."""
    ab = alpha_bars[t]
    sa, sb = torch.sqrt(ab), torch.sqrt(1.0 - ab)
    eps = (x_t - sa * x0_pred) / sb
    if t > 0:
        abp = alpha_bars[t - 1]
        return torch.sqrt(abp) * x0_pred + torch.sqrt(1.0 - abp) * eps, eps
    return x0_pred, eps


# --------------------------------------------------------------------------------------
# network blocks (unet1d.py)
# --------------------------------------------------------------------------------------
def rmsnorm(x: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """unet1d.py:140: F.normalize over channels (x / max(||x||, 1e-12)) * g * sqrt(C).  x (R,C,n), g (1,C,1)."""
    nrm = torch.sqrt((x * x).sum(dim=1, keepdim=True)).clamp_min(RMS_EPS)
    return x / nrm * g * (x.shape[1] ** 0.5)


def sinusoidal_emb(t: torch.Tensor, dim: int, theta: float = 10000.0) -> torch.Tensor:
    """unet1d.py:211-218."""
    half = dim // 2
    k = math.log(theta) / (half - 1)
    f = torch.exp(torch.arange(half, dtype=torch.float32) * -k)
    e = t[:, None] * f[None, :]  # int64 * fp32 -> fp32
    return torch.cat((e.sin(), e.cos()), dim=-1)


def time_mlp(p: Params, t: torch.Tensor, cfg: UNetConfig) -> torch.Tensor:
    """unet1d.py:956-960: sinusoidal -> Linear(dim,4dim) -> exact GELU -> Linear.  Returns (B, time_dim)."""
    e = sinusoidal_emb(t, cfg.dim, cfg.theta).to(p["time_mlp.1.weight"].dtype)  # (float64 parameters: the yardstick runs of the tests)
    h = F.linear(e, p["time_mlp.1.weight"], p["time_mlp.1.bias"])
    h = F.gelu(h)
    return F.linear(h, p["time_mlp.3.weight"], p["time_mlp.3.bias"])


def _scale_shift(p: Params, prefix: str, temb: torch.Tensor, rows_per_sample: int):
    """ResnetBlock.mlp: SiLU -> Linear(time_dim, 2*C_out) -> chunk (unet1d.py:292-296, 315-318).
    Returns (scale, shift) each (B*rows_per_sample, C, 1): sample b's vector on all of its rows."""
    ss = F.linear(F.silu(temb), p[prefix + ".mlp.1.weight"], p[prefix + ".mlp.1.bias"])
    ss = ss.repeat_interleave(rows_per_sample, dim=0)[:, :, None]
    return ss.chunk(2, dim=1)


def block(p: Params, prefix: str, x: torch.Tensor, scale_shift=None) -> torch.Tensor:
    """Block: Conv1d(k3,p1) -> RMSNorm -> (scale+1, shift) -> SiLU (unet1d.py:259-268)."""
    x = F.conv1d(x, p[prefix + ".proj.weight"], p[prefix + ".proj.bias"], padding=1)
    x = rmsnorm(x, p[prefix + ".norm.g"])
    if scale_shift is not None:
        s, sh = scale_shift
        x = x * (s + 1) + sh
    return F.silu(x)


def resnet_block(p: Params, prefix: str, x: torch.Tensor, temb: torch.Tensor, rows_per_sample: int) -> torch.Tensor:
    """ResnetBlock (unet1d.py:302-323): scale/shift in block1 only; res_conv is 1x1 iff C_in != C_out."""
    ss = _scale_shift(p, prefix, temb, rows_per_sample)
    h = block(p, prefix + ".block1", x, ss)
    h = block(p, prefix + ".block2", h)
    if (prefix + ".res_conv.weight") in p:
        x = F.conv1d(x, p[prefix + ".res_conv.weight"], p[prefix + ".res_conv.bias"])
    return h + x


def linear_attention(p: Params, prefix: str, x: torch.Tensor) -> torch.Tensor:
    """Residual(PreNorm(LinearAttention)) (unet1d.py:64-79, 143-176, 466-496).  ``prefix`` is e.g. 'downs.0.2'."""
    R, C, n = x.shape
    y = rmsnorm(x, p[prefix + ".fn.norm.g"])
    qkv = F.conv1d(y, p[prefix + ".fn.fn.to_qkv.weight"])  # no bias
    q, k, v = (t.reshape(R, HEADS, DIM_HEAD, n) for t in qkv.chunk(3, dim=1))
    q = q.softmax(dim=-2) * (DIM_HEAD ** -0.5)
    k = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(R, HEADS * DIM_HEAD, n)
    out = F.conv1d(out, p[prefix + ".fn.fn.to_out.0.weight"], p[prefix + ".fn.fn.to_out.0.bias"])
    out = rmsnorm(out, p[prefix + ".fn.fn.to_out.1.g"])
    return out + x


def rope_freqs(dim: int = DIM_HEAD // 2, theta: float = 10000.0) -> torch.Tensor:
    """rotary_embedding_torch 0.8.x, freqs_for='lang': 1 / theta^(arange(0,dim,2)/dim).  PARITY UNPINNED."""
    return 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))


def rope_rotate(t: torch.Tensor, freqs: torch.Tensor) -> torch.Tensor:
    """``RotaryEmbedding.rotate_queries_or_keys`` on (B,H,N,D): positions 0..N-1, each freq repeated for an
    interleaved pair, applied to the first 2*len(freqs) channels as x*cos + rotate_half(x)*sin with
    rotate_half((a,b)) = (-b,a) on adjacent pairs; remaining channels pass through.  PARITY UNPINNED."""
    n = t.shape[-2]
    ang = torch.arange(n, dtype=t.dtype)[:, None] * freqs.to(t.dtype)[None, :]
    ang = ang.repeat_interleave(2, dim=-1)  # (N, rot)
    rot = ang.shape[-1]
    tm, tr = t[..., :rot], t[..., rot:]
    pr = tm.reshape(*tm.shape[:-1], rot // 2, 2)
    rh = torch.stack((-pr[..., 1], pr[..., 0]), dim=-1).reshape(tm.shape)
    return torch.cat((tm * ang.cos() + rh * ang.sin(), tr), dim=-1)


def mid_attention(p: Params, x: torch.Tensor, cond: torch.Tensor, use_rope: bool = True) -> torch.Tensor:
    """Residual(PreNorm(Attention(use_xattn=True))) (unet1d.py:552-567, 428-443, 1030-1042).
    x (B,Cm,RT); cond (B,8,RT) are the MS1 features that become the keys."""
    B, Cm, n = x.shape
    y = rmsnorm(x, p["mid_attn.fn.norm.g"])
    qv = F.conv1d(y, p["mid_attn.fn.fn.to_qv.weight"])
    q, v = (t.reshape(B, HEADS, DIM_HEAD, n).transpose(2, 3) for t in qv.chunk(2, dim=1))  # b h n c
    k = F.conv1d(cond, p["mid_attn.fn.fn.to_k.weight"]).reshape(B, HEADS, DIM_HEAD, n).transpose(2, 3)
    if use_rope:
        fr = p.get("mid_attn.fn.fn.rotary_emb.freqs", None)
        fr = rope_freqs() if fr is None else fr
        q, k = rope_rotate(q, fr), rope_rotate(k, fr)
    sim = torch.einsum("bhid,bhjd->bhij", q, k) * (DIM_HEAD ** -0.5)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("bhij,bhjd->bhid", attn, v)
    out = out.transpose(2, 3).reshape(B, HEADS * DIM_HEAD, n)  # b (h d) n
    out = F.conv1d(out, p["mid_attn.fn.fn.to_out.weight"], p["mid_attn.fn.fn.to_out.bias"])
    return out + x


def ms1_features(p: Params, ms1: torch.Tensor) -> torch.Tensor:
    """attn_cond path for a 1-D chromatogram (unet1d.py:1120-1130, 970-981): (B,RT) -> (B,1,RT) ->
    Conv1d(1->8,k7,p3) -> exact GELU -> Conv1d(8->8,k1)."""
    a = ms1[:, None, :]
    a = F.conv1d(a, p["attn_cond_proj.1.0.weight"], p["attn_cond_proj.1.0.bias"], padding=3)
    a = F.gelu(a)
    return F.conv1d(a, p["attn_cond_proj.1.2.weight"], p["attn_cond_proj.1.2.bias"])


def unet_forward(
    p: Params,
    cfg: UNetConfig,
    x: torch.Tensor,
    time: torch.Tensor,
    init_cond: torch.Tensor,
    attn_cond: torch.Tensor,
    use_rope: bool = True,
    taps: Optional[dict] = None,
) -> torch.Tensor:
    """UNet1d.forward, simple=True, conditional=True (unet1d.py:1086-1166) with batched semantics.
    x, init_cond (B,RT,MZ); time (B,) int64; attn_cond (B,RT).  Returns (B,RT,MZ)."""
    B, RT, MZ = x.shape
    if MZ != cfg.downsample_dim or MZ % (2 ** (cfg.levels - 1)) != 0:
        raise ValueError("MZ must equal downsample_dim and be divisible by 2**(levels-1)")
    L = cfg.levels
    xr = x.reshape(B * RT, 1, MZ)  # row index b*RT + rt (unet1d.py:1100-1104)
    temb = time_mlp(p, time, cfg)  # (B, 16)
    # K2 (unet1d.py:1107-1118, 662-678): per-sample scalar affine on the mixture, cond is channel 0
    ss = F.linear(F.silu(temb), p["init_cond_proj.to_scale_shift.1.weight"], p["init_cond_proj.to_scale_shift.1.bias"])
    sc, sh = ss.repeat_interleave(RT, dim=0)[:, :, None].chunk(2, dim=1)
    ic = init_cond.reshape(B * RT, 1, MZ) * (sc + 1) + sh
    h0 = F.conv1d(torch.cat((ic, xr), dim=1), p["init_conv.weight"], p["init_conv.bias"], padding=3)
    r = h0
    cond = ms1_features(p, attn_cond)  # K3
    if taps is not None:
        taps["temb"], taps["init"], taps["ms1f"] = temb, h0, cond

    hs = []
    cur = h0
    for lv in range(L):
        pre = f"downs.{lv}"
        cur = resnet_block(p, pre + ".0", cur, temb, RT)
        hs.append(cur)
        cur = resnet_block(p, pre + ".1", cur, temb, RT)
        cur = linear_attention(p, pre + ".2", cur)
        hs.append(cur)
        if lv < L - 1:  # Downsample k4 s2 p1 (unet1d.py:110)
            cur = F.conv1d(cur, p[pre + ".3.weight"], p[pre + ".3.bias"], stride=2, padding=1)
        else:  # last level: k3 p1 (unet1d.py:1021)
            cur = F.conv1d(cur, p[pre + ".3.weight"], p[pre + ".3.bias"], padding=1)
        if taps is not None:
            taps[f"down{lv}"] = cur

    # bottleneck fold (unet1d.py:1144-1148): (b rt) d mz -> b (d mz) rt
    d, mzn = cur.shape[1], cur.shape[2]
    m = cur.reshape(B, RT, d * mzn).transpose(1, 2)
    m = resnet_block(p, "mid_block1", m, temb, 1)
    m = mid_attention(p, m, cond, use_rope)
    m = resnet_block(p, "mid_block2", m, temb, 1)
    if taps is not None:
        taps["mid"] = m
    cur = m.transpose(1, 2).reshape(B * RT, d, mzn)

    for ui in range(L):
        pre = f"ups.{ui}"
        cur = resnet_block(p, pre + ".0", torch.cat((cur, hs.pop()), dim=1), temb, RT)
        cur = resnet_block(p, pre + ".1", torch.cat((cur, hs.pop()), dim=1), temb, RT)
        cur = linear_attention(p, pre + ".2", cur)
        if ui < L - 1:  # nearest x2 then k3 p1 (unet1d.py:93-96)
            cur = F.interpolate(cur, scale_factor=2, mode="nearest")
            cur = F.conv1d(cur, p[pre + ".3.1.weight"], p[pre + ".3.1.bias"], padding=1)
        else:
            cur = F.conv1d(cur, p[pre + ".3.weight"], p[pre + ".3.bias"], padding=1)
        if taps is not None:
            taps[f"up{ui}"] = cur

    cur = resnet_block(p, "final_res_block", torch.cat((cur, r), dim=1), temb, RT)
    out = F.conv1d(cur, p["final_conv.weight"], p["final_conv.bias"])
    return out.reshape(B, RT, MZ)


# --------------------------------------------------------------------------------------
# diffusion process (model.py:244-406)
# --------------------------------------------------------------------------------------
@dataclass
class Diffusion:
    params: Params
    cfg: UNetConfig
    T: int = 1000
    kind: str = "cosine"
    use_rope: bool = True
    sched: Dict[str, torch.Tensor] = field(default_factory=dict)
    pred_type: str = "eps"

    def __post_init__(self):
        if self.pred_type not in ("eps", "x0"):
            raise ValueError(f"Unknown pred_type: {self.pred_type}")  # model.py:213
        self.sched = make_schedule(self.T, self.kind)

    @property
    def loss_weight(self):
        """model.py:205-211: ones for 'eps', SNR = ab / (1 - ab) for 'x0'."""
        ab = self.alpha_bars
        return torch.ones_like(ab) if self.pred_type == "eps" else ab / (1 - ab)

    @property
    def alpha_bars(self):
        return self.sched["alpha_bars"]

    def net(self, x_t, t, ms2_cond, ms1_cond):
        return unet_forward(self.params, self.cfg, x_t, t, ms2_cond, ms1_cond, self.use_rope)

    def p_sample(self, x_t, t: int, ms2_cond, ms1_cond):
        """model.py:244-291.  Conditions are already normalised.  Returns (x_prev, eps_pred) for both pred types."""
        tt = torch.full((x_t.shape[0],), t, dtype=torch.long)
        out = self.net(x_t, tt, ms2_cond, ms1_cond)
        if self.pred_type == "x0":
            return ddim_update_x0(self.alpha_bars, x_t, out, t)
        return ddim_update(self.alpha_bars, x_t, out, t), out

    def sample(self, x_T, ms2_cond, ms1_cond, num_steps: int, trace: Optional[list] = None):
        """model.py:293-324: returns (denoised in [0,1], mixture - denoised)."""
        c2, c1 = normalize(ms2_cond), normalize(ms1_cond)
        x = x_T
        for t in sampler_timesteps(self.T, num_steps):
            x, eps = self.p_sample(x, int(t), c2, c1)
            if trace is not None:
                trace.append((int(t), x.clone(), eps.clone()))
        x = unnormalize(x)
        return x, unnormalize(c2) - x

    def train_loss(self, x0, ms2_cond, ms1_cond, t, noise, ms1_loss_weight: float = 0.0):
        """model.py:344-361, 372-376, 404 with explicit (t, noise): mean over the batch of
        loss_weight[t_b] * loss_b, loss_b = MSE_b (target = noise for 'eps', normalised x0 for 'x0') when ms1_loss_weight = 0.
        At B = 1 this equals the reference's (1,)-shaped loss.  Returns (loss, network output).

        ms1_loss_weight > 0 (model.py:364-371, 379-386, 398-402): loss_b = (1 - w) * MSE_b + w * additional_b.  The reference's
        branch cannot run (``torch.max(x, dim=-1)`` returns a (values, indices) tuple that is then divided -> TypeError), so its
        semantics are CHOSEN here, as close to the text of the branch as a runnable program gets:
          * ``func(..., dim=-1)`` for func = torch.max means its ``.values``;
          * the MS1 chromatogram (B, RT) is the (B, RT, 1) tensor the network itself folds it into (unet1d.py:1122-1124), so
            ``func(ms1_cond, dim=-1)`` is the chromatogram for all three funcs;
          * ``x / torch.max(x)`` normalises by the maximum of that SAMPLE's chromatogram (batched semantics = the B = 1 reference
            per sample; at B = 1 this is the reference's global max);
          * sic = func(x_t - eps_pred) for 'eps' and func(x0_pred) for 'x0', exactly as written; additional_b = sum over
            the three funcs of MSE over RT."""
        x0n, c2, c1 = normalize(x0), normalize(ms2_cond), normalize(ms1_cond)
        x_t = q_sample(self.alpha_bars, x0n, t, noise)
        out = self.net(x_t, t, c2, c1)
        target = noise if self.pred_type == "eps" else x0n
        per = ((out - target) ** 2).flatten(1).mean(dim=1)
        if ms1_loss_weight > 0.0:
            d = (x_t - out) if self.pred_type == "eps" else out
            ms1 = c1 if c1.dim() == 2 else c1[..., 0]
            tgt = ms1 / ms1.max(dim=-1, keepdim=True).values
            add = torch.zeros_like(per)
            for sic in (d.sum(dim=-1), d.mean(dim=-1), d.max(dim=-1).values):
                add = add + ((sic / sic.max(dim=-1, keepdim=True).values - tgt) ** 2).mean(dim=-1)
            per = (1 - ms1_loss_weight) * per + ms1_loss_weight * add
        elif self.pred_type == "eps":
            return F.mse_loss(out, noise), out
        return (per * self.loss_weight[t]).mean(), out


# --------------------------------------------------------------------------------------
# batch formation: the step before the hot path
# --------------------------------------------------------------------------------------
def pair_batch(ms2_data, ms1_data, idx1, idx2, mixture_weights=(0.5, 0.5)):
    """DIAMSDataset.__getitem__ (data_loader.py:70-79, normalize='minmax') for explicit index pairs + the mixture of
    ModelInterface._train_one_epoch (model_interface.py:1073-1075).  numpy in the data's dtype, cast to float32 at the end
    like the reference (:83-88); the mixture is formed on the float32 tensors.  Returns float32 arrays
    (ms2_1, ms1_1, ms2_2, ms1_2, ms2_cond), each stacked over the pairs."""
    import numpy as np

    outs = [[], [], [], [], []]
    for i, j in zip(idx1, idx2):
        a, m1, b, m2 = ms2_data[i], ms1_data[i], ms2_data[j], ms1_data[j]
        lo, hi = np.min([a.min(), b.min()]), np.max([a.max(), b.max()])   # MS2: over both windows
        lo1, hi1 = np.min([m1.min()]), np.max([m1.max()])                 # MS1: window 1 only, applied to both
        with np.errstate(invalid="ignore", divide="ignore"):
            a, b = (a - lo) / (hi - lo), (b - lo) / (hi - lo)
            m1, m2 = (m1 - lo1) / (hi1 - lo1), (m2 - lo1) / (hi1 - lo1)
        a, m1, b, m2 = (np.asarray(v).astype(np.float32) for v in (a, m1, b, m2))
        c = (torch.from_numpy(a) * mixture_weights[0] + torch.from_numpy(b) * mixture_weights[1]).numpy()
        for o, v in zip(outs, (a, m1, b, m2, c)):
            o.append(v)
    return tuple(np.stack(o) for o in outs)


# --------------------------------------------------------------------------------------
# optimiser step (model_interface.py:1011, 1112-1123)
# --------------------------------------------------------------------------------------
def clip_coef(grads: Sequence[torch.Tensor], max_norm: float = 10.0) -> Tuple[float, float]:
    """torch clip_grad_norm_: L2 over all grads; coef = min(1, max_norm / (norm + 1e-6))."""
    tot = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads))
    return tot, min(1.0, max_norm / (tot + 1e-6))


def adamw_step(p, g, m, v, step: int, lr: float, b1=0.9, b2=0.999, eps=1e-8, wd=0.01):
    """torch.optim.AdamW defaults (decoupled decay), single-tensor form.  In place on p, m, v."""
    p.mul_(1 - lr * wd)
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1 = 1 - b1 ** step
    bc2 = 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


def trainable_keys(p: Params) -> List[str]:
    """Everything except the non-trainable RoPE frequencies (SURVEY 2.1 parameter-count note)."""
    return [k for k in p if not k.endswith("rotary_emb.freqs")]
