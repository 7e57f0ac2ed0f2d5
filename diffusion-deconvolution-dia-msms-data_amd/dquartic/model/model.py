"""DDIM diffusion process -- drop-in for the reference's ``dquartic.model.model`` (reference model.py:14-406).

Same module-level helpers and the same ``DDIMDiffusionModel`` constructor, attributes and methods
(``q_sample``, ``p_sample``, ``sample``, ``train_step``).  When the wrapped network is this package's ``UNet1d`` the
arithmetic runs in libdq_hip.so: ``sample`` is one native call that walks all timesteps (network forward + DDIM update per
step, no host sync inside), ``train_step`` is one native call (normalise, q_sample, forward, MSE, backward into the flat
gradient buffer).  Any other ``nn.Module`` is driven with plain tensor ops exactly like the reference does, so the class
stays usable as a generic harness -- but that is the caller's network, not this package's hot path.

Documented deviations: batches are supported and ``train_step`` returns a 0-dim loss (the mean over
samples of the reference's B = 1 loss); both ``pred_type`` values are built; ``ms1_loss_weight > 0`` is built with chosen
semantics (the reference's branch raises TypeError).
"""
import ctypes
import math
import os

import torch
import torch.nn.functional as F

from .. import _native as N
from .model_interface import BucketedAllReduce, ModelInterface
from .unet1d import UNet1d


# beta schedules (reference model.py:14-54): fp64, cast by the caller
def get_linear_beta_schedule(num_timesteps, beta_start=0.0001, beta_end=0.02):
    return torch.linspace(beta_start, beta_end, num_timesteps, dtype=torch.float64)


def get_cosine_beta_schedule(num_timesteps, s=0.008):
    x = torch.linspace(0, num_timesteps, num_timesteps + 1, dtype=torch.float64)
    ac = torch.cos(((x / num_timesteps) + s) / (1 + s) * math.pi * 0.5) ** 2
    ac = ac / ac[0]
    return torch.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)


def get_alphas(betas):  # model.py:57-69
    return 1.0 - betas


def get_alpha_bars(alpha):  # model.py:72-84
    return torch.cumprod(alpha, dim=0)


def normalize_to_neg_one_to_one(img):  # model.py:89-99
    return img * 2 - 1


def unnormalize_to_zero_to_one(t):  # model.py:102-112
    return (t + 1) * 0.5


def identity(t, *args, **kwargs):  # model.py:115-125
    return t


def extract(a, t, x_shape):  # model.py:131-148
    b, *_ = t.shape
    out = a.gather(-1, t)
    return out.reshape(b, *((1,) * (len(x_shape) - 1)))


class DDIMDiffusionModel(ModelInterface):
    def __init__(self, model_class, num_timesteps=1000, beta_schedule_type="cosine", pred_type="eps", auto_normalize=True,
                 ms1_loss_weight=0.0, device="cuda", **kwargs):
        super().__init__()
        self.model = None
        self.build(model_class, **kwargs)
        self.num_timesteps = num_timesteps
        self.device = device
        # schedule exactly as the reference forms it (model.py:196-202): fp64 betas -> device -> fp32, cumprod in fp32
        # The three tensors are formed on the HOST and then moved: a device cumprod (parallel scan) rounds differently
        # from the sequential CPU one, and parity is defined against the reference's CPU path.
        betas = (get_linear_beta_schedule(num_timesteps) if beta_schedule_type == "linear"
                 else get_cosine_beta_schedule(num_timesteps)).to(torch.float32)
        alphas = get_alphas(betas).to(torch.float32)
        alpha_bars = get_alpha_bars(alphas).to(torch.float32)
        self.betas, self.alphas, self.alpha_bars = betas.to(device), alphas.to(device), alpha_bars.to(device)
        snr = (alpha_bars / (1 - alpha_bars)).to(device)  # model.py:205
        if pred_type == "eps":
            self.loss_weight = torch.ones_like(snr)
        elif pred_type == "x0":
            self.loss_weight = snr
        else:
            raise ValueError(f"Unknown pred_type: {pred_type}")
        self.normalize = normalize_to_neg_one_to_one if auto_normalize else identity
        self.unnormalize = unnormalize_to_zero_to_one if auto_normalize else identity
        self.auto_normalize = bool(auto_normalize)
        self.pred_type = pred_type
        self.ms1_loss_weight = ms1_loss_weight
        self._ab_host = None
        self.use_graph = True  # sample(): replay one hipGraph-captured step per timestep (an attribute, not an environment switch)

    # ------------------------------------------------------------------ helpers
    @property
    def native(self) -> bool:
        return isinstance(self.model, UNet1d)

    def _alpha_bars_host(self):
        if self._ab_host is None:
            ab = self.alpha_bars.detach().to("cpu", torch.float32).contiguous()
            self._ab_host = (ab, (ctypes.c_float * ab.numel()).from_buffer_copy(ab.numpy().tobytes()))
        return self._ab_host[1]

    @staticmethod
    def sampler_timesteps(num_timesteps, num_steps):
        """model.py:313"""
        return torch.linspace(num_timesteps - 1, 0, num_steps, dtype=torch.long)

    # ------------------------------------------------------------------ forward process
    def q_sample(self, x_0, t, noise=None):
        """model.py:225-242; ``x_0`` is already normalised by the caller, as in the reference."""
        if noise is None:
            noise = torch.randn_like(x_0)
        # the native kernel when nothing upstream wants a gradient through the noising (the training paths of this package
        # draw x_0 / noise as leaves); otherwise the reference's tensor expressions, which keep the autograd history
        wants_grad = torch.is_grad_enabled() and (x_0.requires_grad or noise.requires_grad)
        if x_0.is_cuda and not wants_grad:
            x0c, nz = x_0.detach().float().contiguous(), noise.detach().float().contiguous()
            tt = t.reshape(-1).to(device=x_0.device, dtype=torch.int64).contiguous()
            ab = self.alpha_bars.to(x_0.device)
            out = torch.empty_like(x0c)
            B = x0c.shape[0]
            N.check(N.lib().dq_q_sample(N.ptr(ab), N.ptr(x0c), N.ptr(tt), N.ptr(nz), N.ptr(out), B, x0c[0].numel(), 0,
                                        N.stream_ptr()), "dq_q_sample")
            return out
        a = torch.sqrt(self.alpha_bars[t])[:, None, None]
        b = torch.sqrt(1.0 - self.alpha_bars[t])[:, None, None]
        return a * x_0 + b * noise

    # ------------------------------------------------------------------ reverse process
    def p_sample(self, x_t, t, init_cond=None, attn_cond=None):
        """model.py:244-291.  ``t`` is a python int; conditions are already normalised."""
        batch_size = x_t.size(0)
        t_tensor = torch.full((batch_size,), int(t), device=x_t.device, dtype=torch.long)
        ab = self.alpha_bars[t]
        sa, sb = torch.sqrt(ab), torch.sqrt(1.0 - ab)
        if self.pred_type not in N.PRED_TYPES:
            raise ValueError(f"Unknown pred_type: {self.pred_type}")
        out = self.model(x_t, t_tensor, init_cond, attn_cond)  # eps_pred or x0_pred (model.py:271 / :276)
        wants_grad = torch.is_grad_enabled() and (x_t.requires_grad or out.requires_grad)
        if x_t.is_cuda and not wants_grad:
            if t > 0:
                abp = self.alpha_bars[t - 1]
                coef = torch.stack([sa, sb, torch.sqrt(abp), torch.sqrt(1.0 - abp)]).to(x_t.device, torch.float32)
            else:
                coef = torch.stack([sa, sb, -torch.ones_like(sa), torch.zeros_like(sa)]).to(x_t.device, torch.float32)
            xt, o = x_t.detach().float().contiguous(), out.detach().float().contiguous()
            x_prev = torch.empty_like(xt)
            if self.pred_type == "eps":
                N.check(N.lib().dq_ddim_step(N.ptr(xt), N.ptr(o), N.ptr(x_prev), N.ptr(coef), xt.numel(), N.stream_ptr()), "dq_ddim_step")
                return x_prev, out
            eps_pred = torch.empty_like(xt)
            N.check(N.lib().dq_ddim_step_x0(N.ptr(xt), N.ptr(o), N.ptr(x_prev), N.ptr(eps_pred), N.ptr(coef), xt.numel(),
                                            N.stream_ptr()), "dq_ddim_step_x0")
            return x_prev, eps_pred
        # host tensors (only reachable with a non-native network) or a caller differentiating through the step: the reference's
        # arithmetic as is
        if self.pred_type == "eps":
            eps_pred, x0_pred = out, (x_t - sb * out) / sa
        else:
            x0_pred, eps_pred = out, (x_t - sa * out) / sb
        if t > 0:
            abp = self.alpha_bars[t - 1]
            x_prev = torch.sqrt(abp) * x0_pred + torch.sqrt(1.0 - abp) * eps_pred
        else:
            x_prev = x0_pred
        return x_prev, eps_pred

    def sample(self, x_t, ms2_cond=None, ms1_cond=None, num_steps=1000, return_trajectory=False):
        """model.py:293-324: returns (denoised, mixture - denoised).  Native loop when the network is UNet1d."""
        if self.native and x_t.is_cuda and ms2_cond is not None and ms1_cond is not None:
            return self._sample_native(x_t, ms2_cond, ms1_cond, num_steps, return_trajectory)
        ms2n = self.normalize(ms2_cond) if ms2_cond is not None else None
        ms1n = self.normalize(ms1_cond) if ms1_cond is not None else None
        pred_noise = None
        for t in self.sampler_timesteps(self.num_timesteps, num_steps):
            x_t, pred_noise = self.p_sample(x_t, int(t.item()), ms2n, ms1n)
        x_t, pred_noise = self.unnormalize(x_t), self.unnormalize(pred_noise)
        if ms2n is not None:
            pred_noise = self.unnormalize(ms2n) - x_t
        return x_t, pred_noise

    def _sample_native(self, x_T, ms2_cond, ms1_cond, num_steps, return_trajectory=False):
        net: UNet1d = self.model
        f32 = lambda v: v.detach().to(torch.float32).contiguous()
        x_T, c2, c1 = f32(x_T), f32(ms2_cond), f32(ms1_cond)
        if c1.dim() == 3:
            c1 = c1[..., 0].contiguous()
        B, RT, MZ = x_T.shape
        flat = net.flat_params
        ws = net.workspace(B, RT, False)
        ts = self.sampler_timesteps(self.num_timesteps, num_steps).to(torch.int32)
        ts_c = (ctypes.c_int32 * num_steps)(*ts.tolist())
        out_x, out_n = torch.empty_like(x_T), torch.empty_like(x_T)
        traj_x = torch.empty((num_steps, B, RT, MZ), device=x_T.device) if return_trajectory else None
        traj_e = torch.empty((num_steps, B, RT, MZ), device=x_T.device) if return_trajectory else None
        N.check(N.lib().dq_ddim_sample(net._plan, N.ptr(flat), N.ptr(net.rope_freqs()), self._alpha_bars_host(), int(self.num_timesteps),
                                       N.ptr(x_T), N.ptr(c2),
                                       N.ptr(c1), 1 if self.auto_normalize else 0, N.PRED_TYPES[self.pred_type], ts_c, num_steps,
                                       N.ptr(out_x), N.ptr(out_n),
                                       N.ptr(traj_x), N.ptr(traj_e), 1 if (self.use_graph and not return_trajectory) else 0, N.ptr(ws),
                                       ws.numel(), B, RT, N.stream_ptr()), "dq_ddim_sample")
        if return_trajectory:
            return out_x, out_n, traj_x, traj_e
        return out_x, out_n

    # ------------------------------------------------------------------ training objective
    def train_step(self, x_0, ms2_cond=None, ms1_cond=None, noise=None, ms1_loss_weight=0.0, t=None):
        """model.py:326-406 (both pred types; ``ms1_loss_weight > 0`` with the semantics of DESIGN.md section 12: the reference's
        branch raises).  Draw order as in the reference: ``randint`` then
        ``randn_like``.  A passed ``noise`` is mapped 2*noise-1 like the reference does (model.py:346).  Returns a 0-dim loss
        (mean over samples of loss_weight[t_b] * MSE_b) that carries autograd history through the native network (generic
        path; the fused path is ``train_step_fused``)."""
        if self.pred_type not in N.PRED_TYPES:
            raise ValueError(f"Unknown pred_type: {self.pred_type}")
        batch_size = x_0.size(0)
        if t is None:
            t = torch.randint(0, self.num_timesteps, (batch_size,), device=x_0.device).long()
        noise = torch.randn_like(x_0) if noise is None else self.normalize(noise)
        x_0 = self.normalize(x_0)
        ms2n = self.normalize(ms2_cond) if ms2_cond is not None else None
        ms1n = self.normalize(ms1_cond) if ms1_cond is not None else None
        x_t = self.q_sample(x_0, t, noise=noise)
        out = self.model(x_t, t, ms2n, ms1n)
        w = float(ms1_loss_weight or 0.0)
        if self.pred_type == "eps" and w <= 0.0:
            return F.mse_loss(out, noise) * 1.0  # loss_weight is all-ones for the eps objective (model.py:208-209, 404)
        target = noise if self.pred_type == "eps" else x_0
        per_sample = ((out - target) ** 2).flatten(1).mean(dim=1)  # model.py:361 / :376 at B = 1, per sample here
        if w > 0.0:  # model.py:364-371, 379-386, 398-402 with the chosen semantics (this generic path: plain tensor expressions)
            d = (x_t - out) if self.pred_type == "eps" else out
            m1 = ms1n if ms1n.dim() == 2 else ms1n[..., 0]
            tgt = m1 / m1.max(dim=-1, keepdim=True).values
            add = torch.zeros_like(per_sample)
            for sic in (d.sum(dim=-1), d.mean(dim=-1), d.max(dim=-1).values):
                add = add + ((sic / sic.max(dim=-1, keepdim=True).values - tgt) ** 2).mean(dim=-1)
            per_sample = (1 - w) * per_sample + w * add
        return (per_sample * self.loss_weight.to(per_sample.device)[t]).mean()  # model.py:404

    def train_step_fused(self, x_0, ms2_cond, ms1_cond, t=None, noise=None, zero_grads=True, ms1_loss_weight=0.0):
        """One native call: normalise, q_sample, U-Net forward, MSE, backward into ``model.flat_grads()`` (+=).
        Returns the loss as a 0-dim device tensor (no host sync)."""
        from .building_blocks import DDIMTransformerAdapter

        if isinstance(self.model, DDIMTransformerAdapter):
            return self._train_step_fused_tfm(x_0, ms1_cond, t, noise, zero_grads, float(ms1_loss_weight or 0.0))
        net: UNet1d = self.model
        if not self.native:
            raise RuntimeError("train_step_fused needs this package's UNet1d or a DDIMTransformerAdapter")
        f32 = lambda v: v.detach().to(torch.float32).contiguous()
        x_0, c2, c1 = f32(x_0), f32(ms2_cond), f32(ms1_cond)
        if c1.dim() == 3:
            c1 = c1[..., 0].contiguous()
        B, RT, MZ = x_0.shape
        if t is None:
            t = torch.randint(0, self.num_timesteps, (B,), device=x_0.device).long()
        if noise is None:
            noise = torch.randn_like(x_0)
        t = t.to(device=x_0.device, dtype=torch.int64).contiguous()
        noise = f32(noise)
        flat = net.flat_params
        grads = net.flat_grads(zero=zero_grads)
        ws = net.workspace(B, RT, True)
        loss = torch.empty((), dtype=torch.float32, device=x_0.device)
        ab = self.alpha_bars.to(x_0.device)
        lw = self.loss_weight.to(device=x_0.device, dtype=torch.float32).contiguous()
        N.check(N.lib().dq_train_step(net._plan, N.ptr(flat), N.ptr(net.rope_freqs()), N.ptr(ab), N.ptr(x_0), N.ptr(c2), N.ptr(c1),
                                      N.ptr(t), N.ptr(noise), 1 if self.auto_normalize else 0, N.PRED_TYPES[self.pred_type], N.ptr(lw),
                                      float(ms1_loss_weight or 0.0), N.ptr(grads), N.ptr(loss), N.ptr(ws), ws.numel(), B, RT,
                                      N.stream_ptr()), "dq_train_step")
        return loss

    def _train_step_fused_tfm(self, x_0, ms1_cond, t=None, noise=None, zero_grads=True, ms1_loss_weight=0.0):
        """train_step for the CustomTransformer behind its adapter: the same sequence as the U-Net's dq_train_step (normalise +
        q_sample, network forward, MSE and its gradient, network backward into the flat gradient buffer), each stage one native
        call -- dq_q_sample, dq_tfm_fwd, dq_mse_loss(_weighted)_fwd_bwd, dq_tfm_bwd -- issued back to back on the current stream."""
        tfm = self.model.transformer
        f32 = lambda v: v.detach().to(torch.float32).contiguous()
        x_0, c1 = f32(x_0), f32(ms1_cond)
        if c1.dim() == 3:
            c1 = c1[..., 0].contiguous()
        B = x_0.shape[0]
        per = x_0[0].numel()
        dev = x_0.device
        if t is None:
            t = torch.randint(0, self.num_timesteps, (B,), device=dev).long()
        if noise is None:
            noise = torch.randn_like(x_0)
        t = t.to(device=dev, dtype=torch.int64).contiguous()
        noise = f32(noise)
        norm = 1 if self.auto_normalize else 0
        lib = N.lib()
        x_t = torch.empty_like(x_0)
        N.check(lib.dq_q_sample(N.ptr(self.alpha_bars.to(dev)), N.ptr(x_0), N.ptr(t), N.ptr(noise), N.ptr(x_t), B, per, norm, N.stream_ptr()),
                "dq_q_sample")
        c1n = self.normalize(c1).contiguous()  # (B, RT) values: the only torch op of the step
        tfm._ensure_flat()
        out = tfm._run_fwd(x_t, t, c1n, training=True)
        grads = tfm.flat_grads(zero=False)  # zero_grads: the backward below overwrites instead of accumulating
        loss = torch.empty((), dtype=torch.float32, device=dev)
        dout = torch.empty_like(out)
        if getattr(self, "_mse_scratch", None) is None or self._mse_scratch.device != dev:
            self._mse_scratch = torch.empty(4096, dtype=torch.float32, device=dev)
        if self.pred_type == "eps":
            N.check(lib.dq_mse_loss_fwd_bwd(N.ptr(out), N.ptr(noise), N.ptr(loss), N.ptr(dout), N.ptr(self._mse_scratch), out.numel(),
                                            N.stream_ptr()), "dq_mse_loss_fwd_bwd")
        else:
            lw = self.loss_weight.to(device=dev, dtype=torch.float32).contiguous()
            N.check(lib.dq_mse_loss_weighted_fwd_bwd(N.ptr(out), N.ptr(x_0), 2.0 if norm else 1.0, -1.0 if norm else 0.0, N.ptr(lw), N.ptr(t),
                                                     N.ptr(loss), N.ptr(dout), N.ptr(self._mse_scratch), B, per, N.stream_ptr()),
                    "dq_mse_loss_weighted_fwd_bwd")
        if ms1_loss_weight > 0.0:  # the MS1 term on top of the MSE part (same kernels as the U-Net's train step)
            B_, RT_, MZ_ = x_0.shape
            sc = torch.empty(5 * B_ * RT_ + B_ + 64, dtype=torch.float32, device=dev)
            lw_x0 = self.loss_weight.to(device=dev, dtype=torch.float32).contiguous() if self.pred_type == "x0" else None  # (kept alive past the call)
            lwp = N.ptr(lw_x0)
            N.check(lib.dq_ms1_loss_fwd_bwd(N.ptr(out), N.ptr(x_t) if self.pred_type == "eps" else None, N.ptr(c1), 2.0 if norm else 1.0,
                                            -1.0 if norm else 0.0, lwp, N.ptr(t), float(ms1_loss_weight), N.ptr(loss), N.ptr(dout), N.ptr(sc),
                                            B_, RT_, MZ_, N.stream_ptr()), "dq_ms1_loss_fwd_bwd")
        if zero_grads and torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            # data parallel: each layer's gradient slice is all-reduced while the layers below it are still in backward
            red = BucketedAllReduce(grads)
            tfm._run_bwd(x_t, c1n, dout, grads, False, False, accumulate=False, on_bucket=red.on_bucket)
            red.finish()
            self._grads_reduced = True  # tells _train_one_batch to skip its flat all-reduce
        else:
            tfm._run_bwd(x_t, c1n, dout, grads, False, False, accumulate=not zero_grads)
        return loss
