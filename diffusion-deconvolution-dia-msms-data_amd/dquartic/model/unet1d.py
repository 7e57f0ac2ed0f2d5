"""UNet1d noise predictor -- drop-in for the reference's ``dquartic.model.unet1d.UNet1d`` (simple=True path).

Same constructor surface (reference unet1d.py:918-939), same ``forward(x, time, init_cond, attn_cond)`` contract
(:1086-1166), same ``state_dict`` keys, shapes and registration order (so reference checkpoints load), same default
initialisation under the same torch seed.  The arithmetic runs in libdq_hip.so (hand-written gfx950 kernels); this
module only owns the parameters and hands raw device pointers to the C ABI (include/dq_hip.h).  There is no PyTorch
fallback: on a CPU tensor or without the library ``forward`` raises.

Differences from the reference, all documented in DESIGN.md:
  * batches work (the reference only runs at B = 1): sample b's time embedding is applied to its own rows;
  * only the working configuration family is built: simple=True, conditional=True, channels=1, init_cond_channels=1,
    attn_cond_channels=1, 4 heads x 32, dim = 4, dim*mult <= 16, downsample_dim divisible by 2**(len(dim_mults)-1) -- which
    includes the reference's shipped configuration (downsample_dim 40000: m/z rows of 40000 .. 625 positions and a
    10,000-channel bottleneck) -- anything else raises at construction;
  * all trainable tensors are views of ONE flat fp32 buffer (``flat_params``), gradients of one flat ``flat_grads``.
"""
import ctypes
import math
from typing import List, Optional

import torch
from torch import nn

from .. import _native as N

__all__ = ["UNet1d"]


class _Node(nn.Module):
    """Bare container; children/parameters are attached under the reference's attribute names."""


def _attach(root: nn.Module, dotted: str, param: nn.Parameter):
    *path, leaf = dotted.split(".")
    mod = root
    for name in path:
        nxt = mod._modules.get(name)
        if nxt is None:
            nxt = _Node()
            mod.add_module(name, nxt)
        mod = nxt
    mod.register_parameter(leaf, param)


class _FlatBuffers:
    """Mixin of the two networks: every trainable tensor is a view of ONE flat fp32 buffer, every ``.grad`` a view of one flat gradient
    buffer.  ``flat_params`` / ``flat_grads()`` are called several times per optimiser step, so the check "are the views still in place" is
    O(1): whatever re-creates parameter storage goes through ``nn.Module._apply`` (``.to()`` / ``.cuda()`` / ``.float()``), which raises
    ``_flat_stale`` here, and two sentinels (the first and the last tensor of the layout) catch a caller that re-pointed ``.data`` /
    ``.grad`` wholesale (a foreign optimiser's ``zero_grad(set_to_none=True)``).  Only then is the full walk over the ~400 tensors made
    (it cost 70 + 210 us of host time per call: ~0.7 ms per eager step, most of a batch-1 step)."""

    _flat_stale = True
    _grad_stale = True

    def _apply(self, fn, recurse=True):
        self._flat_stale = True
        self._grad_stale = True
        return super()._apply(fn, recurse)

    def zero_grad(self, set_to_none: bool = True):
        self._grad_stale = True
        return super().zero_grad(set_to_none=set_to_none)

    def _sentinels(self):
        return self._layout[0], self._layout[-1]

    def _flat_buffer(self) -> torch.Tensor:
        """Re-establish 'every parameter is a view of self._flat' after .to()/.cuda()/load_state_dict replaced storage."""
        base = self._flat.data_ptr()
        if not self._flat_stale and all(self._by_name[n].data_ptr() == base + 4 * o for n, o, _ in self._sentinels()):
            return self._flat
        dev = self._by_name[self._layout[0][0]].device
        ok = self._flat.device == dev and all(self._by_name[n].data_ptr() == base + 4 * o for n, o, _ in self._layout)
        if not ok:
            flat = torch.zeros(self._flat.numel(), dtype=torch.float32, device=dev)  # (zeros: a wide bottleneck's layout has alignment gaps)
            for pname, o, shape in self._layout:
                p = self._by_name[pname]
                flat[o:o + p.numel()].copy_(p.detach().reshape(-1).to(torch.float32))
                p.data = flat[o:o + p.numel()].view(shape)
            self._flat = flat
            self._flat_grad = None
            self._grad_stale = True
        self._flat_stale = False
        return self._flat

    def flat_grads(self, zero: bool = False) -> torch.Tensor:
        """The flat gradient buffer; every parameter's ``.grad`` is a view of it."""
        flat = self._flat_buffer()
        if self._flat_grad is None or self._flat_grad.device != flat.device:
            self._flat_grad = torch.zeros_like(flat)
            self._grad_stale = True
            zero = False
        if zero:
            self._flat_grad.zero_()
        base = self._flat_grad.data_ptr()
        if not self._grad_stale:
            for n, o, _ in self._sentinels():
                g = self._by_name[n].grad
                if g is None or g.data_ptr() != base + 4 * o:
                    self._grad_stale = True
        if self._grad_stale:
            for pname, o, shape in self._layout:
                p = self._by_name[pname]
                if p.grad is None or p.grad.data_ptr() != base + 4 * o:
                    p.grad = self._flat_grad[o:o + p.numel()].view(shape)
            self._grad_stale = False
        return self._flat_grad


class UNet1d(_FlatBuffers, nn.Module):
    def __init__(
        self,
        dim,
        init_dim=None,
        out_dim=None,
        dim_mults=(1, 2, 4, 8),
        channels=3,
        dropout=0.0,
        conditional=True,
        init_cond_channels=None,
        attn_cond_channels=None,
        attn_cond_init_dim=None,
        learned_variance=False,
        sinusoidal_pos_emb_theta=10000,
        attn_heads=4,
        attn_dim_head=32,
        tfer_dim_mult=620,
        tfer_depth=4,
        downsample_dim=40000,
        simple=True,
        pos_output_only=False,
    ):
        super().__init__()
        unsupported = []
        if not simple:
            unsupported.append("simple=False (crashes in the reference too, SURVEY F3)")
        if not conditional:
            unsupported.append("conditional=False")
        if channels != 1 or init_cond_channels != 1 or attn_cond_channels != 1:
            unsupported.append("channels/init_cond_channels/attn_cond_channels other than 1")
        if init_dim not in (None, dim) or out_dim not in (None, 1) or attn_cond_init_dim not in (None, 2 * dim):
            unsupported.append("non-default init_dim/out_dim/attn_cond_init_dim")
        if learned_variance or pos_output_only or dropout != 0.0:
            unsupported.append("learned_variance / pos_output_only / dropout")
        if attn_heads != 4 or attn_dim_head != 32 or sinusoidal_pos_emb_theta != 10000:
            unsupported.append("attn_heads/attn_dim_head/theta other than 4/32/10000")
        if unsupported:
            raise NotImplementedError("UNet1d (MI355X build) does not implement: " + "; ".join(unsupported))

        self.channels = channels
        self.conditional = conditional
        self.out_dim = 1
        self.dim = int(dim)
        self.dim_mults = tuple(int(m) for m in dim_mults)
        self.downsample_dim = int(downsample_dim)
        self.downsampled_n = self.downsample_dim // (2 ** (len(self.dim_mults) - 1))
        self.num_timesteps_hint = 1000

        lib = N.lib()
        mults = (ctypes.c_int * len(self.dim_mults))(*self.dim_mults)
        self._plan = lib.dq_plan_create(self.dim, len(self.dim_mults), mults, self.downsample_dim, 1000)
        if not self._plan:
            raise ValueError("UNet1d: " + (lib.dq_last_error() or b"?").decode())

        # ---- parameters: views of one flat buffer, attached under the reference's state_dict names
        self._layout = []  # (name, offset, shape)
        total = lib.dq_plan_param_floats(self._plan)
        name = ctypes.create_string_buffer(256)
        off, nd, shp = ctypes.c_int64(), ctypes.c_int(), (ctypes.c_int64 * 4)()
        for i in range(lib.dq_plan_num_params(self._plan)):
            N.check(lib.dq_plan_param_info(self._plan, i, name, 256, ctypes.byref(off), ctypes.byref(nd), shp), "dq_plan_param_info")
            self._layout.append((name.value.decode(), int(off.value), tuple(int(shp[k]) for k in range(nd.value))))
        self._flat = torch.zeros(total, dtype=torch.float32)
        self._flat_grad: Optional[torch.Tensor] = None
        self._by_name = {}
        for pname, o, shape in self._layout:
            p = nn.Parameter(self._flat[o:o + math.prod(shape)].view(shape))
            self._by_name[pname] = p
        # registration order == reference state_dict order; RoPE freqs sit where the reference has them
        for pname, _, _ in self._layout:
            if pname == "mid_attn.fn.fn.to_qv.weight":
                freqs = 1.0 / (10000 ** (torch.arange(0, 16, 2)[:8].float() / 16))
                _attach(self, "mid_attn.fn.fn.rotary_emb.freqs", nn.Parameter(freqs, requires_grad=False))
            _attach(self, pname, self._by_name[pname])
        self._reset_parameters()
        self._ws = {}
        self._ws_pool = {}  # training workspaces of the autograd bridge: one per forward that still awaits its backward
        self.use_rope = True

    # ------------------------------------------------------------------ initialisation
    def _init_order(self) -> List[str]:
        """Tensor names in the reference's CONSTRUCTION order (unet1d.py:949-1082: downs, then the mid blocks, then
        ups), which is the order the default initialisers consume the RNG in."""
        names = [n for n, _, _ in self._layout]
        head = [n for n in names if n.split(".")[0] in ("init_conv", "time_mlp", "init_cond_proj", "attn_cond_proj")]
        downs = [n for n in names if n.startswith("downs.")]
        mid = [n for n in names if n.startswith("mid_")]
        ups = [n for n in names if n.startswith("ups.")]
        tail = [n for n in names if n.startswith("final_")]
        assert len(head) + len(downs) + len(mid) + len(ups) + len(tail) == len(names)
        return head + downs + mid + ups + tail

    @torch.no_grad()
    def _reset_parameters(self):
        """nn.Conv1d / nn.Linear defaults (kaiming_uniform(a=sqrt(5)) weight, U(+-1/sqrt(fan_in)) bias); norm gains = 1."""
        fan_in = {}
        for pname in self._init_order():
            p = self._by_name[pname]
            if pname.endswith(".g"):
                p.fill_(1.0)
            elif pname.endswith(".weight"):
                nn.init.kaiming_uniform_(p, a=math.sqrt(5))
                fan_in[pname[: -len(".weight")]] = p[0].numel()
            elif pname.endswith(".bias"):
                bound = 1.0 / math.sqrt(fan_in[pname[: -len(".bias")]])
                nn.init.uniform_(p, -bound, bound)
            else:
                raise AssertionError(pname)

    # ------------------------------------------------------------------ flat buffers
    def trainable_named(self):
        return [(n, self._by_name[n]) for n, _, _ in self._layout]

    def _ensure_flat(self):
        return self._flat_buffer()

    @property
    def flat_params(self) -> torch.Tensor:
        return self._flat_buffer()

    def rope_freqs(self) -> Optional[torch.Tensor]:
        if not self.use_rope:
            return None
        return self.mid_attn.fn.fn.rotary_emb.freqs

    def workspace(self, B: int, RT: int, training: bool) -> torch.Tensor:
        dev = self._flat.device
        key = (B, RT, bool(training), str(dev))
        ws = self._ws.get(key)
        if ws is None:
            nbytes = N.lib().dq_unet_workspace_bytes(self._plan, B, RT, 1 if training else 0)
            if nbytes < 0:
                raise RuntimeError("dq_unet_workspace_bytes failed")
            self._ws = {k: v for k, v in self._ws.items() if k[2] != bool(training)}  # keep one per mode
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._ws[key] = ws
        return ws

    def checkout_train_workspace(self, B: int, RT: int) -> torch.Tensor:
        """A training workspace (activation arena + gradient twin) owned by ONE forward of the autograd bridge until its
        backward hands it back: two forwards with grad enabled before a backward (micro-batches whose losses are summed, a
        consistency term) each keep their own saved activations instead of overwriting a shared buffer."""
        dev = self._flat.device
        key = (B, RT, str(dev))
        pool = self._ws_pool.get(key)
        if pool:
            return pool.pop()
        nbytes = N.lib().dq_unet_workspace_bytes(self._plan, B, RT, 1)
        if nbytes < 0:
            raise RuntimeError("dq_unet_workspace_bytes failed")
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)

    def return_train_workspace(self, B: int, RT: int, ws: torch.Tensor):
        key = (B, RT, str(ws.device))
        self._ws_pool = {k: v for k, v in self._ws_pool.items() if k == key}  # keep buffers of the current shape only
        pool = self._ws_pool.setdefault(key, [])
        if len(pool) < 2:
            pool.append(ws)

    def __del__(self):
        try:
            if getattr(self, "_plan", None):
                N.lib().dq_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    # ------------------------------------------------------------------ forward
    def _prep(self, x, time, init_cond, attn_cond):
        if not x.is_cuda:
            raise RuntimeError("UNet1d (MI355X build): tensors must live on the GPU; there is no CPU fallback")
        if x.dim() == 2:  # reference accepts (rt, mz) (unet1d.py:1099-1104)
            x = x[None]
        B, RT, MZ = x.shape
        if MZ != self.downsample_dim:
            raise ValueError(f"UNet1d: m/z length {MZ} must equal downsample_dim {self.downsample_dim}")
        if init_cond is None:  # unet1d.py:1108
            init_cond = torch.zeros_like(x)
        if init_cond.dim() == 2:
            init_cond = init_cond[None]
        if attn_cond is None:
            raise ValueError("UNet1d: attn_cond (MS1) is required when conditional=True (the reference's None path is inconsistent)")
        if attn_cond.dim() == 3:
            if attn_cond.shape[-1] != 1:
                raise ValueError("UNet1d: 3-D attn_cond needs a trailing dimension of attn_cond_channels=1")
            attn_cond = attn_cond[..., 0]
        if tuple(init_cond.shape) != (B, RT, MZ) or tuple(attn_cond.shape) != (B, RT):
            raise ValueError("UNet1d: init_cond must be (B,RT,MZ) and attn_cond (B,RT)")
        time = time.reshape(-1).to(device=x.device, dtype=torch.int64)
        if time.numel() == 1 and B > 1:
            time = time.expand(B)
        if time.numel() != B:
            raise ValueError("UNet1d: time must have one entry per sample")
        f32 = lambda t: t.detach().to(torch.float32).contiguous()
        return f32(x), time.contiguous(), f32(init_cond), f32(attn_cond), B, RT, MZ

    def forward(self, x, time, init_cond=None, attn_cond=None):
        squeeze = x.dim() == 2
        xs, ts, ic, ac, B, RT, MZ = self._prep(x, time, init_cond, attn_cond)
        self._ensure_flat()
        need_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for _, p in self.trainable_named()))
        if need_grad:
            x_in = x[None] if squeeze else x  # keep x's autograd history so that d loss / d x reaches the caller's tensor
            x_in = x_in.to(torch.float32).contiguous() if x.requires_grad else xs
            out = _UNetFn.apply(self, x_in, ts, ic, ac, *[p for _, p in self.trainable_named()])
        else:
            out = self._run_fwd(xs, ts, ic, ac, training=False)
        return out[0] if squeeze else out

    def _run_fwd(self, xs, ts, ic, ac, training, cond_mul=1.0, cond_add=0.0, ws=None):
        B, RT, MZ = xs.shape
        if ws is None:
            ws = self.workspace(B, RT, training)
        out = torch.empty_like(xs)
        fr = self.rope_freqs()
        N.check(N.lib().dq_unet_fwd(self._plan, N.ptr(self._flat), N.ptr(fr), N.ptr(xs), N.ptr(ts), 0, N.ptr(ic), N.ptr(ac),
                                    cond_mul, cond_add, N.ptr(out), 1 if training else 0, N.ptr(ws), ws.numel(), B, RT,
                                    N.stream_ptr()), "dq_unet_fwd")
        return out


class _UNetFn(torch.autograd.Function):
    """Generic autograd bridge (loss.backward() through the network).  The fused training path
    (DDIMDiffusionModel._train_one_batch -> dq_train_step) does not go through here."""

    @staticmethod
    def forward(ctx, net, xs, ts, ic, ac, *params):
        ctx.net = net
        ctx.save_for_backward(ic)
        ctx.shape = xs.shape
        ctx.x_needs = xs.requires_grad
        ctx.ws = net.checkout_train_workspace(xs.shape[0], xs.shape[1])  # this forward's saved activations live here
        return net._run_fwd(xs.detach(), ts, ic, ac, training=True, ws=ctx.ws)

    @staticmethod
    def backward(ctx, gout):
        net = ctx.net
        (ic,) = ctx.saved_tensors
        B, RT, MZ = ctx.shape
        ws = ctx.ws
        if ws is None:
            raise RuntimeError("UNet1d: backward through the same forward twice is not supported (its workspace was released)")
        gout = gout.contiguous().to(torch.float32)
        grads = torch.zeros_like(net._flat)
        gx = torch.empty(ctx.shape, dtype=torch.float32, device=gout.device) if ctx.x_needs else None
        fr = net.rope_freqs()
        N.check(N.lib().dq_unet_bwd(net._plan, N.ptr(net._flat), N.ptr(fr), N.ptr(ic), 1.0, 0.0, N.ptr(gout), N.ptr(grads), N.ptr(gx),
                                    N.ptr(ws), ws.numel(), B, RT, N.stream_ptr()), "dq_unet_bwd")
        ctx.ws = None
        net.return_train_workspace(B, RT, ws)
        pg = [grads[o:o + math.prod(shape)].view(shape) for _, o, shape in net._layout]
        return (None, gx, None, None, None, *pg)
