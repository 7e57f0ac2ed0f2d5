"""CustomTransformer noise predictor -- drop-in for the reference's ``dquartic.model.building_blocks.CustomTransformer``.

Same constructor (reference building_blocks.py:205), same ``forward(x_t, t, x_cond)`` contract (:224-260), same
``state_dict`` keys, shapes and registration order (reference checkpoints load), same default initialisation under the same
torch seed.  The arithmetic runs in libdq_hip.so: every dense layer and both attention products are the hand-written fp32
matrix-core GEMM (csrc/k_gemm.hip), the rest is csrc/k_tfm.hip; this module owns the parameters (views of ONE flat fp32
buffer) and hands raw device pointers to the C ABI (``dq_tfm_*`` in include/dq_hip.h).  No PyTorch fallback: on a CPU tensor or
without the library ``forward`` raises.

``DDIMTransformerAdapter`` gives it the 4-argument call ``DDIMDiffusionModel`` makes (reference model.py:271, 359) -- the
reference itself cannot drive this network through its DDIM class: ``model(x_t, t, ms2_cond, ms1_cond)`` ->
``transformer(x_t, t, ms1_cond)``.
"""
import ctypes
import math
from typing import Optional

import torch
from torch import nn

from .. import _native as N
from .unet1d import _attach, _FlatBuffers

_BUCKET_FN = ctypes.CFUNCTYPE(None, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int64)  # dq_tfm_bucket_fn

__all__ = ["CustomTransformer", "DDIMTransformerAdapter", "apply_rope_tables", "time_embedding_freqs"]


def apply_rope_tables(seqlen: int, hidden_dim: int):
    """sin / cos of apply_rope's angles (reference building_blocks.py:31-49), same torch expressions, fp32 on the host."""
    half = hidden_dim // 2
    freq_seq = torch.arange(half, dtype=torch.float32) / half
    inv_freq = 10000 ** (-freq_seq)
    angles = torch.einsum("i,j->ij", torch.arange(seqlen, dtype=torch.float32), inv_freq)
    return torch.sin(angles).contiguous(), torch.cos(angles).contiguous()


def time_embedding_freqs(hidden_dim: int):
    """TimeEmbedding's frequency vector (reference building_blocks.py:104-106)."""
    half = hidden_dim // 2
    return torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1))).contiguous()


class CustomTransformer(_FlatBuffers, nn.Module):
    def __init__(self, input_dim=40000, hidden_dim=128, num_heads=1, num_layers=1):
        super().__init__()
        self.input_dim, self.hidden_dim, self.num_heads, self.num_layers = int(input_dim), int(hidden_dim), int(num_heads), int(num_layers)
        lib = N.lib()
        self._tfm = lib.dq_tfm_create(self.input_dim, self.hidden_dim, self.num_heads, self.num_layers)
        if not self._tfm:
            raise NotImplementedError("CustomTransformer (MI355X build): " + (lib.dq_last_error() or b"?").decode())
        self._layout = []
        name = ctypes.create_string_buffer(256)
        off, nd, shp = ctypes.c_int64(), ctypes.c_int(), (ctypes.c_int64 * 2)()
        for i in range(lib.dq_tfm_num_params(self._tfm)):
            N.check(lib.dq_tfm_param_info(self._tfm, i, name, 256, ctypes.byref(off), ctypes.byref(nd), shp), "dq_tfm_param_info")
            self._layout.append((name.value.decode(), int(off.value), tuple(int(shp[k]) for k in range(nd.value))))
        self._flat = torch.zeros(lib.dq_tfm_param_floats(self._tfm), dtype=torch.float32)
        self._flat_grad: Optional[torch.Tensor] = None
        self._by_name = {}
        for pname, o, shape in self._layout:
            p = nn.Parameter(self._flat[o:o + math.prod(shape)].view(shape))
            self._by_name[pname] = p
            _attach(self, pname, p)
        self._reset_parameters()
        self.precision = "fp32"
        self._ws = {}
        self._ws_pool = {}  # training workspaces of the autograd bridge: one per forward that still awaits its backward
        self._tables = {}

    @torch.no_grad()
    def _reset_parameters(self):
        """The reference modules' defaults, consuming the RNG in the reference's construction order: nn.Linear
        (kaiming_uniform(a=sqrt(5)) weight, U(+-1/sqrt(fan_in)) bias); nn.MultiheadAttention builds out_proj (a Linear) first,
        then xavier_uniform on in_proj_weight and zero in_proj_bias / out_proj.bias; nn.LayerNorm ones / zeros."""
        def linear(prefix):
            w, b = self._by_name[prefix + ".weight"], self._by_name[prefix + ".bias"]
            nn.init.kaiming_uniform_(w, a=math.sqrt(5))
            bound = 1.0 / math.sqrt(w.shape[1])
            nn.init.uniform_(b, -bound, bound)

        for pre in ("input_projection", "output_projection", "conditional_projection", "time_embedding.linear1", "time_embedding.linear2"):
            linear(pre)
        for l in range(self.num_layers):
            pre = f"layers.{l}."
            linear(pre + "attention.out_proj")
            nn.init.xavier_uniform_(self._by_name[pre + "attention.in_proj_weight"])
            self._by_name[pre + "attention.in_proj_bias"].zero_()
            self._by_name[pre + "attention.out_proj.bias"].zero_()
            self._by_name[pre + "norm1.weight"].fill_(1.0)
            self._by_name[pre + "norm1.bias"].zero_()
            linear(pre + "ff.0")
            linear(pre + "ff.2")
            self._by_name[pre + "norm2.weight"].fill_(1.0)
            self._by_name[pre + "norm2.bias"].zero_()

    # ------------------------------------------------------------------ flat buffers (same scheme as UNet1d)
    def trainable_named(self):
        return [(n, self._by_name[n]) for n, _, _ in self._layout]

    def _ensure_flat(self):
        return self._flat_buffer()

    @property
    def flat_params(self) -> torch.Tensor:
        return self._flat_buffer()

    def workspace(self, B, S1, S2, training):
        dev = self._flat.device
        key = (B, S1, S2, bool(training), str(dev))
        ws = self._ws.get(key)
        if ws is None:
            nbytes = N.lib().dq_tfm_workspace_bytes(self._tfm, B, S1, S2, 1 if training else 0)
            if nbytes <= 0:
                raise RuntimeError("dq_tfm_workspace_bytes failed")
            self._ws = {k: v for k, v in self._ws.items() if k[3] != bool(training)}
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            self._ws[key] = ws
        return ws

    def set_precision(self, precision: str):
        """Arithmetic of the dense products: "fp32" (default; exact fp32 on the matrix cores -- the precision parity is stated in) or
        "bf16x3" (three bf16 matrix-core passes over hi/lo-split operands, fp32 accumulation: ~1e-5 relative per product term, a
        separate and faster mode with its own tolerance)."""
        if precision not in N.PRECISIONS:
            raise ValueError(f"CustomTransformer: precision must be one of {sorted(N.PRECISIONS)}")
        N.check(N.lib().dq_tfm_set_precision(self._tfm, N.PRECISIONS[precision]), "dq_tfm_set_precision")
        self.precision = precision
        return self

    def checkout_train_workspace(self, B, S1, S2):
        """A training workspace owned by ONE forward of the autograd bridge until its backward returns it (see UNet1d)."""
        dev = self._flat.device
        key = (B, S1, S2, str(dev))
        pool = self._ws_pool.get(key)
        if pool:
            return pool.pop()
        nbytes = N.lib().dq_tfm_workspace_bytes(self._tfm, B, S1, S2, 1)
        if nbytes <= 0:
            raise RuntimeError("dq_tfm_workspace_bytes failed")
        return torch.empty(nbytes, dtype=torch.uint8, device=dev)

    def return_train_workspace(self, B, S1, S2, ws):
        key = (B, S1, S2, str(ws.device))
        self._ws_pool = {k: v for k, v in self._ws_pool.items() if k == key}
        pool = self._ws_pool.setdefault(key, [])
        if len(pool) < 2:
            pool.append(ws)

    def tables(self, S, device):
        key = (S, str(device))
        t = self._tables.get(key)
        if t is None:
            sin, cos = apply_rope_tables(S, self.hidden_dim)
            t = (sin.to(device), cos.to(device), time_embedding_freqs(self.hidden_dim).to(device))
            self._tables = {key: t}
        return t

    def __del__(self):
        try:
            if getattr(self, "_tfm", None):
                N.lib().dq_tfm_destroy(self._tfm)
                self._tfm = None
        except Exception:
            pass

    # ------------------------------------------------------------------ forward
    def _prep(self, x_t, t, x_cond):
        if not x_t.is_cuda:
            raise RuntimeError("CustomTransformer (MI355X build): tensors must live on the GPU; there is no CPU fallback")
        if x_t.dim() != 3 or x_t.shape[-1] != self.input_dim:
            raise ValueError(f"CustomTransformer: x_t must be (batch, seqlen1, input_dim={self.input_dim})")
        if x_cond.dim() != 2 or x_cond.shape[0] != x_t.shape[0]:
            raise ValueError("CustomTransformer: x_cond must be (batch, seqlen2) -- one value per conditional position "
                             "(reference building_blocks.py:241-242 projects it with Linear(1, hidden))")
        t = t.reshape(-1).to(device=x_t.device, dtype=torch.int64)
        if t.numel() != x_t.shape[0]:
            raise ValueError("CustomTransformer: t must have one entry per sample")
        f32 = lambda v: v.detach().to(torch.float32).contiguous()
        return f32(x_t), t.contiguous(), f32(x_cond)

    def forward(self, x_t, t, x_cond):
        xs, ts, cs = self._prep(x_t, t, x_cond)
        self._ensure_flat()
        need_grad = torch.is_grad_enabled() and (x_t.requires_grad or x_cond.requires_grad or any(p.requires_grad for _, p in self.trainable_named()))
        if need_grad:
            x_in = x_t.to(torch.float32).contiguous() if x_t.requires_grad else xs
            c_in = x_cond.to(torch.float32).contiguous() if x_cond.requires_grad else cs
            return _TfmFn.apply(self, x_in, ts, c_in, *[p for _, p in self.trainable_named()])
        return self._run_fwd(xs, ts, cs, training=False)

    def _run_fwd(self, xs, ts, cs, training, ws=None):
        B, S1, _ = xs.shape
        S2 = cs.shape[1]
        if ws is None:
            ws = self.workspace(B, S1, S2, training)
        sin, cos, freqs = self.tables(max(S1, S2), xs.device)
        out = torch.empty_like(xs)
        N.check(N.lib().dq_tfm_fwd(self._tfm, N.ptr(self._flat), N.ptr(sin), N.ptr(cos), N.ptr(freqs), N.ptr(xs), N.ptr(ts), N.ptr(cs),
                                   N.ptr(out), 1 if training else 0, N.ptr(ws), ws.numel(), B, S1, S2, N.stream_ptr()), "dq_tfm_fwd")
        return out

    def _run_bwd(self, xs, cs, gout, grads, want_dx, want_dc, accumulate=True, ws=None, on_bucket=None):
        """``on_bucket(i, offset, count)``: called while the backward is being enqueued, once per gradient bucket, when the kernels
        writing ``grads[offset:offset+count]`` are all on the stream (dq_tfm_bwd_buckets; model_interface.BucketedAllReduce)."""
        B, S1, _ = xs.shape
        S2 = cs.shape[1]
        if ws is None:
            ws = self.workspace(B, S1, S2, True)
        sin, cos, _ = self.tables(max(S1, S2), xs.device)
        gx = torch.empty_like(xs) if want_dx else None
        gc = torch.empty_like(cs) if want_dc else None
        args = (self._tfm, N.ptr(self._flat), N.ptr(sin), N.ptr(cos), N.ptr(xs), N.ptr(cs), N.ptr(gout), N.ptr(grads),
                1 if accumulate else 0, N.ptr(gx), N.ptr(gc), N.ptr(ws), ws.numel(), B, S1, S2, N.stream_ptr())
        if on_bucket is None:
            N.check(N.lib().dq_tfm_bwd(*args), "dq_tfm_bwd")
        else:
            raised = []

            def hook(_user, i, off, cnt):  # an exception cannot cross the C frame: keep it and re-raise after the call
                try:
                    on_bucket(int(i), int(off), int(cnt))
                except BaseException as e:  # noqa: BLE001
                    raised.append(e)

            cb = _BUCKET_FN(hook)
            N.check(N.lib().dq_tfm_bwd_buckets(*args, ctypes.cast(cb, ctypes.c_void_p), None), "dq_tfm_bwd_buckets")
            if raised:
                raise raised[0]
        return gx, gc

    def grad_buckets(self):
        """[(offset, count)] of the flat gradient buffer in the order the backward completes them (dq_tfm_bucket_info)."""
        out = []
        for i in range(N.lib().dq_tfm_num_buckets(self._tfm)):
            off, cnt = ctypes.c_int64(), ctypes.c_int64()
            N.check(N.lib().dq_tfm_bucket_info(self._tfm, i, ctypes.byref(off), ctypes.byref(cnt)), "dq_tfm_bucket_info")
            out.append((off.value, cnt.value))
        return out


class _TfmFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, net, xs, ts, cs, *params):
        ctx.net = net
        ctx.save_for_backward(xs.detach(), cs.detach())
        ctx.needs = (xs.requires_grad, cs.requires_grad)
        ctx.ws = net.checkout_train_workspace(xs.shape[0], xs.shape[1], cs.shape[1])
        return net._run_fwd(xs.detach(), ts, cs.detach(), training=True, ws=ctx.ws)

    @staticmethod
    def backward(ctx, gout):
        net = ctx.net
        xs, cs = ctx.saved_tensors
        gout = gout.contiguous().to(torch.float32)
        if ctx.ws is None:
            raise RuntimeError("CustomTransformer: backward through the same forward twice is not supported (its workspace was released)")
        grads = torch.empty_like(net._flat)  # written (not accumulated into) by the backward: no zeroing pass
        gx, gc = net._run_bwd(xs, cs, gout, grads, *ctx.needs, accumulate=False, ws=ctx.ws)
        net.return_train_workspace(xs.shape[0], xs.shape[1], cs.shape[1], ctx.ws)
        ctx.ws = None
        pg = [grads[o:o + math.prod(shape)].view(shape) for _, o, shape in net._layout]
        return (None, gx, None, gc, *pg)


class DDIMTransformerAdapter(nn.Module):
    """``model(x_t, t, init_cond, attn_cond)`` as DDIMDiffusionModel calls it (reference model.py:271, 276, 359, 374), served
    by the 3-argument transformer: the MS1 chromatogram ``attn_cond`` (B, RT) is its conditional sequence; the MS2 mixture
    ``init_cond`` has no input on this network (building_blocks.py:224) and is ignored."""

    def __init__(self, transformer: CustomTransformer):
        super().__init__()
        self.transformer = transformer

    # checkpoints carry the transformer's own keys (no "transformer." prefix): interchangeable with the reference module's
    def state_dict(self, *args, **kwargs):
        return self.transformer.state_dict(*args, **kwargs)

    def load_state_dict(self, state_dict, *args, **kwargs):
        return self.transformer.load_state_dict(state_dict, *args, **kwargs)

    # the flat-buffer surface FlatAdamW and the data-parallel all-reduce use (model_interface.py)
    @property
    def _layout(self):
        return self.transformer._layout

    @property
    def flat_params(self):
        return self.transformer.flat_params

    def flat_grads(self, zero: bool = False):
        return self.transformer.flat_grads(zero=zero)

    def trainable_named(self):
        return self.transformer.trainable_named()

    def forward(self, x_t, t, init_cond=None, attn_cond=None):
        if attn_cond is None:
            raise ValueError("DDIMTransformerAdapter: attn_cond (MS1, (B, RT)) is required")
        if attn_cond.dim() == 3:
            attn_cond = attn_cond[..., 0]
        return self.transformer(x_t, t, attn_cond)
