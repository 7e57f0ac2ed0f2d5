"""CPU oracle for the reference's alternative noise predictor ``CustomTransformer``.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
(``dquartic.model.building_blocks`` on top of ``libdq_hip.so``) never does.

A from-scratch functional restatement in plain PyTorch fp32 on the CPU of ``dquartic/model/building_blocks.py``; no
``nn.MultiheadAttention`` / ``nn.LayerNorm`` modules -- the arithmetic is spelled out so that the HIP path can be read against
it.  Parameters are a flat ``dict`` under the reference's ``state_dict`` keys.

Parity status: PINNED.  ``building_blocks.py`` imports in the build container with no stand-ins (it depends on torch only);
``oracle/make_golden_tfm.py`` records the reference's forward output and its autograd gradients for a small configuration in
``tests/golden/tfm_tiny.npz`` and ``tests/test_oracle_golden.py`` checks this restatement against them.
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import torch

Params = Dict[str, torch.Tensor]
LN_EPS = 1e-5  # nn.LayerNorm default (building_blocks.py:139,145)


def rope_tables(seqlen: int, hidden_dim: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """building_blocks.py:31-49: inv_freq_j = 10000 ** -(j / half), angle[s][j] = s * inv_freq_j; fp32 throughout."""
    half = hidden_dim // 2
    freq_seq = torch.arange(half, dtype=torch.float32) / half
    inv_freq = 10000 ** (-freq_seq)
    positions = torch.arange(seqlen, dtype=torch.float32)
    angles = torch.einsum("i,j->ij", positions, inv_freq)
    return torch.sin(angles), torch.cos(angles)


def apply_rope(x: torch.Tensor) -> torch.Tensor:
    """building_blocks.py:6-66: adjacent channel pairs (2j, 2j+1) of the HIDDEN axis rotated by the angle of (position, j)."""
    B, S, H = x.shape
    assert H % 2 == 0, "hidden_dim must be even"
    sin, cos = rope_tables(S, H)
    x1, x2 = x[..., 0::2], x[..., 1::2]
    r1 = x1 * cos - x2 * sin
    r2 = x1 * sin + x2 * cos
    return torch.stack([r1, r2], dim=-1).reshape(B, S, H)


def time_freqs(hidden_dim: int) -> torch.Tensor:
    """building_blocks.py:104-106."""
    half = hidden_dim // 2
    return torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))


def gelu(x: torch.Tensor) -> torch.Tensor:
    return 0.5 * x * (1.0 + torch.erf(x * 0.7071067811865476))  # nn.GELU() default = exact erf form


def linear(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    return x @ w.t() + b


def layer_norm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)  # biased, like nn.LayerNorm
    return (x - mu) / torch.sqrt(var + LN_EPS) * g + b


def time_embedding(p: Params, t: torch.Tensor, hidden_dim: int, prefix: str = "time_embedding.") -> torch.Tensor:
    """building_blocks.py:92-112: [sin(t f) | cos(t f)] -> Linear(H, 4H) -> GELU -> Linear(4H, H)."""
    emb = t[:, None] * time_freqs(hidden_dim)[None, :]  # int64 t promotes to fp32
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=1)
    emb = gelu(linear(emb, p[prefix + "linear1.weight"], p[prefix + "linear1.bias"]))
    return linear(emb, p[prefix + "linear2.weight"], p[prefix + "linear2.bias"])


def attention(p: Params, prefix: str, query: torch.Tensor, kv: torch.Tensor, num_heads: int) -> torch.Tensor:
    """nn.MultiheadAttention(batch_first=True, need_weights=False) as called at building_blocks.py:164-166:
    q = query Wq^T + bq, k|v = kv Wk|v^T + bk|v, heads split along the hidden axis, softmax(q k^T / sqrt(dh)) v, out_proj."""
    B, S1, H = query.shape
    Sk = kv.shape[1]
    dh = H // num_heads
    w, b = p[prefix + "in_proj_weight"], p[prefix + "in_proj_bias"]
    q = linear(query, w[:H], b[:H]).reshape(B, S1, num_heads, dh).transpose(1, 2)
    k = linear(kv, w[H:2 * H], b[H:2 * H]).reshape(B, Sk, num_heads, dh).transpose(1, 2)
    v = linear(kv, w[2 * H:], b[2 * H:]).reshape(B, Sk, num_heads, dh).transpose(1, 2)
    prob = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(dh), dim=-1)
    o = (prob @ v).transpose(1, 2).reshape(B, S1, H)
    return linear(o, p[prefix + "out_proj.weight"], p[prefix + "out_proj.bias"])


def layer(p: Params, prefix: str, x_t: torch.Tensor, x_cond: torch.Tensor, num_heads: int) -> torch.Tensor:
    """building_blocks.py:147-176: queries = x_t, keys/values = [x_cond ; x_t]; post-norm residual blocks."""
    combined = torch.cat([x_cond, x_t], dim=1)
    x_t = layer_norm(x_t + attention(p, prefix + "attention.", x_t, combined, num_heads), p[prefix + "norm1.weight"], p[prefix + "norm1.bias"])
    ff = linear(gelu(linear(x_t, p[prefix + "ff.0.weight"], p[prefix + "ff.0.bias"])), p[prefix + "ff.2.weight"], p[prefix + "ff.2.bias"])
    return layer_norm(x_t + ff, p[prefix + "norm2.weight"], p[prefix + "norm2.bias"])


def forward(p: Params, x_t: torch.Tensor, t: torch.Tensor, x_cond: torch.Tensor, num_heads: int) -> torch.Tensor:
    """building_blocks.py:224-260.  x_t (B, S1, input_dim); t (B,); x_cond (B, S2) -- one value per conditional position (the
    reference unsqueezes it and projects with Linear(1, hidden))."""
    H = p["input_projection.weight"].shape[0]
    n_layers = 1 + max(int(k.split(".")[1]) for k in p if k.startswith("layers."))
    x = apply_rope(linear(x_t, p["input_projection.weight"], p["input_projection.bias"]))
    c = apply_rope(linear(x_cond.unsqueeze(2), p["conditional_projection.weight"], p["conditional_projection.bias"]))
    x = x + time_embedding(p, t, H)[:, None, :]
    for i in range(n_layers):
        x = layer(p, f"layers.{i}.", x, c, num_heads)
    return linear(x, p["output_projection.weight"], p["output_projection.bias"])


def init_params(input_dim: int, hidden_dim: int, num_layers: int, seed: int = 0, scale: float = 1.0) -> Params:
    """Random parameters under the reference's state_dict keys in its registration order (building_blocks.py:205-222,
    84-89, 133-145) -- an initialisation for tests and benchmarks, not the reference's nn.Linear / xavier defaults."""
    g = torch.Generator().manual_seed(seed)
    H = hidden_dim

    def lin(o, i):
        return torch.randn(o, i, generator=g) * (scale / math.sqrt(i)), torch.randn(o, generator=g) * 0.1

    p: Params = {}
    p["input_projection.weight"], p["input_projection.bias"] = lin(H, input_dim)
    p["output_projection.weight"], p["output_projection.bias"] = lin(input_dim, H)
    p["conditional_projection.weight"], p["conditional_projection.bias"] = lin(H, 1)
    p["time_embedding.linear1.weight"], p["time_embedding.linear1.bias"] = lin(4 * H, H)
    p["time_embedding.linear2.weight"], p["time_embedding.linear2.bias"] = lin(H, 4 * H)
    for i in range(num_layers):
        pre = f"layers.{i}."
        p[pre + "attention.in_proj_weight"], p[pre + "attention.in_proj_bias"] = lin(3 * H, H)
        p[pre + "attention.out_proj.weight"], p[pre + "attention.out_proj.bias"] = lin(H, H)
        p[pre + "norm1.weight"], p[pre + "norm1.bias"] = 1.0 + 0.1 * torch.randn(H, generator=g), 0.1 * torch.randn(H, generator=g)
        p[pre + "ff.0.weight"], p[pre + "ff.0.bias"] = lin(4 * H, H)
        p[pre + "ff.2.weight"], p[pre + "ff.2.bias"] = lin(H, 4 * H)
        p[pre + "norm2.weight"], p[pre + "norm2.bias"] = 1.0 + 0.1 * torch.randn(H, generator=g), 0.1 * torch.randn(H, generator=g)
    return p
