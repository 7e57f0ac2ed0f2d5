"""CPU: the lane-level emulation of the shipped LinearAttention algorithm (oracle/wave_emu.py::la_fwd_reassoc_row -- the
re-associated M / P / W2 form on the 32x32x2 and 4x4x1 MFMA lane maps) against the oracle's restatement of the reference."""
import numpy as np
import pytest
import torch

from oracle import dq_oracle as O
from oracle import wave_emu as E


@pytest.mark.parametrize("C,n", [(4, 64), (8, 32), (4, 32)])
def test_reassociated_forward_matches_oracle(C, n):
    rng = np.random.default_rng(C * 100 + n)
    x = rng.standard_normal((3, C, n)).astype(np.float32)
    Wqkv = (rng.standard_normal((384, C)) * 0.4).astype(np.float32)
    Wo = (rng.standard_normal((C, 128)) * 0.2).astype(np.float32)
    bo = (rng.standard_normal(C) * 0.1).astype(np.float32)
    g_pre = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    g_out = (1 + 0.1 * rng.standard_normal(C)).astype(np.float32)
    p = {"la.fn.norm.g": torch.from_numpy(g_pre).reshape(1, C, 1), "la.fn.fn.to_qkv.weight": torch.from_numpy(Wqkv).reshape(384, C, 1),
         "la.fn.fn.to_out.0.weight": torch.from_numpy(Wo).reshape(C, 128, 1), "la.fn.fn.to_out.0.bias": torch.from_numpy(bo),
         "la.fn.fn.to_out.1.g": torch.from_numpy(g_out).reshape(1, C, 1)}
    ref = O.linear_attention(p, "la", torch.from_numpy(x)).numpy()
    for r in range(x.shape[0]):
        y = E.la_fwd_reassoc_row(x[r], Wqkv, Wo, bo, g_pre, g_out)
        assert np.abs(y - ref[r]).max() <= 2e-5 * max(1.0, np.abs(ref[r]).max())


def test_split_bf16_parts_are_exact():
    """The split the projections rest on (k_linattn.hip: la_split_x / la_bf16_image_dword), restated in numpy: an fp32 value is EXACTLY
    H + M + L with every part a bf16 (top 16 bits), and the six kept part products reproduce x * w to 2^-21 of |x||w| in the worst case (the dropped M*L, L*M, L*L terms), ~2^-23 typically."""
    rng = np.random.default_rng(0)
    v = np.concatenate([rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-6, 6, 20000).astype(np.float32),
                        np.array([0.0, 1.0, -1.0, 3.4e38, 1.2e-30, -7.0], np.float32)])

    def split(a):
        h = (a.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
        r = a - h
        m = (r.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)
        lo = r - m
        assert np.all((lo.view(np.uint32) & np.uint32(0xFFFF)) == 0)  # the rest fits a bf16: nothing is lost
        return h, m, lo

    h, m, lo = split(v)
    assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + lo.astype(np.float64), v.astype(np.float64))
    w = rng.standard_normal(v.size).astype(np.float32)
    wh, wm, wl = split(w)
    kept = sum(a.astype(np.float64) * b.astype(np.float64) for a, b in ((h, wh), (h, wm), (m, wh), (m, wm), (h, wl), (lo, wh)))
    exact = v.astype(np.float64) * w.astype(np.float64)
    assert np.max(np.abs(kept - exact) / np.maximum(np.abs(exact), 1e-300)) < 2.0 ** -20
