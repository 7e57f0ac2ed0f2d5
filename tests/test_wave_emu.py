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
