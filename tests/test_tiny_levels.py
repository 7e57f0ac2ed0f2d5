"""GPU parity of the round-4 deep-level kernels against the launches they replace, on the SAME seeded network and inputs:
k_tiny (levels with rows of 1 / 2 positions: stage + ResnetBlocks + the n = 1 LinearAttention + the bottleneck folds) against
k_level_fwd / k_linattn_fwd / k_conv_fwd / k_fold (DQ_NO_TINY=1), k_la_small against the register-resident LinearAttention
(dq_set_option: the register-resident forms at every row count), and the side-stream scheduling of the train step against the single chain (DQ_NO_FWD_FORK=1 DQ_NO_TAIL_FORK=1).
The switches live in the DEVELOPMENT build of the library only (csrc/dq_dev.h; the product reads no environment variable) and are read once per
process, so every variant runs in its own child process (one at a time) on build/dev/libdq_hip_dev.so and leaves an .npz.
RT = 70 leaves the last tile of every sample partly filled (32-row and 64-row tiles)."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import REPO

pytestmark = pytest.mark.gpu
DEV_LIB = os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd", "build", "dev", "libdq_hip_dev.so")

CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1])
from dquartic.model.model import DDIMDiffusionModel
from dquartic.model.unet1d import UNet1d
import os
from dquartic import _native as N
if os.environ.get("TEST_LA_FORM") == "rows":  # the per-row LinearAttention forms at every row count (the product's rule picks by row count)
    N.set_option("la_small_min_rows", 0); N.set_option("la_rows_bwd_min_rows", 0)
elif os.environ.get("TEST_LA_FORM") == "register":
    N.set_option("la_small_min_rows", 1 << 40); N.set_option("la_rows_bwd_min_rows", 1 << 40)
torch.manual_seed(3)
net = UNet1d(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1, attn_cond_channels=1, downsample_dim=64, simple=True).cuda()
with torch.no_grad():
    for _, p in net.trainable_named():
        p.add_(torch.randn_like(p) * 0.05)  # biases / gains off their initial 0 / 1
dm = DDIMDiffusionModel(model_class=net, device="cuda")
g = torch.Generator().manual_seed(11)
B, RT, MZ = (int(sys.argv[3]), int(sys.argv[4]), 64) if len(sys.argv) > 4 else (3, 70, 64)
x0 = torch.rand(B, RT, MZ, generator=g).cuda(); c2 = torch.rand(B, RT, MZ, generator=g).cuda(); c1 = torch.rand(B, RT, generator=g).cuda()
t = torch.tensor([999, 417, 3] * ((B + 2) // 3), dtype=torch.long)[:B].cuda(); noise = torch.randn(B, RT, MZ, generator=g).cuda()
with torch.no_grad():
    eps = net(dm.q_sample(dm.normalize(x0), t, noise), t, dm.normalize(c2), dm.normalize(c1))
loss = dm.train_step_fused(x0, c2, c1, t=t, noise=noise, zero_grads=True)
xs, pn = dm.sample(torch.randn(B, RT, MZ, generator=g).cuda(), c2, c1, num_steps=4)
torch.cuda.synchronize()
np.savez(sys.argv[2], eps=eps.cpu().numpy(), loss=float(loss), grads=net.flat_grads().cpu().numpy(), xs=xs.cpu().numpy(), pn=pn.cpu().numpy())
"""


def _run(tmp_path, tag, env, shape=()):
    out = str(tmp_path / f"{tag}.npz")
    e = dict(os.environ)
    e.update(env)
    # the A-B switches (DQ_NO_*) exist only in the development build of the library (csrc/dq_dev.h, `make dev`): the product build reads no
    # environment variable.  BOTH sides of a comparison run the development build, whose default paths are the product's.
    assert os.path.exists(DEV_LIB), "build the development library first: make -C diffusion-deconvolution-dia-msms-data_amd dev (or __graft_entry__.build())"
    e["DQ_HIP_LIB"] = DEV_LIB
    e.setdefault("TEST_LA_FORM", "rows")  # (the deep-level kernels under test include the per-row LinearAttention forms)
    r = subprocess.run([sys.executable, "-c", CHILD, os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd"), out] + [str(v) for v in shape], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    return dict(np.load(out))


def _rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def test_deep_level_kernels_match_the_launches_they_replace(tmp_path):
    new = _run(tmp_path, "new", {})
    old = _run(tmp_path, "old", {"DQ_NO_TINY": "1", "TEST_LA_FORM": "register", "DQ_NO_FWD_FORK": "1", "DQ_NO_TAIL_FORK": "1", "DQ_NO_TRAIN_INIT": "1", "DQ_NO_HEAD_LOSS": "1"})
    assert _rel(new["eps"], old["eps"]) < 1e-5          # network output (inference path: head epilogue, no saved tensors)
    assert abs(new["loss"] - old["loss"]) < 2e-6 * abs(old["loss"])
    # gradients: the forward's saved tensors differ in the last bits; the heavily cancelling tensors bound the flat comparison
    assert _rel(new["grads"], old["grads"]) < 1e-4
    assert _rel(new["xs"], old["xs"]) < 2e-5 and _rel(new["pn"], old["pn"]) < 2e-5  # 4 DDIM steps


def test_side_stream_schedule_is_bitwise_neutral(tmp_path):
    """Moving launches to the side stream changes WHEN they run, not what they compute: bit-identical loss and gradients."""
    a = _run(tmp_path, "fork", {})
    b = _run(tmp_path, "chain", {"DQ_NO_FWD_FORK": "1", "DQ_NO_TAIL_FORK": "1"})
    assert a["loss"] == b["loss"]
    assert np.array_equal(a["grads"], b["grads"]) and np.array_equal(a["eps"], b["eps"])


def test_training_head_matches_the_three_launches_it_replaces(tmp_path):
    """final_conv + squared error + d eps + final_conv's backward data path in the final block's launch (k_level_fwd's training head) against
    k_conv_fwd / k_mse_fwd_bwd / k_conv_bwd_data (DQ_NO_HEAD_LOSS=1): the per-element arithmetic is the same, so every gradient is bit-identical;
    the loss is summed in another (fixed) order."""
    a = _run(tmp_path, "head", {})
    b = _run(tmp_path, "nohead", {"DQ_NO_HEAD_LOSS": "1"})
    assert np.array_equal(a["grads"], b["grads"])
    assert abs(a["loss"] - b["loss"]) < 2e-6 * abs(b["loss"])


def test_q_sample_in_the_first_level_launch_is_bitwise_neutral(tmp_path):
    """x_t = sqrt(ab) x0 + sqrt(1 - ab) noise formed by level 0's INIT stage (the arithmetic of k_q_sample) against the launch (DQ_NO_QSAMPLE_FUSE=1)."""
    a = _run(tmp_path, "qsf", {})
    b = _run(tmp_path, "noqsf", {"DQ_NO_QSAMPLE_FUSE": "1"})
    assert a["loss"] == b["loss"] and np.array_equal(a["grads"], b["grads"])


def test_upsample_transpose_in_the_tiny_backward(tmp_path):
    """The Upsample conv behind the deepest up level: its backward data path as a stage of k_tiny_bwd (+ the generic weight-gradient kernel on the
    side stream) against k_conv_bwd_wg (DQ_NO_TINY_UPT=1)."""
    a = _run(tmp_path, "upt", {})
    b = _run(tmp_path, "noupt", {"DQ_NO_TINY_UPT": "1"})
    assert a["loss"] == b["loss"]
    assert _rel(a["grads"], b["grads"]) < 2e-5


def test_large_batch_keeps_the_per_kernel_backward(tmp_path):
    """More samples than a LinearAttention layer's slot reservation holds workgroups of the tiny backward (one slot each): the launch is declined
    and the per-kernel backward runs -- same loss, gradients within the fp32 tolerance of the small-batch comparison."""
    new = _run(tmp_path, "big_new", {}, shape=(1100, 4))
    old = _run(tmp_path, "big_old", {"DQ_NO_TINY_BWD": "1"}, shape=(1100, 4))
    assert abs(new["loss"] - old["loss"]) < 2e-6 * abs(old["loss"])
    assert _rel(new["grads"], old["grads"]) < 1e-4
