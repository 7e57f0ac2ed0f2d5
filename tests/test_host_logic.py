"""CPU: host-side mirror of the reference interface -- parameter layout / state_dict compatibility, default
initialisation, flat buffers, schedulers, checkpoints, config, datasets.  No kernels run here."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import sub

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KW = dict(dim=4, channels=1, dim_mults=(1, 2, 2, 3, 3, 4, 4), conditional=True, init_cond_channels=1, attn_cond_channels=1,
          tfer_dim_mult=620, downsample_dim=64, simple=True)


def test_state_dict_matches_reference_keys_order_shapes(golden):
    from dquartic.model.unet1d import UNet1d

    g = golden("unet_default_rt16.npz")
    ref = [(k[2:], g[k].shape) for k in g if k.startswith("w/")]
    sd = UNet1d(**KW).state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == ref
    assert len(ref) == 396  # 395 trainable + RoPE freqs


def test_default_init_equals_reference_under_same_seed(golden):
    from dquartic.model.unet1d import UNet1d

    g = golden("init_seed123.npz")
    torch.manual_seed(123)
    sd = UNet1d(**KW).state_dict()
    for k, v in sd.items():
        assert np.array_equal(v.reshape(-1)[:4].numpy(), g[f"head/{k}"]), k
        assert abs(v.double().sum().item() - float(g[f"sum/{k}"])) <= 1e-9 * max(1.0, abs(float(g[f"sum/{k}"]))), k


def test_flat_buffer_views_survive_load_and_move(golden):
    from dquartic.model.unet1d import UNet1d

    g = golden("unet_default_rt16.npz")
    net = UNet1d(**KW)
    net.load_state_dict(sub(g, "w/"))
    flat = net.flat_params
    assert flat.numel() == 128847
    for name, off, shape in net._layout:
        p = dict(net.named_parameters())[name]
        assert p.data_ptr() == flat.data_ptr() + 4 * off and tuple(p.shape) == shape
        assert torch.equal(p.detach().reshape(-1), flat[off:off + p.numel()])
    net = net.double().float()  # _apply replaces parameter storage; the flat view must be re-established
    flat2 = net.flat_params
    p0 = dict(net.named_parameters())["init_conv.weight"]
    assert p0.data_ptr() == flat2.data_ptr() and torch.equal(flat2, flat)
    gr = net.flat_grads(zero=True)
    assert all(p.grad is not None and p.grad.data_ptr() == gr.data_ptr() + 4 * off for (n, off, _), p in
               zip(net._layout, [dict(net.named_parameters())[n] for n, _, _ in net._layout]))


def test_no_cpu_fallback_and_unsupported_configs():
    from dquartic.model.unet1d import UNet1d

    net = UNet1d(**KW)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 4, 64), torch.zeros(1, dtype=torch.long), torch.zeros(1, 4, 64), torch.zeros(1, 4))
    with pytest.raises(NotImplementedError):
        UNet1d(**{**KW, "simple": False})
    with pytest.raises(NotImplementedError):
        UNet1d(**{**KW, "channels": 3})
    with pytest.raises(ValueError):
        UNet1d(**{**KW, "downsample_dim": 100})  # MZ not divisible by 2**6 must fail loudly


def test_schedule_and_bad_pred_type(golden):
    from dquartic.model.model import DDIMDiffusionModel, extract, get_cosine_beta_schedule
    from dquartic.model.unet1d import UNet1d

    s = golden("schedule.npz")
    dm = DDIMDiffusionModel(model_class=UNet1d(**KW), device="cpu")
    assert np.array_equal(dm.alpha_bars.numpy(), s["cosine/alpha_bars"]) and np.array_equal(dm.alphas.numpy(), s["cosine/alphas"])
    dl = DDIMDiffusionModel(model_class=UNet1d(**KW), beta_schedule_type="linear", device="cpu")
    assert np.array_equal(dl.betas.numpy(), s["linear/betas"])
    assert dm.sampler_timesteps(1000, 50).tolist() == s["timesteps50"].tolist()
    assert get_cosine_beta_schedule(10).dtype == torch.float64
    assert extract(dm.alpha_bars, torch.tensor([0, 999]), (2, 3, 4)).shape == (2, 1, 1)
    with pytest.raises(ValueError, match="pred_type"):
        DDIMDiffusionModel(model_class=UNet1d(**KW), pred_type="v", device="cpu")
    assert torch.equal(dm.loss_weight, torch.ones(1000))


def test_warmup_cosine_lr_schedule():
    from dquartic.model.model_interface import WarmupLR_Scheduler

    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    sch = WarmupLR_Scheduler(opt, num_warmup_steps=5, num_training_steps=25)
    lrs = []
    for _ in range(25):
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sch.step()
    assert np.allclose(lrs[:5], [0.2, 0.4, 0.6, 0.8, 1.0])  # (step+1)/warm (reference model_interface.py:149-150)
    assert abs(lrs[15] - 0.5 * (1 + np.cos(np.pi * 0.5))) < 1e-6 and lrs[-1] < 0.01 and all(l >= 1e-10 for l in lrs)


def test_checkpoint_roundtrip_and_optimizer_state_layout(tmp_path):
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.model_interface import FlatAdamW
    from dquartic.model.unet1d import UNet1d

    dm = DDIMDiffusionModel(model_class=UNet1d(**KW), device="cpu")
    dm._set_optimizer(1e-5)
    assert isinstance(dm.optimizer, FlatAdamW)
    dm.optimizer._buffers()
    dm.optimizer._m.uniform_(), dm.optimizer._v.uniform_()
    dm.optimizer._step = 7
    dm.optimizer._publish_state()
    sch = dm._get_lr_schedule_with_warmup(2, 10)
    path = str(tmp_path / "ck.ckpt")
    dm.save_checkpoint(sch, 3, 0.25, path)
    ck = torch.load(path, weights_only=False)
    assert set(ck) == {"epoch", "model_state_dict", "optimizer_state_dict", "scheduler_state_dict", "best_loss"}
    st = ck["optimizer_state_dict"]["state"]
    assert len(st) == 395 and set(st[0]) == {"step", "exp_avg", "exp_avg_sq"}  # torch AdamW layout
    dm2 = DDIMDiffusionModel(model_class=UNet1d(**KW), device="cpu")
    dm2._set_optimizer(1e-5)
    sch2 = dm2._get_lr_schedule_with_warmup(2, 10)
    ep, best, _ = dm2.load_checkpoint(sch2, path, "cpu")
    assert (ep, best) == (3, 0.25)
    assert torch.equal(dm2.model.flat_params, dm.model.flat_params)
    assert torch.equal(dm2.optimizer._m, dm.optimizer._m) and torch.equal(dm2.optimizer._v, dm.optimizer._v) and dm2.optimizer._step == 7
    ep0, best0, _ = dm2.load_checkpoint(None, str(tmp_path / "missing.ckpt"), "cpu")
    assert ep0 == 0 and best0 == float("inf")


def test_config_loader_and_default_config(tmp_path):
    from dquartic.utils.config_loader import generate_train_config, load_train_config

    p = str(tmp_path / "c.json")
    generate_train_config(p)
    cfg = load_train_config(p, batch_size="8", threads="3", use_wandb=False, checkpoint_path="x/y.ckpt")
    assert cfg["model"]["batch_size"] == 8 and cfg["threads"] == 3 and cfg["wandb"]["use_wandb"] is False
    assert cfg["model"]["checkpoint_path"] == "x/y.ckpt" and cfg["model"]["UNet1d"]["dim_mults"] == [1, 2, 2, 3, 3, 4, 4]
    assert set(json.load(open(p))) == {"data", "model", "wandb", "threads"}


def test_diams_dataset_npy_and_parquet(tmp_path):
    import pyarrow as pa
    import pyarrow.parquet as pq

    from dquartic.utils.data_loader import DIAMSDataset

    rng = np.random.default_rng(0)
    ms2, ms1 = rng.random((5, 6, 8)).astype(np.float32) * 50, rng.random((5, 6)).astype(np.float32) * 9
    np.save(tmp_path / "ms2.npy", ms2), np.save(tmp_path / "ms1.npy", ms1)
    ds = DIAMSDataset(ms2_file=str(tmp_path / "ms2.npy"), ms1_file=str(tmp_path / "ms1.npy"), normalize="minmax")
    assert len(ds) == 5
    seen = set()
    for _ in range(10):  # all 10 distinct pairs exactly once per epoch
        a, m1, b, m2 = ds[0]
        assert a.shape == (6, 8) and m1.shape == (6,) and a.dtype == torch.float32
        assert float(min(a.min(), b.min())) == 0.0 and float(max(a.max(), b.max())) == 1.0  # pair-wise min-max
        assert float(m1.min()) == 0.0 and float(m1.max()) == 1.0                              # MS1 statistics from window 1
    assert len(ds.used_pairs) == 10
    ds.reset_epoch()
    assert not ds.used_pairs and ds.epoch_reset
    with pytest.raises(ValueError):
        DIAMSDataset(ms2_file=str(tmp_path / "ms2.npy"), ms1_file=str(tmp_path / "ms1.npy"), normalize=None)[0]
    with pytest.raises(ValueError):
        DIAMSDataset()
    # parquet slices with the reference ETL's schema (data_generation.py:206-223)
    rows = [{"file": "f", "slice_index": i, "mz_isolation_target": 400.0 + i, "mz_start": 0.0, "mz_end": 1.0, "rt_start": 0.0,
             "rt_end": 1.0, "ms1_data": ms1[i].tolist(), "ms2_data": ms2[i].reshape(-1).tolist(), "ms1_shape": [6],
             "ms2_shape": [6, 8], "rt_values": [0.0], "mz_values_ms1": [0.0], "mz_values_ms2": [0.0]} for i in range(4)]
    (tmp_path / "pq").mkdir()
    pq.write_table(pa.Table.from_pylist(rows), str(tmp_path / "pq" / "a.parquet"))
    dp = DIAMSDataset(parquet_directory=str(tmp_path / "pq"), normalize="minmax")
    a, m1, b, m2 = dp[0]
    assert len(dp) == 4 and a.shape == (6, 8) and m2.shape == (6,)


def test_synthetic_dataset_contract_and_sharding():
    from dquartic.utils.synthetic import SyntheticDIAMSDataset, make_window

    w0, w0b = make_window(3, 40, 16), make_window(3, 40, 16)
    assert np.array_equal(w0[0], w0b[0]) and w0[0].shape == (40, 16) and w0[1].shape == (40,) and w0[0].min() >= 0
    d0 = SyntheticDIAMSDataset(8, RT=40, MZ=16, rank=0, world=2)
    d1 = SyntheticDIAMSDataset(8, RT=40, MZ=16, rank=1, world=2)
    assert len(d0) == len(d1) == 4 and not any(np.array_equal(a, b) for a in d0.ms2 for b in d1.ms2)
    a, m1, b, m2 = d0[0]
    assert a.shape == (40, 16) and float(max(a.max(), b.max())) == 1.0
    d0.reset_epoch()


def test_train_one_epoch_builds_the_reference_triple(golden):
    """target = ms2_1, MS1 cond = ms1_1, mixture = 0.5/0.5, ms1_2 unused (reference model_interface.py:1070-1075)"""
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d

    g = golden("harness.npz")
    seen = {}

    class Cap(DDIMDiffusionModel):
        def _train_one_batch(self, x_0, ms2_cond=None, ms1_cond=None, noise=None, ms1_loss_weight=0.0, **kw):
            seen.update(x_0=x_0, ms2_cond=ms2_cond, ms1_cond=ms1_cond)
            return 0.0

    cap = Cap(model_class=UNet1d(**{**KW, "dim_mults": (1, 2), "downsample_dim": 8}), device="cpu")
    T = torch.from_numpy
    out = cap._train_one_epoch(0, [(T(g["in/ms2_1"]), T(g["in/ms1_1"]), T(g["in/ms2_2"]), T(g["in/ms1_2"]))])
    assert out == [0.0]
    for k in ("x_0", "ms2_cond", "ms1_cond"):
        assert np.array_equal(seen[k].numpy(), g[f"out/{k}"]), k


def test_train_signatures_keep_the_reference_positional_order(tmp_path):
    """``train_with_warmup(loader, 100, 5, 1e-4)`` must mean 100 epochs, 5 warm-up epochs, lr 1e-4 as in the reference
    (model_interface.py:348-357), and ``train`` keeps (dataloader, batch_size, epochs, warmup_epochs, learning_rate, use_wandb,
    checkpoint_path) (:453-463)."""
    import inspect

    from dquartic.model.model_interface import ModelInterface

    names = lambda f: [p.name for p in inspect.signature(f).parameters.values() if p.kind == p.POSITIONAL_OR_KEYWORD][1:]
    assert names(ModelInterface.train_with_warmup) == ["dataloader", "num_epochs", "num_warmup_steps", "learning_rate", "use_wandb",
                                                       "log_every_n_epochs", "checkpoint_path"]
    assert names(ModelInterface.train) == ["dataloader", "batch_size", "epochs", "warmup_epochs", "learning_rate", "use_wandb",
                                           "checkpoint_path"]

    class Harness(ModelInterface):
        def __init__(self):
            super().__init__(device="cpu")
            self.build(torch.nn.Linear(2, 2))
            self.seen = []

        def _train_one_batch(self, x_0, ms2_cond=None, ms1_cond=None, noise=None, ms1_loss_weight=0.0, **kw):
            self.seen.append(self.optimizer.param_groups[0]["lr"])
            return 1.0 / len(self.seen)

    class DS(list):
        def reset_epoch(self):
            pass

    class Loader(list):
        dataset = DS()

    z = torch.zeros(1, 2, 2)
    h = Harness()
    h.train_with_warmup(Loader([(z, z[..., 0], z, z[..., 0])]), 6, 3, 1e-2, False, 100, str(tmp_path / "best.ckpt"))
    assert len(h.seen) == 6                                        # 6 epochs, not "3"
    assert abs(h.seen[0] - 1e-2 / 3) < 1e-12 and abs(h.seen[2] - 1e-2) < 1e-12   # 3 warm-up epochs up to lr = 1e-2
    assert (tmp_path / "best.ckpt").exists() and (tmp_path / "dquartic_latest_checkpoint.ckpt").exists()
    h2 = Harness()
    (tmp_path / "c").mkdir()
    h2.train(Loader([(z, z[..., 0], z, z[..., 0])]), 1, 2, 0, 5e-3, False, str(tmp_path / "c" / "b.ckpt"))
    assert h2.seen == [5e-3, 5e-3]                                 # warmup_epochs <= 0: constant lr (reference :498-559)
