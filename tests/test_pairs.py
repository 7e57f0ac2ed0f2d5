"""Batch formation: per-pair min-max normalisation + mixture.
CPU: the oracle restatement and the host ``DIAMSDataset`` against fixtures captured from the reference's own
``DIAMSDataset`` (oracle/make_golden_pairs.py -> tests/golden/pairs.npz).  GPU: ``dq_pair_batch`` through
``ResidentPairLoader`` -- bit-exact against the same fixtures and against the oracle at BASELINE window size."""
import random

import numpy as np
import pytest
import torch

from oracle import dq_oracle as O

KEYS = ("ms2_1", "ms1_1", "ms2_2", "ms1_2")


def _write(tmp_path, g):
    np.save(tmp_path / "ms2.npy", g["ms2"])
    np.save(tmp_path / "ms1.npy", g["ms1"])
    return str(tmp_path / "ms2.npy"), str(tmp_path / "ms1.npy")


def test_oracle_pair_batch_bit_exact(golden):
    g = golden("pairs.npz")
    pairs = g["pairs"]
    for w in ((0.5, 0.5), (0.7, 0.3)):
        outs = O.pair_batch(g["ms2"], g["ms1"], pairs[:, 0], pairs[:, 1], w)
        for k in range(len(pairs)):
            for got, key in zip(outs, KEYS + (f"cond_{w[0]}_{w[1]}",)):
                assert np.array_equal(got[k], g[f"item{k}/{key}"], equal_nan=True), (k, key)
    a, m1, b, m2, c = O.pair_batch(g["ms2"], g["ms1"], [5], [6])
    assert np.isnan(a).all() and np.isnan(b).all() and np.isnan(c).all()  # constant pair: 0/0 like the reference
    for got, key in zip((a, m1, b, m2), KEYS):
        assert np.array_equal(got[0], g[f"const/{key}"], equal_nan=True)


def test_host_dataset_matches_reference_stream(golden, tmp_path):
    from dquartic.utils.data_loader import DIAMSDataset

    g = golden("pairs.npz")
    f2, f1 = _write(tmp_path, g)
    ds = DIAMSDataset(ms2_file=f2, ms1_file=f1, normalize="minmax")
    random.seed(int(g["seed"]))
    with np.errstate(invalid="ignore", divide="ignore"):
        for k in range(len(g["pairs"])):
            item = ds[0]
            for got, key in zip(item, KEYS):
                assert got.dtype == torch.float32
                assert np.array_equal(got.numpy(), g[f"item{k}/{key}"], equal_nan=True), (k, key)


@pytest.mark.gpu
def test_resident_loader_bit_exact_golden(golden, tmp_path):
    from dquartic.utils.data_loader import DIAMSDataset, PairBatch, ResidentPairLoader

    g = golden("pairs.npz")
    f2, f1 = _write(tmp_path, g)
    ds = DIAMSDataset(ms2_file=f2, ms1_file=f1, normalize="minmax")
    pairs = g["pairs"]
    for w in ((0.5, 0.5), (0.7, 0.3)):
        ld = ResidentPairLoader(ds, batch_size=16, device="cuda", mixture_weights=w)
        pb = ld.form(pairs[:, 0].tolist(), pairs[:, 1].tolist())
        assert isinstance(pb, PairBatch) and len(pb) == 4 and pb.mixture_weights == w
        for k in range(len(pairs)):
            for got, key in zip(pb, KEYS):
                assert np.array_equal(got[k].cpu().numpy(), g[f"item{k}/{key}"], equal_nan=True), (k, key)
            assert np.array_equal(pb.ms2_cond[k].cpu().numpy(), g[f"item{k}/cond_{w[0]}_{w[1]}"], equal_nan=True), k
    # constant pair -> NaN exactly like the reference (no epsilon in the min-max)
    pb = ld.form([5], [6])
    for got, key in zip(pb, KEYS):
        assert np.array_equal(got[0].cpu().numpy(), g[f"const/{key}"], equal_nan=True), key
    # iterating consumes the reference's `random` stream: same pairs, same items, epoch = len(dataset) pairs
    ld = ResidentPairLoader(ds, batch_size=3, device="cuda")
    assert len(ld) == 3  # ceil(7 / 3)
    ds.reset_epoch()
    random.seed(int(g["seed"]))
    k = 0
    for batch in ld:
        ms2_1, ms1_1, ms2_2, ms1_2 = batch
        for r in range(ms2_1.shape[0]):
            assert np.array_equal(ms2_1[r].cpu().numpy(), g[f"item{k}/ms2_1"], equal_nan=True), k
            assert np.array_equal(ms1_2[r].cpu().numpy(), g[f"item{k}/ms1_2"], equal_nan=True), k
            k += 1
    assert k == 7


@pytest.mark.gpu
def test_pair_batch_full_size_vs_oracle_and_bad_index():
    from dquartic import _native as N

    rng = np.random.default_rng(3)
    n, RT, MZ, B = 40, 400, 64, 32
    ms2 = (rng.lognormal(0, 1, (n, RT, MZ)) * 50).astype(np.float32)
    ms1 = (rng.lognormal(0, 1, (n, RT)) * 500).astype(np.float32)
    i1, i2 = rng.integers(0, n, B), rng.integers(0, n, B)
    ref = O.pair_batch(ms2, ms1, i1, i2, (0.5, 0.5))
    d2, d1 = torch.from_numpy(ms2).cuda(), torch.from_numpy(ms1).cuda()
    idx = torch.from_numpy(np.concatenate([i1, i2]).astype(np.int64)).cuda()
    new = lambda *s: torch.empty(s, device="cuda")
    a, b, c, m1, m2 = new(B, RT, MZ), new(B, RT, MZ), new(B, RT, MZ), new(B, RT), new(B, RT)
    sc = torch.empty(N.lib().dq_pair_batch_scratch_bytes(B) // 4, device="cuda")

    def run(ix):
        N.check(N.lib().dq_pair_batch(N.ptr(d2), N.ptr(d1), n, N.ptr(ix), B, RT, MZ, RT, 0.5, 0.5, N.ptr(a), N.ptr(m1), N.ptr(b), N.ptr(m2),
                                      N.ptr(c), N.ptr(sc), sc.numel() * 4, N.stream_ptr()), "dq_pair_batch")
        torch.cuda.synchronize()

    run(idx)
    for got, want in zip((a, m1, b, m2, c), ref):
        assert np.array_equal(got.cpu().numpy(), want)  # bit-exact fp32
    # size-independent properties: each pair spans exactly [0, 1]; window 1's MS1 spans [0, 1]
    lo = torch.minimum(a.flatten(1).min(1).values, b.flatten(1).min(1).values)
    hi = torch.maximum(a.flatten(1).max(1).values, b.flatten(1).max(1).values)
    assert torch.all(lo == 0) and torch.all(hi == 1) and torch.all(m1.min(1).values == 0) and torch.all(m1.max(1).values == 1)
    # an out-of-range index is never dereferenced: that pair is NaN, the others are untouched
    bad = idx.clone()
    bad[3] = n
    bad[B + 7] = -1
    run(bad)
    assert torch.isnan(a[3]).all() and torch.isnan(c[3]).all() and torch.isnan(m1[3]).all()
    assert torch.isnan(b[7]).all() and torch.isnan(m2[7]).all()
    keep = [r for r in range(B) if r not in (3, 7)]
    assert np.array_equal(a[keep].cpu().numpy(), ref[0][keep]) and np.array_equal(c[keep].cpu().numpy(), ref[4][keep])
    # too-small scratch is rejected with a message
    rc = N.lib().dq_pair_batch(N.ptr(d2), N.ptr(d1), n, N.ptr(idx), B, RT, MZ, RT, 0.5, 0.5, N.ptr(a), N.ptr(m1), None, None, None,
                               N.ptr(sc), 8, N.stream_ptr())
    assert rc != 0 and b"scratch" in N.lib().dq_last_error()


@pytest.mark.gpu
def test_train_one_epoch_uses_resident_mixture(golden, tmp_path):
    """_train_one_epoch fed by the resident loader: target = ms2_1, MS1 cond = ms1_1, mixture from the kernel."""
    from dquartic.model.model import DDIMDiffusionModel
    from dquartic.model.unet1d import UNet1d
    from dquartic.utils.data_loader import DIAMSDataset, ResidentPairLoader

    g = golden("pairs.npz")
    f2, f1 = _write(tmp_path, g)
    ds = DIAMSDataset(ms2_file=f2, ms1_file=f1, normalize="minmax")
    net = UNet1d(dim=4, channels=1, dim_mults=(1, 2), conditional=True, init_cond_channels=1, attn_cond_channels=1,
                 downsample_dim=8, simple=True).cuda()
    seen = []

    class Cap(DDIMDiffusionModel):
        def _train_one_batch(self, x_0, ms2_cond=None, ms1_cond=None, noise=None, ms1_loss_weight=0.0, **kw):
            seen.append((x_0.cpu().numpy(), ms2_cond.cpu().numpy(), ms1_cond.cpu().numpy()))
            return 0.0

    dm = Cap(model_class=net, device="cuda")
    random.seed(int(g["seed"]))
    dm._train_one_epoch(0, ResidentPairLoader(ds, batch_size=2, device="cuda", drop_last=True))
    assert len(seen) == 3
    k = 0
    for x0, c2, c1 in seen:
        for r in range(2):
            assert np.array_equal(x0[r], g[f"item{k}/ms2_1"], equal_nan=True)
            assert np.array_equal(c2[r], g[f"item{k}/cond_0.5_0.5"], equal_nan=True)
            assert np.array_equal(c1[r], g[f"item{k}/ms1_1"], equal_nan=True)
            k += 1
