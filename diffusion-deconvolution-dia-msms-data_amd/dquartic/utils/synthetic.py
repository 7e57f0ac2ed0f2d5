"""Synthetic multiplexed-MS2 windows.  The reference ships no generator (its data come from
instrument files through an offline ETL); this one exists so that the hot path can be benchmarked and tested on data of
the right shape and statistics.

Window i (``rng = np.random.default_rng(1234 + i)``): K ~ U{3..12} co-eluting "peptides", each with an RT apex
~ U(0, RT), an elution width sigma ~ U(3, 15), F ~ U{4..12} fragment m/z bins ~ U{0..MZ-1} with LogNormal(0,1)
intensities; ``ms2[rt, mz] = sum_f I_f * exp(-(rt - apex)^2 / (2 sigma^2))`` and the MS1 chromatogram is the
precursor trace ``ms1[rt] = sum_k I_k * exp(...)``.  ``SyntheticDIAMSDataset`` then follows the reference dataset
contract (data_loader.py:60-90): ``__getitem__`` ignores its index, draws a random PAIR of windows, min-max
normalises the pair (MS2: min/max over both windows; MS1: min/max of window 1, applied to both) and returns
``(ms2_1, ms1_1, ms2_2, ms1_2)``; ``reset_epoch()`` exists because the harness calls it.
"""
import numpy as np
import torch
from torch.utils.data import Dataset


def make_window(i: int, RT: int = 400, MZ: int = 64):
    rng = np.random.default_rng(1234 + i)
    rt = np.arange(RT, dtype=np.float32)[:, None]
    ms2 = np.zeros((RT, MZ), np.float32)
    ms1 = np.zeros((RT,), np.float32)
    for _ in range(int(rng.integers(3, 13))):
        apex = rng.uniform(0, RT)
        sigma = rng.uniform(3, 15)
        prof = np.exp(-((rt - apex) ** 2) / (2 * sigma * sigma)).astype(np.float32)  # (RT, 1)
        nfrag = int(rng.integers(4, 13))
        bins = rng.integers(0, MZ, size=nfrag)
        inten = rng.lognormal(0.0, 1.0, size=nfrag).astype(np.float32)
        np.add.at(ms2, (slice(None), bins), prof * inten[None, :])
        ms1 += float(rng.lognormal(0.0, 1.0)) * prof[:, 0]
    return ms2, ms1


def make_pool(n: int, RT: int = 400, MZ: int = 64, start: int = 0):
    ms2 = np.empty((n, RT, MZ), np.float32)
    ms1 = np.empty((n, RT), np.float32)
    for i in range(n):
        ms2[i], ms1[i] = make_window(start + i, RT, MZ)
    return ms2, ms1


def normalize_pair(ms2_1, ms1_1, ms2_2, ms1_2):
    """Per-pair min-max exactly as data_loader.py:70-79 (MS1 statistics come from window 1 only)."""
    lo, hi = min(ms2_1.min(), ms2_2.min()), max(ms2_1.max(), ms2_2.max())
    ms2_1, ms2_2 = (ms2_1 - lo) / (hi - lo), (ms2_2 - lo) / (hi - lo)
    lo1, hi1 = ms1_1.min(), ms1_1.max()
    ms1_1, ms1_2 = (ms1_1 - lo1) / (hi1 - lo1), (ms1_2 - lo1) / (hi1 - lo1)
    return ms2_1, ms1_1, ms2_2, ms1_2


class SyntheticDIAMSDataset(Dataset):
    def __init__(self, n_windows: int = 32, RT: int = 400, MZ: int = 64, normalize="minmax", seed: int = 0, rank: int = 0, world: int = 1):
        if normalize is None:
            raise ValueError("normalize must be 'minmax' (the reference raises on None, data_loader.py:80-81)")
        # rank r of `world` owns windows i with i % world == r
        ids = [i for i in range(n_windows) if i % world == rank]
        self.ms2 = np.stack([make_window(i, RT, MZ)[0] for i in ids])
        self.ms1 = np.stack([make_window(i, RT, MZ)[1] for i in ids])
        self.normalize = normalize
        self._rng = np.random.default_rng(seed + 7919 * rank)

    def __len__(self):
        return len(self.ms2)

    def reset_epoch(self):
        pass

    def __getitem__(self, idx):
        a, b = self._rng.choice(len(self.ms2), size=2, replace=len(self.ms2) < 2)
        out = normalize_pair(self.ms2[a], self.ms1[a], self.ms2[b], self.ms1[b])
        return tuple(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for v in out)
