#!/usr/bin/env python3
"""Fixture generator for the batch-formation step.  TEST INFRASTRUCTURE, runs only in the build container.

Imports the reference's ``dquartic/utils/data_loader.py`` (``DIAMSDataset``, npy backend, ``normalize="minmax"``) from
/root/reference and records, for small float32 arrays written to a temp dir, what ``__getitem__`` returns for a seeded
``random`` stream, plus the mixture the reference's ``_train_one_epoch`` forms from it (model_interface.py:1073-1075).
``duckdb`` (absent here) is imported at module level by the reference but used only by the parquet backend: an EMPTY
stand-in module lets the import succeed; no arithmetic goes through it.  Output: tests/golden/pairs.npz (raw arrays,
seed, drawn index pairs, normalised 4-tuples, mixtures; one constant pair -> NaN case)."""
import os
import random
import sys
import tempfile

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def main():
    stubs = tempfile.mkdtemp(prefix="dq_stubs_")
    open(os.path.join(stubs, "duckdb.py"), "w").write("# empty stand-in: only the parquet backend uses duckdb\n")
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, stubs)
    from dquartic.utils.data_loader import DIAMSDataset  # the reference's

    rng = np.random.default_rng(42)
    N, RT, MZ = 7, 12, 8
    ms2 = (rng.lognormal(0.0, 1.0, (N, RT, MZ)) * 100.0).astype(np.float32)
    ms1 = (rng.lognormal(0.0, 1.0, (N, RT)) * 1000.0).astype(np.float32)
    ms2[5] = 3.0  # windows 5 and 6 are constant and equal: the pair (5,6) normalises to 0/0 = NaN like the reference
    ms2[6] = 3.0
    tmp = tempfile.mkdtemp(prefix="dq_pairs_")
    np.save(os.path.join(tmp, "ms2.npy"), ms2)
    np.save(os.path.join(tmp, "ms1.npy"), ms1)
    ds = DIAMSDataset(ms2_file=os.path.join(tmp, "ms2.npy"), ms1_file=os.path.join(tmp, "ms1.npy"), normalize="minmax")
    out = {"ms2": ms2, "ms1": ms1, "seed": np.int64(9)}
    K = 12
    random.seed(9)
    items = [ds[0] for _ in range(K)]
    # replay the index stream the reference consumed (data_loader.py:118-131): two randint per attempt, retry on i == j / used pair
    random.seed(9)
    used, pairs = set(), []
    while len(pairs) < K:
        i, j = random.randint(0, N - 1), random.randint(0, N - 1)
        if i == j or tuple(sorted((i, j))) in used:
            continue
        used.add(tuple(sorted((i, j))))
        pairs.append((i, j))
    out["pairs"] = np.asarray(pairs, np.int64)
    for k, (a, m1, b, m2) in enumerate(items):
        assert a.dtype == torch.float32
        out[f"item{k}/ms2_1"], out[f"item{k}/ms1_1"], out[f"item{k}/ms2_2"], out[f"item{k}/ms1_2"] = a.numpy(), m1.numpy(), b.numpy(), m2.numpy()
        for w in ((0.5, 0.5), (0.7, 0.3)):
            out[f"item{k}/cond_{w[0]}_{w[1]}"] = ((a * w[0]) + (b * w[1])).numpy()  # model_interface.py:1073-1075
    # the constant pair explicitly (whatever the random stream drew)
    ds.used_pairs.clear()
    import unittest.mock as um
    with um.patch("random.randint", side_effect=[5, 6]):
        a, m1, b, m2 = ds[0]
    out["const/ms2_1"], out["const/ms1_1"], out["const/ms2_2"], out["const/ms1_2"] = a.numpy(), m1.numpy(), b.numpy(), m2.numpy()
    assert np.isnan(out["const/ms2_1"]).all()
    np.savez_compressed(os.path.join(REPO, "tests", "golden", "pairs.npz"), **out)
    # oracle cross-check
    sys.path.insert(0, REPO)
    from oracle import dq_oracle as O
    for k, (i, j) in enumerate(pairs):
        a, m1, b, m2, c = O.pair_batch(ms2, ms1, [i], [j], (0.7, 0.3))
        for got, key in ((a, "ms2_1"), (m1, "ms1_1"), (b, "ms2_2"), (m2, "ms1_2"), (c, "cond_0.7_0.3")):
            assert np.array_equal(got[0], out[f"item{k}/{key}"], equal_nan=True), (k, key)
    print("pairs.npz written;", K, "items, pairs", pairs)


if __name__ == "__main__":
    main()
