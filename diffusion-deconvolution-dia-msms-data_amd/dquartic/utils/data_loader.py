"""``DIAMSDataset`` -- same contract as the reference's ``dquartic/utils/data_loader.py`` (:10-185): ``__getitem__`` ignores
its index, draws a random PAIR of distinct windows not yet used this epoch, min-max normalises the pair (MS2 over both
windows, MS1 from window 1 only; :70-79) and returns ``(ms2_1, ms1_1, ms2_2, ms1_2)`` as float32 tensors;
``reset_epoch()`` clears the used-pair set.  Backends: two ``.npy`` files (mmap) or a directory of parquet slices with the
schema the reference's ETL writes (data_generation.py:206-223).  The parquet backend uses pyarrow (duckdb is not a
dependency here); slices are addressed by (file, row) instead of a six-column SQL filter, same result."""
import glob
import os
import random
from typing import Literal

import numpy as np
import torch
from torch.utils.data import Dataset


class DIAMSDataset(Dataset):
    def __init__(self, parquet_directory=None, ms2_file=None, ms1_file=None, normalize: Literal[None, "minmax"] = None):
        if parquet_directory is None and ms1_file is not None and ms2_file is not None:
            self.ms2_data = np.load(ms2_file, mmap_mode="r")
            self.ms1_data = np.load(ms1_file, mmap_mode="r")
            self.data_type = "npy"
            print(f"Info: Loaded  {len(self.ms2_data)} MS2 slice samples and {len(self.ms1_data)} MS1 slice samples from NPY files.")
        elif parquet_directory is not None and ms1_file is None and ms2_file is None:
            self.parquet_directory = parquet_directory
            self.meta = self.read_parquet_meta(parquet_directory)
            self.data_type = "parquet"
            print(f"Info: Loaded {len(self.meta)} MS2 slice samples and MS1 slice samples from Parquet files.")
        else:
            raise ValueError("Invalid input data arguments. Please provide either a `parquet_directory` or `ms2_file` and `ms1_file`. "
                             f"Got parquet_directory={parquet_directory}, ms2_file={ms2_file}, ms1_file={ms1_file}.")
        self.normalize = normalize
        self.used_pairs = set()
        self.epoch_reset = False

    def __len__(self):
        return len(self.meta) if self.data_type == "parquet" else len(self.ms2_data)

    def reset_epoch(self):
        self.used_pairs.clear()
        self.epoch_reset = True

    # ---- parquet backend
    def read_parquet_meta(self, parquet_directory):
        import pyarrow.parquet as pq

        meta = []
        for path in sorted(glob.glob(os.path.join(parquet_directory, "*.parquet"))):
            t = pq.read_table(path, columns=["slice_index", "mz_isolation_target"])
            si, mt = t.column("slice_index").to_pylist(), t.column("mz_isolation_target").to_pylist()
            meta.extend((path, i, si[i], mt[i]) for i in range(len(si)))
        return meta

    def _get_parquet_data(self, entry):
        import pyarrow.parquet as pq

        path, row, _, _ = entry
        t = pq.read_table(path, columns=["ms2_data", "ms1_data", "ms2_shape", "ms1_shape"]).slice(row, 1).to_pylist()[0]
        ms2 = np.asarray(t["ms2_data"], dtype=np.float32).reshape(t["ms2_shape"])
        ms1 = np.asarray(t["ms1_data"], dtype=np.float32).reshape(t["ms1_shape"])
        return ms1, ms2

    # ---- pair sampling
    def _draw_pair(self, n, same=None):
        if n < 2:
            raise ValueError("DIAMSDataset needs at least two windows to form a pair")
        if len(self.used_pairs) >= n * (n - 1) // 2:
            self.used_pairs.clear()  # every pair used: start over instead of spinning forever
        while True:
            i, j = random.randint(0, n - 1), random.randint(0, n - 1)
            if i == j or (same is not None and same(i, j)):
                continue
            pair = (min(i, j), max(i, j))
            if pair in self.used_pairs:
                continue
            self.used_pairs.add(pair)
            return i, j

    def __getitem__(self, idx):
        if self.data_type == "npy":
            i, j = self._draw_pair(len(self.ms2_data))
            ms2_1, ms1_1, ms2_2, ms1_2 = self.ms2_data[i], self.ms1_data[i], self.ms2_data[j], self.ms1_data[j]
        else:
            same = lambda a, b: self.meta[a][2] == self.meta[b][2] and self.meta[a][3] == self.meta[b][3]
            i, j = self._draw_pair(len(self.meta), same)
            ms1_1, ms2_1 = self._get_parquet_data(self.meta[i])
            ms1_2, ms2_2 = self._get_parquet_data(self.meta[j])
        if self.normalize == "minmax":
            lo, hi = min(ms2_1.min(), ms2_2.min()), max(ms2_1.max(), ms2_2.max())
            lo1, hi1 = ms1_1.min(), ms1_1.max()
            ms2_1, ms2_2 = (ms2_1 - lo) / (hi - lo), (ms2_2 - lo) / (hi - lo)
            ms1_1, ms1_2 = (ms1_1 - lo1) / (hi1 - lo1), (ms1_2 - lo1) / (hi1 - lo1)
        else:
            raise ValueError("Invalid normalization method. Valid options are: None, 'minmax'.")
        return tuple(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)) for v in (ms2_1, ms1_1, ms2_2, ms1_2))
