import os
import sys

import numpy as np
import pytest
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "diffusion-deconvolution-dia-msms-data_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")

# No environment default is set here: the suite runs the library the way bench.py runs it.  Tests that must see BOTH LinearAttention forms
# (register-resident k_linattn.hip / k_la_bwd.hip against the one-register-group-per-position k_la_small.hip / k_la_rows_bwd.hip, chosen by row
# count in the product) take the `la_form` fixture below, which flips the library's own tuning option (dq_set_option) in-process.
LA_FORMS = {
    # name: (la_small_min_rows, la_rows_bwd_min_rows); -1 = the library's default rule
    "default": (-1, -1),          # what bench.py / a user gets: by row count (forward), every row count (backward)
    "rows": (0, 0),               # the per-row forms at every row count
    "register": (1 << 40, 1 << 40),  # the register-resident forms at every row count
}


def set_la_form(name):
    from dquartic import _native as N

    fwd, bwd = LA_FORMS[name]
    N.set_option("la_small_min_rows", fwd)
    N.set_option("la_rows_bwd_min_rows", bwd)


@pytest.fixture(params=["default", "rows", "register"])
def la_form(request):
    """Runs the test once per LinearAttention dispatch; restores the default rule afterwards."""
    set_la_form(request.param)
    yield request.param
    set_la_form("default")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name))
    return {k: z[k] for k in z.files}


def sub(d, prefix):
    """Entries of ``d`` under ``prefix`` as torch tensors, prefix stripped."""
    return {k[len(prefix):]: torch.from_numpy(np.array(v)) for k, v in d.items() if k.startswith(prefix)}


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def get(name):
        if name not in cache:
            cache[name] = load_golden(name)
        return cache[name]

    return get
