import os
import shutil
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The CPU oracle is test infrastructure; build it if this checkout has no copy yet.
    if not os.path.exists(os.path.join(ROOT, "oracle", "_build", "liboracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    # The product library is built in-tree by __graft_entry__.build(); a checkout that has not run it
    # yet gets it here if hipcc is around (the tests themselves never build anything on a GPU box).
    lib = os.path.join(ROOT, "gptq_v2_amd", "lib", "libgptq_amd.so")
    if not os.path.exists(lib) and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "gptq_v2_amd", "build.py")])


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: d[k] for k in d.files}


def t(a, dtype=None):
    """numpy fixture array -> torch tensor (bf16 fixtures are stored as raw uint16 words)."""
    x = torch.from_numpy(np.ascontiguousarray(a))
    if dtype == torch.bfloat16:
        return x.view(torch.int16).view(torch.bfloat16) if x.dtype == torch.uint16 else x.view(torch.bfloat16)
    return x if dtype is None else x.to(dtype)


@pytest.fixture(scope="session")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def amd(gpu):
    """The product package; importing it loads libgptq_amd.so (GPU tests only)."""
    import gptq_v2_amd
    return gptq_v2_amd


@pytest.fixture(scope="session")
def oracle():
    """The CPU restatement, as the checker."""
    from oracle import gptq_oracle
    return gptq_oracle
