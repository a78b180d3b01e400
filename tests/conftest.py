import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# kzg_msm_g1_batch uses its dedicated accumulation streams only when every stream can have a hardware queue of its own
# (kzg_amd/csrc/capi.hip); the HIP runtime reads this when it initialises, so it has to be set before the first HIP call.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
