import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
  return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _torch_hip_first():
  """On a GPU box initialise torch's HIP runtime BEFORE libmyriad_hip.so touches the device: torch ships its own ROCm
  runtime, and when the system runtime (loaded by the library) initialises first, torch later reports
  'No HIP GPUs are available'.  bench.py and smoke() do the same by calling torch.cuda first."""
  try:
    import torch
    if torch.cuda.is_available():
      torch.cuda.init()
  except Exception:
    pass
  yield
