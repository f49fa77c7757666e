"""CPU tests of the C-ABI surface: the library loads, exports every symbol include/myriad_hip.h declares,
and argument validation works without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from myriad_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
  if not os.path.exists(_lib.LIB_PATH):
    import __graft_entry__ as g
    g.build()
  return _lib.load()


def test_exports_match_header(lib):
  hdr = open(os.path.join(ROOT, "include", "myriad_hip.h")).read()
  declared = set(re.findall(r"\b(myr_[a-z0-9_]+)\s*\(", hdr))
  assert declared == set(_lib.EXPORTS)
  for s in declared:
    assert hasattr(lib, s), s


def test_struct_layouts_match_header():
  assert C.sizeof(_lib.ProblemDesc) == 8 * 4 + 8
  assert C.sizeof(_lib.Dims) == 10 * 4
  assert C.sizeof(_lib.SolveOpts) == 2 * 4 + 4 * 8 + 2 * 4          # (round 4: + restoration, park_iter)


def test_default_opts(lib):
  o = _lib.SolveOpts()
  lib.myr_default_solve_opts(C.byref(o))
  assert o.max_iter == 1000 and o.tol_feas == 1e-8 and o.tol_stat == 1e-6 and o.restoration == -1


def test_create_rejects_bad_arguments(lib):
  h = C.c_void_p()
  d = _lib.ProblemDesc(99, 0, 1, 10, 1, 0, 16, 0, 2.0)
  assert lib.myr_create(C.byref(d), C.byref(h)) == -1
  assert b"system_id" in lib.myr_last_error()
  d = _lib.ProblemDesc(0, 0, 1, 0, 1, 0, 16, 0, 2.0)
  assert lib.myr_create(C.byref(d), C.byref(h)) == -1
  d = _lib.ProblemDesc(0, 7, 1, 10, 1, 0, 16, 0, 2.0)
  assert lib.myr_create(C.byref(d), C.byref(h)) == -1
  assert lib.myr_create(None, C.byref(h)) == -1


def test_engine_fails_loudly_without_gpu():
  import torch
  if torch.cuda.is_available():
    pytest.skip("GPU present")
  with pytest.raises((_lib.MyriadHipError, ValueError)):
    _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", 10, 2.0)
