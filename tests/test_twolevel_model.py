"""The interface algebra of the two-level Riccati sweep (hs_solver_fused.h: riccati_chunk / tl_join / tl_theta), restated in numpy
(tools/dev/twolevel/model.py), against the plain backward recursion on random staged QPs: same value form at the first point, same
terminal multipliers, same states and controls -- for 2 and 4 chunks, chunks of unequal length, pinned and free terminal states,
convex and (mildly) indefinite stage Hessians.  CPU only; the device kernels are held to the one-wavefront kernel in tests/test_gpu_*.py."""
import importlib.util, os
import numpy as np
import pytest

_spec = importlib.util.spec_from_file_location("twolevel_model", os.path.join(os.path.dirname(__file__), "..", "tools", "dev", "twolevel", "model.py"))
model = importlib.util.module_from_spec(_spec); _spec.loader.exec_module(model)


@pytest.mark.parametrize("convex", [True, False])
@pytest.mark.parametrize("N,W", [(24, 2), (24, 4), (10, 4), (7, 2), (5, 4), (4, 4)])
@pytest.mark.parametrize("pinned", [(0, 1, 3), (), (0, 1, 2, 3)])
def test_two_level_equals_plain_recursion(N, W, pinned, convex):
  r = model.compare(N=N, W=W, pinned=pinned, convex=convex)
  assert r["pmin_seq"] > 0 and r["pmin_chunks"] > 0 and r["pmin_if"] > 0          # a positive definite problem shows positive pivots at every level
  tol = 2e-8 * r["scale"]                                                          # (rho = 1e4 costs about four digits in the interface solve)
  assert r["dw"] <= tol and r["dq"] <= tol and r["dnu"] <= tol * 10, r
  assert r["pinned"] <= 1e-12 * r["scale"], r
