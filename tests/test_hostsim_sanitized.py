"""The solver cores (hs_solver.h, os_solver.h: the source the GPU lane kernels are compiled from) under AddressSanitizer +
UndefinedBehaviorSanitizer on the host, with the scratch arrays NaN-poisoned (the device scratch is not zero-initialised,
the host twin's std::vector is): no diagnostic, and the poisoned run takes exactly the path of the clean one -- i.e. no
out-of-bounds access, no undefined arithmetic and no read-before-write in the shared source (the
GPU-only lane-position failures were suspected to be UB; they are not reproducible on the current build either, with
the penalty floor MYR_TRAP_LAM_FLOOR switched on for the trapezoidal core)."""
import os
import subprocess
import sys
import tempfile

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))

DRIVER = r'''
import ctypes as C, os, sys, numpy as np
lib = C.CDLL(sys.argv[1]); dp = C.c_void_p
lib.hostsim_solve_trap.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp, dp, dp, dp]
lib.hostsim_solve_shoot.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, dp, dp, dp, dp, C.c_int, C.c_int, dp, dp, dp, dp, dp]
A = lambda a: a.ctypes.data
B, N = 6, 40
rng = np.random.default_rng(5)
x0 = np.clip(0.1 * rng.standard_normal((B, 4)), -2, 2)
xT = np.array([1.0, np.pi, 0.0, 0.0]); bnd = np.array([[-2, 2], [-2 * np.pi, 2 * np.pi], [-5, 5], [-10, 10], [-20, 20.0]])
lin = np.linspace(0, 1, N + 1)[None, :, None]
z = np.ascontiguousarray(np.concatenate([(x0[:, None, :] * (1 - lin) + xT[None, None, :] * lin).reshape(B, -1), np.zeros((B, N + 1))], 1))
lb = np.empty_like(z); ub = np.empty_like(z)
lb[:, :(N + 1) * 4] = np.tile(bnd[:4, 0], N + 1); ub[:, :(N + 1) * 4] = np.tile(bnd[:4, 1], N + 1)
lb[:, (N + 1) * 4:] = bnd[4, 0]; ub[:, (N + 1) * 4:] = bnd[4, 1]
lb[:, :4] = x0; ub[:, :4] = x0; lb[:, N * 4:(N + 1) * 4] = xT; ub[:, N * 4:(N + 1) * 4] = xT
lam = np.zeros((B, N * 4)); cost = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32); kkt = np.zeros((B, 3))
os.environ["DWARM"] = "1"
lib.hostsim_solve_trap(0, N, 2.0, B, A(z), A(lb), A(ub), None, 0, 1000, A(lam), A(cost), A(st), A(it), A(kkt))
print("TRAP", st.tolist(), it.tolist(), ["%.12e" % c for c in cost])
# VANDERPOL single shooting 1 x 20, Heun
B = 4
x0 = np.clip(np.array([0., 1.]) + 0.1 * rng.standard_normal((B, 2)), -4, 4); cpi = 20
z = np.ascontiguousarray(np.concatenate([np.stack([x0, np.zeros((B, 2))], 1).reshape(B, -1), np.zeros((B, cpi + 1))], 1))
lb = np.empty_like(z); ub = np.empty_like(z)
lb[:, :4] = np.tile([-4.0, -4.0], 2); ub[:, :4] = np.tile([4.0, 4.0], 2); lb[:, 4:] = -0.75; ub[:, 4:] = 1.0
lb[:, :2] = x0; ub[:, :2] = x0; lb[:, 2:4] = 0.0; ub[:, 2:4] = 0.0
lam = np.zeros((B, 2)); cost = np.zeros(B); st = np.zeros(B, np.int32); it = np.zeros(B, np.int32)
lib.hostsim_solve_shoot(1, 1, cpi, 1, 10.0, B, A(z), A(lb), A(ub), None, 0, 300, A(lam), A(cost), A(st), A(it), None)
print("SHOOT", st.tolist(), it.tolist(), ["%.12e" % c for c in cost])
'''


def test_solver_cores_are_clean_under_asan_ubsan_with_poisoned_scratch():
  asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
  if not os.path.isabs(asan) or not os.path.exists(asan):
    pytest.skip("no libasan in this toolchain")
  with tempfile.TemporaryDirectory() as td:
    so = os.path.join(td, "libhostsim_san.so")
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fopenmp", "-fPIC", "-shared", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
                    "-DMYR_TRAP_LAM_FLOOR=true", os.path.join(HERE, "hostsim", "hostsim.cpp"), "-o", so], check=True)
    drv = os.path.join(td, "drv.py")
    open(drv, "w").write(DRIVER)
    outs = []
    for poison in ("", "1"):
      env = dict(os.environ, LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="4")
      env.pop("POISON", None)
      if poison:
        env["POISON"] = "1"
      r = subprocess.run([sys.executable, drv, so], capture_output=True, text=True, env=env, timeout=900)
      assert r.returncode == 0, r.stderr[-3000:]
      assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
      lines = [l for l in r.stdout.splitlines() if l.startswith(("TRAP", "SHOOT"))]
      assert len(lines) == 2, r.stdout
      outs.append(lines)
    assert outs[0] == outs[1], (outs[0], outs[1])          # NaN-poisoned scratch changes nothing: no read-before-write
    assert "TRAP [0, 0, 0, 0, 0, 0]" in outs[0][0], outs[0][0]
