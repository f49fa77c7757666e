"""ctypes binding of libmyriad_hip.so (C-ABI: include/myriad_hip.h).

The product path has no CPU fallback: if the HIP library is missing or no MI355X is visible, the
calls raise -- they never route to the oracle or to any host implementation.
"""
from __future__ import annotations

import ctypes as C
import os
import sys
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MYRIAD_HIP_LIB") or os.path.join(_HERE, "libmyriad_hip.so")   # override: development builds

SYS_IDS = {"CARTPOLE": 0, "VANDERPOL": 1, "CANCERTREATMENT": 2, "SIMPLECASE": 3, "NODE_CARTPOLE": 4, "BIOREACTOR": 5,
           "GLUCOSE": 6, "MOULDFUNGICIDE": 7, "SIMPLECASEWITHBOUNDS": 8, "HIVTREATMENT": 9, "EPIDEMICSEIRN": 10, "SEIR": 11,
           "BEARPOPULATIONS": 12, "PENDULUM": 13, "MOUNTAINCAR": 14, "ROCKETLANDING": 15,
           "BACTERIA": 16, "TUMOUR": 17, "HARVEST": 18, "TIMBERHARVEST": 19, "PREDATORPREY": 20,
           "INVASIVEPLANT": 21,   # INVASIVEPLANT: discrete-time, myr_fbsm only
           "PENDULUM_ELASTIC": 113, "ROCKETLANDING_ELASTIC": 115, "CARTPOLE_ELASTIC": 100, "VANDERPOL_ELASTIC": 101, "MOUNTAINCAR_ELASTIC": 114}   # elastic twins (slack controls on the dynamics), see include/myriad_hip.h
TR_IDS = {"HERMITE_SIMPSON": 0, "TRAPEZOIDAL": 1, "SHOOTING": 2}
INT_IDS = {"EULER": 0, "HEUN": 1, "MIDPOINT": 2, "RK4": 3}
MEM_HOST, MEM_DEVICE = 0, 1
K_EVAL, K_SOLVE, K_ROLLOUT, K_RESID, K_PROD, K_FBSM = 0, 1, 2, 3, 4, 5
STATUS_NAMES = {0: "CONVERGED", 1: "MAXITER", 2: "NAN", 3: "STALLED", 4: "INFEASIBLE"}
STATUS_INFEASIBLE = 4   # assigned by the library's restoration phase inside myr_solve (csrc/myriad_hip.hip: solve_restored), never by a kernel

EXPORTS = ["myr_create", "myr_destroy", "myr_get_dims", "myr_default_solve_opts", "myr_eval", "myr_solve", "myr_solve_x0",
           "myr_set_var_scale", "myr_rollout", "myr_vjp", "myr_jvp", "myr_exgd", "myr_fbsm", "myr_kernel_time", "myr_kernel_time_reset", "myr_last_error",
           "myr_version", "myr_device_count", "myr_solve_info", "myr_abi_sizeof", "myr_solve_plan"]


class ProblemDesc(C.Structure):
  _fields_ = [("system_id", C.c_int32), ("transcription", C.c_int32), ("integration_method", C.c_int32),
              ("intervals", C.c_int32), ("controls_per_interval", C.c_int32), ("device", C.c_int32),
              ("max_batch", C.c_int32), ("reserved", C.c_int32), ("T", C.c_double)]


class Dims(C.Structure):
  _fields_ = [("n", C.c_int32), ("m", C.c_int32), ("ns", C.c_int32), ("nu", C.c_int32), ("np", C.c_int32),
              ("x_rows", C.c_int32), ("u_rows", C.c_int32), ("jblk", C.c_int32), ("ngrad", C.c_int32),
              ("reserved", C.c_int32)]


class SolveOpts(C.Structure):
  _fields_ = [("max_iter", C.c_int32), ("restarts", C.c_int32), ("tol_feas", C.c_double),
              ("tol_stat", C.c_double), ("tol_compl", C.c_double), ("mu_init", C.c_double),
              ("restoration", C.c_int32), ("park_iter", C.c_int32)]


class MyriadHipError(RuntimeError):
  pass


_lib = None


def load() -> C.CDLL:
  """Load the HIP library; raise loudly when it has not been built (python __graft_entry__.py build)."""
  global _lib
  if _lib is not None:
    return _lib
  if not os.path.exists(LIB_PATH):
    raise MyriadHipError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`. "
                         "There is no CPU fallback.")
  # PyTorch-ROCm ships its own HIP runtime.  If this library brought the system runtime in first, a later torch.cuda
  # initialisation in the same process reports "No HIP GPUs are available"; loading torch's libraries first makes one
  # runtime serve both (torch is optional: nothing else here uses it).
  if "torch" not in sys.modules and os.environ.get("MYRIAD_NO_TORCH_PRELOAD") is None:
    try:
      import torch  # noqa: F401
    except Exception:
      pass
  lib = C.CDLL(LIB_PATH)
  vp, dp, ip = C.c_void_p, C.c_void_p, C.c_void_p   # raw addresses (host numpy or device pointers)
  lib.myr_create.argtypes = [C.POINTER(ProblemDesc), C.POINTER(C.c_void_p)]
  lib.myr_create.restype = C.c_int
  lib.myr_destroy.argtypes = [C.c_void_p]
  lib.myr_destroy.restype = C.c_int
  lib.myr_get_dims.argtypes = [C.c_void_p, C.POINTER(Dims)]
  lib.myr_get_dims.restype = C.c_int
  lib.myr_default_solve_opts.argtypes = [C.POINTER(SolveOpts)]
  lib.myr_default_solve_opts.restype = None
  lib.myr_device_count.argtypes = []
  lib.myr_device_count.restype = C.c_int
  lib.myr_eval.argtypes = [vp, C.c_int32, dp, dp, C.c_int32, dp, dp, dp, dp, C.c_int32]
  lib.myr_eval.restype = C.c_int
  lib.myr_solve.argtypes = [vp, C.c_int32, dp, dp, dp, dp, C.c_int32, C.POINTER(SolveOpts), dp, dp, ip, ip, dp, C.c_int32]
  lib.myr_solve.restype = C.c_int
  lib.myr_solve_x0.argtypes = [vp, C.c_int32, dp, dp, dp, dp, dp, dp, C.c_int32, C.POINTER(SolveOpts), dp, dp, dp, ip, ip, dp, C.c_int32]
  lib.myr_solve_x0.restype = C.c_int
  lib.myr_solve_info.argtypes = [vp, C.c_int32, ip, ip, ip]
  lib.myr_solve_info.restype = C.c_int
  lib.myr_solve_plan.argtypes = [vp, ip]
  lib.myr_solve_plan.restype = C.c_int
  lib.myr_rollout.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, dp, dp, dp, C.c_int32, dp, dp, C.c_int32]
  lib.myr_rollout.restype = C.c_int
  lib.myr_set_var_scale.argtypes = [vp, dp]
  lib.myr_set_var_scale.restype = C.c_int
  lib.myr_vjp.argtypes = [vp, C.c_int32, dp, dp, dp, C.c_int32, dp, C.c_int32, C.c_int32]
  lib.myr_vjp.restype = C.c_int
  lib.myr_jvp.argtypes = [vp, C.c_int32, dp, dp, dp, C.c_int32, dp, C.c_int32]
  lib.myr_jvp.restype = C.c_int
  lib.myr_exgd.argtypes = [vp, C.c_int32, dp, dp, dp, dp, dp, C.c_int32, C.c_double, C.c_double, C.c_int32, C.c_int32]
  lib.myr_exgd.restype = C.c_int
  lib.myr_fbsm.argtypes = [vp, C.c_int32, C.c_int32, dp, dp, dp, C.c_int32, dp, dp, C.c_double, C.c_double, C.c_int32,
                           dp, dp, dp, ip, C.c_int32]
  lib.myr_fbsm.restype = C.c_int
  lib.myr_kernel_time.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int32)]
  lib.myr_kernel_time.restype = C.c_int
  lib.myr_kernel_time_reset.argtypes = [vp]
  lib.myr_kernel_time_reset.restype = C.c_int
  lib.myr_last_error.restype = C.c_char_p
  lib.myr_version.restype = C.c_char_p
  # ABI guard: the structs have no size field; a library built from another header would read past (or short of) ours
  if not hasattr(lib, "myr_abi_sizeof"):
    raise MyriadHipError(f"{LIB_PATH} ({lib.myr_version().decode()}) predates myr_abi_sizeof: rebuild it (python __graft_entry__.py build)")
  lib.myr_abi_sizeof.argtypes = [C.c_int32]
  lib.myr_abi_sizeof.restype = C.c_int32
  for which, st in ((0, SolveOpts), (1, ProblemDesc), (2, Dims)):
    if lib.myr_abi_sizeof(which) != C.sizeof(st):
      raise MyriadHipError(f"{LIB_PATH} ({lib.myr_version().decode()}): {st.__name__} is {lib.myr_abi_sizeof(which)} bytes in the library, "
                           f"{C.sizeof(st)} in this binding -- library and binding come from different headers")
  _lib = lib
  return lib


def _chk(rc: int, what: str):
  if rc != 0:
    msg = load().myr_last_error().decode()
    if rc == -2:
      raise NotImplementedError(f"{what}: {msg}")
    if rc == -1:
      raise ValueError(f"{what}: {msg}")
    raise MyriadHipError(f"{what} failed ({rc}): {msg}")


def _addr(a) -> Optional[int]:
  """Address of a numpy array (host) or of anything exposing data_ptr() (torch device tensor); None -> NULL."""
  if a is None:
    return None
  if isinstance(a, np.ndarray):
    return a.ctypes.data
  if hasattr(a, "data_ptr"):
    return a.data_ptr()
  if isinstance(a, int):
    return a
  raise TypeError(type(a))


def _f64(a, shape=None):
  a = np.ascontiguousarray(a, dtype=np.float64)
  if shape is not None and tuple(a.shape) != tuple(shape):
    raise ValueError(f"expected shape {shape}, got {a.shape}")
  return a


def device_count() -> int:
  """Devices the library can create handles on (0 without a GPU)."""
  return int(load().myr_device_count())


class Engine:
  """One handle = one (system, transcription, sizes, device) problem family; owns device scratch and a stream."""

  def __init__(self, system: str, transcription: str, intervals: int, T: float, controls_per_interval: int = 1,
               integration_method: str = "HEUN", device: int = 0, max_batch: int = 4096):
    self.lib = load()
    d = ProblemDesc(SYS_IDS[system], TR_IDS[transcription], INT_IDS[integration_method], int(intervals),
                    int(controls_per_interval), int(device), int(max_batch), 0, float(T))
    self._h = C.c_void_p()
    _chk(self.lib.myr_create(C.byref(d), C.byref(self._h)), "myr_create")
    dm = Dims()
    _chk(self.lib.myr_get_dims(self._h, C.byref(dm)), "myr_get_dims")
    self.dims = dm
    self.desc = d
    for k in ("n", "m", "ns", "nu", "np", "x_rows", "u_rows", "jblk", "ngrad"):
      setattr(self, k, int(getattr(dm, k)))

  def close(self):
    if getattr(self, "_h", None) and self._h.value:
      self.lib.myr_destroy(self._h)
      self._h = C.c_void_p()

  def __del__(self):
    try:
      self.close()
    except Exception:
      pass

  # ---- host (numpy) entry points -------------------------------------------------------------
  def _params(self, params, B):
    if params is None:
      return None, 0
    p = _f64(params)
    if p.ndim == 1:
      if p.shape[0] != self.np:
        raise ValueError("params must have np entries")
      return p, 0
    if p.shape != (B, self.np):
      raise ValueError(f"params must be [B,{self.np}]")
    return p, self.np

  def eval(self, z, params=None, want=("f", "gradf", "c", "jblk")):
    z = _f64(z)
    if z.ndim == 1:
      z = z[None]
    B = z.shape[0]
    if z.shape[1] != self.n:
      raise ValueError(f"z must be [B,{self.n}]")
    p, ps = self._params(params, B)
    out = {"f": np.empty(B) if "f" in want else None,
           "gradf": np.empty((B, self.ngrad)) if "gradf" in want else None,
           "c": np.empty((B, self.m)) if "c" in want else None,
           "jblk": np.empty((B, self.jblk)) if "jblk" in want else None}
    _chk(self.lib.myr_eval(self._h, B, _addr(z), _addr(p), ps, _addr(out["f"]), _addr(out["gradf"]),
                           _addr(out["c"]), _addr(out["jblk"]), MEM_HOST), "myr_eval")
    return {k: v for k, v in out.items() if v is not None}

  def eval_device(self, B, z, params=None, params_stride=0, f=None, gradf=None, c=None, jblk=None):
    """Zero-copy variant: every argument is a device tensor (anything with data_ptr()) on this handle's device."""
    _chk(self.lib.myr_eval(self._h, int(B), _addr(z), _addr(params), int(params_stride), _addr(f), _addr(gradf),
                           _addr(c), _addr(jblk), MEM_DEVICE), "myr_eval")

  def default_opts(self) -> SolveOpts:
    o = SolveOpts()
    self.lib.myr_default_solve_opts(C.byref(o))
    return o

  def set_var_scale(self, scale=None):
    """Variable scales [ns+nu] of the solve path (None = unscaled); see include/myriad_hip.h, myr_set_var_scale."""
    if scale is None:
      _chk(self.lib.myr_set_var_scale(self._h, None), "myr_set_var_scale")
      return
    s = _f64(scale)
    if s.shape != (self.ns + self.nu,):
      raise ValueError(f"scale must have ns+nu = {self.ns + self.nu} entries")
    _chk(self.lib.myr_set_var_scale(self._h, _addr(s)), "myr_set_var_scale")

  # Result arrays of solve / solve_x0: fresh numpy arrays per call by default.  A caller that solves batch after batch sets
  # `result_buffers` to a dict of its own arrays ("z" [B,n], "lam" [B,m], "cost" [B], "status" int32 [B], "iters" int32 [B],
  # "kkt" [B,3] -- e.g. views of pinned memory); a call whose batch size matches writes its results there (no allocation, no
  # page faults) and returns views of them, valid until the next call.
  result_buffers = None

  def _results(self, B, z=None):
    rb = self.result_buffers
    if rb is not None and rb["cost"].shape[0] == B:
      if z is not None:
        rb["z"][...] = z
      return rb["z"], rb["lam"], rb["cost"], rb["status"], rb["iters"], rb["kkt"]
    return (np.empty((B, self.n)) if z is None else z, np.empty((B, self.m)), np.empty(B), np.empty(B, dtype=np.int32),
            np.empty(B, dtype=np.int32), np.empty((B, 3)))

  def solve(self, z0, lb, ub, params=None, opts: Optional[SolveOpts] = None):
    z = _f64(z0)
    if z.ndim == 1:
      z = z[None]
    B = z.shape[0]
    if not (self.result_buffers is not None and self.result_buffers["cost"].shape[0] == B):
      z = z.copy()
    lb = np.ascontiguousarray(np.broadcast_to(_f64(lb), z.shape))
    ub = np.ascontiguousarray(np.broadcast_to(_f64(ub), z.shape))
    p, ps = self._params(params, B)
    o = opts or self.default_opts()
    z, lam, cost, status, iters, kkt = self._results(B, z)
    _chk(self.lib.myr_solve(self._h, B, _addr(z), _addr(lb), _addr(ub), _addr(p), ps, C.byref(o), _addr(lam),
                            _addr(cost), _addr(status), _addr(iters), _addr(kkt), MEM_HOST), "myr_solve")
    return self._with_info({"z": z, "lam": lam, "cost": cost, "status": status, "iters": iters, "kkt": kkt}, B)

  def _with_info(self, res, B):
    """myr_solve_info: which start produced each instance (restoration inside the library, include/myriad_hip.h)"""
    start = np.empty(B, dtype=np.int32); attempts = np.empty(B, dtype=np.int32); restored = np.empty(B, dtype=np.int32)
    _chk(self.lib.myr_solve_info(self._h, B, _addr(start), _addr(attempts), _addr(restored)), "myr_solve_info")
    res["start"] = start; res["attempts"] = attempts; res["restored"] = restored
    return res

  def solve_x0(self, x0s, g0, g1, lb, ub, params=None, opts: Optional[SolveOpts] = None):
    """myr_solve_x0: the instances differ in their start state only; guess and bounds are expanded on the device
    (z0[b] = g0 + g1 * tile(x0s[b]) on the state rows, lb/ub [n] with the first point pinned to x0s[b])."""
    x0s = np.ascontiguousarray(_f64(x0s))
    if x0s.ndim == 1:
      x0s = x0s[None]
    B = x0s.shape[0]
    if x0s.shape[1] != self.ns:
      raise ValueError(f"x0s must be [B,{self.ns}]")
    tpl = [np.ascontiguousarray(_f64(a)).reshape(-1) for a in (g0, g1, lb, ub)]
    if any(a.shape[0] != self.n for a in tpl):
      raise ValueError(f"g0, g1, lb, ub must have {self.n} entries")
    p, ps = self._params(params, B)
    o = opts or self.default_opts()
    z, lam, cost, status, iters, kkt = self._results(B)
    _chk(self.lib.myr_solve_x0(self._h, B, _addr(x0s), _addr(tpl[0]), _addr(tpl[1]), _addr(tpl[2]), _addr(tpl[3]), _addr(p), ps,
                               C.byref(o), _addr(z), _addr(lam), _addr(cost), _addr(status), _addr(iters), _addr(kkt), MEM_HOST), "myr_solve_x0")
    return self._with_info({"z": z, "lam": lam, "cost": cost, "status": status, "iters": iters, "kkt": kkt}, B)

  def solve_device(self, B, z, lb, ub, params, params_stride, opts, lam, cost, status, iters, kkt=None):
    _chk(self.lib.myr_solve(self._h, int(B), _addr(z), _addr(lb), _addr(ub), _addr(params), int(params_stride),
                            C.byref(opts), _addr(lam), _addr(cost), _addr(status), _addr(iters), _addr(kkt),
                            MEM_DEVICE), "myr_solve")

  def rollout(self, x0, us, num_steps, params=None, want_xs=True):
    x0 = _f64(x0)
    us = _f64(us)
    if x0.ndim == 1:
      x0 = x0[None]
    if us.ndim == 2:
      us = us[None]
    B = x0.shape[0]
    p, ps = self._params(params, B)
    xs = np.empty((B, num_steps + 1, self.ns)) if want_xs else None
    cost = np.empty(B)
    _chk(self.lib.myr_rollout(self._h, B, int(num_steps), int(us.shape[1]), _addr(x0), _addr(us), _addr(p), ps,
                              _addr(xs), _addr(cost), MEM_HOST), "myr_rollout")
    return xs, cost

  # ---- Lagrangian products / extragradient (collocation transcriptions) ------------------------
  def _batch2(self, a, width, name):
    a = _f64(a)
    if a.ndim == 1:
      a = a[None]
    if a.shape[1] != width:
      raise ValueError(f"{name} must be [B,{width}]")
    return a

  def vjp(self, z, lam, params=None, add_gradf=False):
    """J(z)^T lam (+ grad f(z) when add_gradf: the gradient of the Lagrangian in z), [B,n]."""
    z = self._batch2(z, self.n, "z"); lam = self._batch2(lam, self.m, "lam")
    B = z.shape[0]
    if lam.shape[0] != B:
      raise ValueError("z and lam must have the same batch size")
    p, ps = self._params(params, B)
    out = np.empty((B, self.n))
    _chk(self.lib.myr_vjp(self._h, B, _addr(z), _addr(lam), _addr(p), ps, _addr(out), int(bool(add_gradf)), MEM_HOST), "myr_vjp")
    return out

  def jvp(self, z, v, params=None):
    """J(z) v, [B,m]."""
    z = self._batch2(z, self.n, "z"); v = self._batch2(v, self.n, "v")
    B = z.shape[0]
    if v.shape[0] != B:
      raise ValueError("z and v must have the same batch size")
    p, ps = self._params(params, B)
    out = np.empty((B, self.m))
    _chk(self.lib.myr_jvp(self._h, B, _addr(z), _addr(v), _addr(p), ps, _addr(out), MEM_HOST), "myr_jvp")
    return out

  def exgd(self, z, lam, lb, ub, eta_x, eta_v, nsteps, params=None):
    """`nsteps` extragradient iterations; returns the new (z, lam)."""
    z = self._batch2(z, self.n, "z").copy(); lam = self._batch2(lam, self.m, "lam").copy()
    B = z.shape[0]
    lb = np.ascontiguousarray(np.broadcast_to(_f64(lb), z.shape))
    ub = np.ascontiguousarray(np.broadcast_to(_f64(ub), z.shape))
    p, ps = self._params(params, B)
    _chk(self.lib.myr_exgd(self._h, B, _addr(z), _addr(lam), _addr(lb), _addr(ub), _addr(p), ps, float(eta_x), float(eta_v),
                           int(nsteps), MEM_HOST), "myr_exgd")
    return z, lam

  def products_device(self, op, B, z, w, out, params=None, params_stride=0, add_gradf=0):
    """Zero-copy vjp ('vjp') / jvp ('jvp') on device tensors."""
    if op == "vjp":
      _chk(self.lib.myr_vjp(self._h, int(B), _addr(z), _addr(w), _addr(params), int(params_stride), _addr(out), int(add_gradf), MEM_DEVICE), "myr_vjp")
    else:
      _chk(self.lib.myr_jvp(self._h, int(B), _addr(z), _addr(w), _addr(params), int(params_stride), _addr(out), MEM_DEVICE), "myr_jvp")

  def fbsm(self, x0, N, clip_lo, clip_hi, params=None, adj_T=None, delta=0.001, max_sweeps=10000, bang=0.0, discrete=False):
    """Batched Forward-Backward Sweep: returns {'x' [B,N+1,ns], 'u' [B,N+1,nu], 'adj' [B,N+1,ns], 'sweeps' [B]};
    for a discrete system (`discrete=True`, INVASIVEPLANT) 'u' is [B,N,nu], one row per step."""
    x0 = _f64(x0)
    if x0.ndim == 1:
      x0 = x0[None]
    B = x0.shape[0]
    p, ps = self._params(params, B)
    aT = None if adj_T is None else _f64(adj_T)
    lo = np.ascontiguousarray(np.broadcast_to(_f64(clip_lo), (self.nu,))); hi = np.ascontiguousarray(np.broadcast_to(_f64(clip_hi), (self.nu,)))
    xs = np.empty((B, N + 1, self.ns)); us = np.empty((B, N + (0 if discrete else 1), self.nu)); adjs = np.empty((B, N + 1, self.ns))
    sw = np.empty(B, dtype=np.int32)
    _chk(self.lib.myr_fbsm(self._h, B, int(N), _addr(x0), _addr(aT), _addr(p), ps, _addr(lo), _addr(hi), float(bang), float(delta),
                           int(max_sweeps), _addr(xs), _addr(us), _addr(adjs), _addr(sw), MEM_HOST), "myr_fbsm")
    return {"x": xs, "u": us, "adj": adjs, "sweeps": sw}

  def solve_plan(self) -> dict:
    """myr_solve_plan: how the library launched the first attempt of the last solve on this handle (include/myriad_hip.h)"""
    p = np.zeros(8, dtype=np.int32)
    _chk(self.lib.myr_solve_plan(self._h, _addr(p)), "myr_solve_plan")
    return {"form": ("lane", "fused", "wave", "shooting_wave")[int(p[0])], "waves_per_trajectory": int(p[1]), "park_iter": int(p[2]),
            "launches_per_solve": int(p[3]), "slots": int(p[4]), "helpers_max": int(p[5])}

  def kernel_time(self, kernel_id: int):
    ms = C.c_double(); n = C.c_int32()
    _chk(self.lib.myr_kernel_time(self._h, kernel_id, C.byref(ms), C.byref(n)), "myr_kernel_time")
    return ms.value, n.value

  def kernel_time_reset(self):
    _chk(self.lib.myr_kernel_time_reset(self._h), "myr_kernel_time_reset")
