"""TEST INFRASTRUCTURE ONLY -- never imported by the package, bench.py or any `-m gpu` test.

A name-forwarding stand-in for the small part of JAX (and of gin / seaborn / tensorboardX / simple_parsing / cyipopt) that the reference's
hot-path files touch, so that `tests/golden/make_reference_fixtures.py` can execute THE REFERENCE'S OWN LINES in the build container (which has
no jax): `jax.numpy` -> numpy, `jit` = identity, `vmap` / `lax.scan` as Python loops, `ravel_pytree`, `grad` of a scalar function by the
complex-step derivative (exact to rounding for the analytic scalar functions the reference differentiates, mountain_car.py:91).  It implements
none of the reference's algorithm: every number the fixtures hold is computed by the files under /root/reference/myriad, read in place.

What this is NOT: the reference "run here" in the sense of the parity rules -- a stand-in for a library the image lacks does not make it so, and
DESIGN.md keeps saying "parity unpinned at the JAX / IPOPT boundary".  What it buys: the oracle's restatement of the transcriptions is held to
numbers the reference's lines produced (1e-13), instead of to a reading of those lines.

install() puts the stand-ins into sys.modules; nothing is written anywhere."""
import sys
import types

import numpy as np


def _vmap(f, in_axes=0, out_axes=0):
  def stack(items):
    first = items[0]
    if isinstance(first, tuple):
      return tuple(stack([it[k] for it in items]) for k in range(len(first)))
    if isinstance(first, list):
      return [stack([it[k] for it in items]) for k in range(len(first))]
    if first is None:
      return None
    return np.stack([np.asarray(it) for it in items], axis=0)

  def mapped(*args):
    axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
    if len(axes) < len(args):
      raise ValueError("vmap: in_axes shorter than the arguments")      # (jax raises too; quirk Q8 of the reference trips this)
    def axis_len(a, ax):
      if isinstance(a, (tuple, list)):
        return axis_len(a[0], ax)
      return np.asarray(a).shape[ax]
    n = None
    for a, ax in zip(args, axes):
      if ax is not None:
        n = axis_len(a, ax); break
    def take(a, ax, i):
      if ax is None:
        return a
      if isinstance(a, (tuple, list)):
        return type(a)(take(x, ax, i) for x in a)
      return np.asarray(np.take(np.asarray(a), i, axis=ax)).view(_Arr)
    return stack([f(*[take(a, ax, i) for a, ax in zip(args, axes)]) for i in range(n)])
  return mapped


def _scan(f, init, xs, length=None):
  carry = init
  ys = []
  n = len(xs) if xs is not None else length
  for i in range(n):
    carry, y = f(carry, None if xs is None else xs[i])
    ys.append(y)
  if ys and ys[0] is None:
    return carry, None
  if ys and isinstance(ys[0], tuple):
    return carry, tuple(np.stack([np.asarray(y[k]) for y in ys]) for k in range(len(ys[0])))
  return carry, np.stack([np.asarray(y) for y in ys]) if ys else np.zeros((0,))


def _jit(f=None, **kw):
  if f is None:
    return lambda g: g
  return f


def _grad(f, argnums=0):
  """d f / d x_argnums for a scalar-valued analytic f of a scalar (or 1-element) argument: complex-step derivative, step 1e-30 (no
  subtractive cancellation: the result equals the exact derivative to rounding)."""
  def df(*args):
    x = np.asarray(args[argnums], dtype=np.float64)
    out = np.zeros_like(x)
    flat = x.reshape(-1)
    for i in range(flat.size):
      xc = flat.astype(np.complex128)
      xc[i] += 1e-30j
      a = list(args); a[argnums] = xc.reshape(x.shape).view(_Arr)
      out.reshape(-1)[i] = np.imag(np.asarray(f(*a)).reshape(-1)[0]) / 1e-30
    return out
  return df


def _jacrev(f, argnums=0):
  """Jacobian of a vector-valued analytic f by the complex-step derivative, one column per input entry (the reference calls jax.jacrev on
  its constraints, nlp_solvers/__init__.py:37; rounding-exact for the smooth transcriptions, not usable through abs / clip / max)."""
  def jac(*args):
    x = np.asarray(args[argnums], dtype=np.float64).reshape(-1)
    cols = []
    for i in range(x.size):
      xc = x.astype(np.complex128); xc[i] += 1e-30j
      a = list(args); a[argnums] = xc.view(_Arr)
      cols.append(np.imag(np.asarray(f(*a)).reshape(-1)) / 1e-30)
    return np.stack(cols, axis=1)
  return jac


def _ravel_pytree(tree):
  leaves = [np.asarray(x) for x in tree]
  leaves = [x if x.dtype.kind == "c" else x.astype(np.float64) for x in leaves]
  shapes = [x.shape for x in leaves]
  sizes = [x.size for x in leaves]
  flat = np.concatenate([x.reshape(-1) for x in leaves]) if leaves else np.zeros((0,))
  def unravel(v):
    v = np.asarray(v)
    out, o = [], 0
    for sh, sz in zip(shapes, sizes):
      out.append(np.asarray(v[o:o + sz]).reshape(sh).view(_Arr)); o += sz
    return type(tree)(out) if isinstance(tree, (tuple, list)) else out
  return flat, unravel


class _At:
  def __init__(self, a): self.a = a
  def __getitem__(self, idx):
    a = self.a
    class _S:
      def set(self, v):
        b = np.array(a, copy=True); b[idx] = v; return b.view(_Arr)
      def add(self, v):
        b = np.array(a, copy=True); b[idx] += v; return b.view(_Arr)
    return _S()


class _Arr(np.ndarray):
  """numpy array with two of jax's array semantics the reference relies on: the functional `.at[idx].set(v)` (forward_backward_sweep.py:86), and
  READS with an out-of-range integer index clamp to the last element instead of raising (jnp gather semantics; the reference's RK4 paths index
  control rows past the end -- quirk Q6: tests/tests.py:37, and the shooting guess under RK4, shooting.py:66-75 with utils.py:91-96)"""
  @property
  def at(self): return _At(self)

  def __getitem__(self, idx):
    if isinstance(idx, (int, np.integer)) or (isinstance(idx, np.ndarray) and idx.ndim == 0 and idx.dtype.kind in "iu"):
      i = int(idx)
      n = self.shape[0]
      if i >= n: i = n - 1
      elif i < -n: i = 0
      return super().__getitem__(i)
    return super().__getitem__(idx)

  def __iter__(self):                    # (iteration and unpacking stop at the end: only explicit indexing clamps)
    for i in range(self.shape[0]):
      yield np.ndarray.__getitem__(self, i)


def install():
  if "jax" in sys.modules and getattr(sys.modules["jax"], "__refshim__", False):
    return
  jnp = types.ModuleType("jax.numpy")
  for name in dir(np):
    if not name.startswith("_"):
      try:
        setattr(jnp, name, getattr(np, name))
      except Exception:
        pass
  jnp.NINF = -np.inf
  jnp.float = float
  jnp.ndarray = np.ndarray
  def _array(x, dtype=None, **kw):
    a = np.array(x, dtype=dtype if dtype is not None else None)
    if a.dtype.kind in "iub" and dtype is None:
      return a.view(_Arr)
    return (a.astype(np.float64) if a.dtype.kind == "f" else a).view(_Arr)
  jnp.array = _array
  jnp.asarray = lambda x, dtype=None: _array(x, dtype)
  jnp.zeros = lambda *a, **k: np.zeros(*a, **k).view(_Arr)
  jnp.ones = lambda *a, **k: np.ones(*a, **k).view(_Arr)
  jnp.empty_like = lambda *a, **k: np.empty_like(*a, **k).view(_Arr)
  jnp.zeros_like = lambda *a, **k: np.zeros_like(*a, **k).view(_Arr)
  jnp.linspace = lambda *a, **k: np.linspace(*a, **k).view(_Arr)
  jnp.arange = lambda *a, **k: np.arange(*a, **k).view(_Arr)
  for _n in ("hstack", "vstack", "concatenate", "append", "ravel", "squeeze", "stack"):
    (lambda n: setattr(jnp, n, lambda *a, **k: np.asarray(getattr(np, n)(*a, **k)).view(_Arr)))(_n)

  lax = types.ModuleType("jax.lax"); lax.scan = _scan
  flat = types.ModuleType("jax.flatten_util"); flat.ravel_pytree = _ravel_pytree
  rnd = types.ModuleType("jax.random")
  rnd.PRNGKey = lambda seed: np.array([0, int(seed)], dtype=np.uint32)
  rnd.split = lambda key, num=2: tuple(np.array([int(key[1]) + 1 + i, int(key[1])], dtype=np.uint32) for i in range(num))
  def _no_random(*a, **k): raise NotImplementedError("refshim: jax.random draws are not reproduced (the fixture generator never asks for them)")
  rnd.normal = rnd.uniform = _no_random
  cfgm = types.ModuleType("jax.config")
  class _Cfg:
    def update(self, *a, **k): pass
  cfgm.config = _Cfg()

  jax = types.ModuleType("jax"); jax.__refshim__ = True; jax.__path__ = []
  jax.numpy = jnp; jax.lax = lax; jax.flatten_util = flat; jax.random = rnd; jax.config = cfgm.config
  jax.jit = _jit; jax.vmap = _vmap; jax.grad = _grad
  def _unsupported(*a, **k): raise NotImplementedError("refshim: second derivatives are not provided")
  jax.jacrev = jax.jacfwd = _jacrev
  jax.hessian = _unsupported
  sys.modules.update({"jax": jax, "jax.numpy": jnp, "jax.lax": lax, "jax.flatten_util": flat, "jax.random": rnd, "jax.config": cfgm})

  gin = types.ModuleType("gin")
  def configurable(*a, **k):
    if len(a) == 1 and callable(a[0]) and not k:
      return a[0]
    return lambda f: f
  gin.configurable = configurable; gin.REQUIRED = object()
  gin.parse_config_file = gin.parse_config = gin.bind_parameter = lambda *a, **k: None
  sys.modules["gin"] = gin
  sns = types.ModuleType("seaborn")
  sns.set = sns.set_style = sns.set_context = sns.set_theme = sns.despine = lambda *a, **k: None
  sns.color_palette = lambda *a, **k: ["C%d" % i for i in range(10)]
  sys.modules["seaborn"] = sns
  tbx = types.ModuleType("tensorboardX")
  class SummaryWriter:
    def __init__(self, *a, **k): pass
    def add_scalar(self, *a, **k): pass
    def close(self): pass
  tbx.SummaryWriter = SummaryWriter
  sys.modules["tensorboardX"] = tbx
  sp = types.ModuleType("simple_parsing"); sp.ArgumentParser = object
  sys.modules["simple_parsing"] = sp
  cy = types.ModuleType("cyipopt")
  def minimize_ipopt(*a, **k): raise NotImplementedError("refshim: cyipopt is not in this image")
  cy.minimize_ipopt = minimize_ipopt
  sys.modules["cyipopt"] = cy
