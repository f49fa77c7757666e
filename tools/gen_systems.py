#!/usr/bin/env python3
"""
Generate myriad_amd/csrc/systems_gen.h: closed-form f, (A,B)=df/d(x,u), cost g, dg, and the
Lagrangian-Hessian contraction  W = w*d2g + sum_i mu_i * d2f_i  for each control system on the
hot path, as plain `double` straight-line code usable from both hipcc (device) and g++ (host).

The symbolic definitions below RESTATE the reference's system definitions (citations are relative
to /root/reference/); derivatives and common-subexpression elimination are done by sympy here, at
development time.  The generated header is committed; this script is only re-run when a system
is added.  Run:  python tools/gen_systems.py
"""
import os
import sympy as sp
from sympy.printing.c import C99CodePrinter

OUT = os.path.join(os.path.dirname(__file__), "..", "myriad_amd", "csrc", "systems_gen.h")


class inboxf(sp.Function):
  """1 where lo <= x <= hi else 0 (the derivative of clip inside the box); piecewise constant."""
  nargs = 3
  def fdiff(self, argindex=1):
    return sp.S.Zero


class clipf(sp.Function):
  """jnp.clip(x, lo, hi)"""
  nargs = 3
  def fdiff(self, argindex=1):
    x, lo, hi = self.args
    return inboxf(x, lo, hi) if argindex == 1 else sp.S.Zero


class angnormf(sp.Function):
  """angle_normalize(x) = ((x + pi) % (2 pi)) - pi  (pendulum.py:17-18); derivative 1 almost everywhere"""
  nargs = 1
  def fdiff(self, argindex=1):
    return sp.S.One


class Printer(C99CodePrinter):
  def _print_clipf(self, e):
    return "myr_clip(%s, %s, %s)" % tuple(self._print(a) for a in e.args)

  def _print_inboxf(self, e):
    return "myr_inbox(%s, %s, %s)" % tuple(self._print(a) for a in e.args)

  def _print_angnormf(self, e):
    return "myr_angnorm(%s)" % self._print(e.args[0])

  def _print_Pow(self, expr):
    b, e = expr.as_base_exp()
    if e == 2:
      s = self._print(b)
      return f"(({s})*({s}))"
    if e == -1:
      return f"(1.0/({self._print(b)}))"
    if e == -2:
      s = self._print(b)
      return f"(1.0/(({s})*({s})))"
    if e == 3:
      s = self._print(b)
      return f"(({s})*({s})*({s}))"
    return super()._print_Pow(expr)


PR = Printer()


def systems():
  out = []
  # ---- CARTPOLE: myriad/systems/classical_control/cartpole.py:76-87 (dynamics), :106-108 (cost) ----
  x = sp.symbols("x0:4", real=True)
  u = sp.symbols("u0:1", real=True)
  g, m1, m2, L = p = sp.symbols("p0:4", real=True)   # g, m1, m2, length  (ctor order, cartpole.py:50)
  th, dx, dth = x[1], x[2], x[3]
  ddx = (L * m2 * sp.sin(th) * dth ** 2 + u[0] + m2 * g * sp.cos(th) * sp.sin(th)) / (m1 + m2 * (1 - sp.cos(th) ** 2))
  ddth = -((L * m2 * sp.cos(th) * dth ** 2 + u[0] * sp.cos(th) + (m1 + m2) * g * sp.sin(th))
           / (L * m1 + L * m2 * (1 - sp.cos(th) ** 2)))
  out.append(dict(name="CARTPOLE", id=0, x=x, u=u, p=p, f=[dx, dth, ddx, ddth], g=u[0] ** 2,
                  pdefault=[9.81, 1.0, 0.3, 0.5], pnames=["g", "m1", "m2", "length"]))
  # ---- VANDERPOL: myriad/systems/miscellaneous/van_der_pol.py:46-50, :59-60 ----
  x = sp.symbols("x0:2", real=True)
  u = sp.symbols("u0:1", real=True)
  p = sp.symbols("p0:1", real=True)   # a
  out.append(dict(name="VANDERPOL", id=1, x=x, u=u, p=p,
                  f=[p[0] * (1 - x[1] ** 2) * x[0] - x[1] + u[0], x[0]],
                  g=x[0] ** 2 + x[1] ** 2 + u[0] ** 2, pdefault=[1.0], pnames=["a"]))
  # ---- CANCERTREATMENT: myriad/systems/lenhart/cancer_treatment.py:62-65, :75-76 ----
  x = sp.symbols("x0:1", positive=True)
  u = sp.symbols("u0:1", real=True)
  p = sp.symbols("p0:3", real=True)   # r, a, delta (ctor order, cancer_treatment.py:40)
  out.append(dict(name="CANCERTREATMENT", id=2, x=x, u=u, p=p,
                  f=[p[0] * x[0] * sp.log(1 / x[0]) - u[0] * p[2] * x[0]],
                  g=p[1] * x[0] ** 2 + u[0] ** 2, pdefault=[0.3, 3.0, 0.45], pnames=["r", "a", "delta"]))
  # ---- SIMPLECASE: myriad/systems/lenhart/simple_case.py:46-53 ----
  x = sp.symbols("x0:1", real=True)
  u = sp.symbols("u0:1", real=True)
  p = sp.symbols("p0:3", real=True)   # A, B, C
  out.append(dict(name="SIMPLECASE", id=3, x=x, u=u, p=p,
                  f=[-sp.Rational(1, 2) * x[0] ** 2 + p[2] * u[0]],
                  g=-p[0] * x[0] + p[1] * u[0] ** 2, pdefault=[1.0, 1.0, 4.0], pnames=["A", "B", "C"]))
  # ==== SURVEY.md 8(f4): further autonomous systems without terminal cost =====================================
  # ---- BIOREACTOR: myriad/systems/lenhart/bioreactor.py:60-69 (dynamics), :82-83 (cost) ----
  x = sp.symbols("x0:1", real=True)
  u = sp.symbols("u0:1", real=True)
  p = sp.symbols("p0:3", real=True)   # K, G, D
  out.append(dict(name="BIOREACTOR", id=5, x=x, u=u, p=p,
                  f=[p[1] * u[0] * x[0] - p[2] * x[0] ** 2], g=-p[0] * x[0] + u[0],
                  pdefault=[2.0, 1.0, 1.0], pnames=["K", "G", "D"]))
  # ---- GLUCOSE: myriad/systems/lenhart/glucose.py:74-84, :103-104 ----
  x = sp.symbols("x0:2", real=True)
  u = sp.symbols("u0:1", real=True)
  p = sp.symbols("p0:5", real=True)   # a, b, c, A, l
  out.append(dict(name="GLUCOSE", id=6, x=x, u=u, p=p,
                  f=[-p[0] * x[0] - p[1] * x[1], -p[2] * x[1] + u[0]],
                  g=100000 * (p[3] * (x[0] - p[4]) ** 2 + u[0] ** 2),
                  pdefault=[1.0, 1.0, 1.0, 2.0, 0.5], pnames=["a", "b", "c", "A", "l"]))
  # ---- MOULDFUNGICIDE: myriad/systems/lenhart/mould_fungicide.py:54-58, :69-70 ----
  x = sp.symbols("x0:1", real=True)
  u = sp.symbols("u0:1", real=True)
  p = sp.symbols("p0:3", real=True)   # r, M, A
  out.append(dict(name="MOULDFUNGICIDE", id=7, x=x, u=u, p=p,
                  f=[p[0] * (p[1] - x[0]) - u[0] * x[0]], g=p[2] * x[0] ** 2 + u[0] ** 2,
                  pdefault=[0.3, 10.0, 10.0], pnames=["r", "M", "A"]))
  # ---- SIMPLECASEWITHBOUNDS: myriad/systems/lenhart/simple_case_with_bounds.py:47-55 ----
  x = sp.symbols("x0:1", real=True)
  u = sp.symbols("u0:1", real=True)
  p = sp.symbols("p0:2", real=True)   # A, C
  out.append(dict(name="SIMPLECASEWITHBOUNDS", id=8, x=x, u=u, p=p,
                  f=[-sp.Rational(1, 2) * x[0] ** 2 + p[1] * u[0]], g=-p[0] * x[0] + u[0] ** 2,
                  pdefault=[1.0, 4.0], pnames=["A", "C"]))
  # ---- HIVTREATMENT: myriad/systems/lenhart/hiv_treatment.py:72-84, :110-111 ----
  x = sp.symbols("x0:3", real=True)
  u = sp.symbols("u0:1", real=True)
  s_, m1_, m2_, m3_, r_, Tm_, k_, N_, A_ = p = sp.symbols("p0:9", real=True)   # s, m_1, m_2, m_3, r, T_max, k, N, A
  out.append(dict(name="HIVTREATMENT", id=9, x=x, u=u, p=p,
                  f=[s_ / (1 + x[2]) - m1_ * x[0] + r_ * x[0] * (1 - (x[0] + x[1]) / Tm_) - u[0] * k_ * x[0] * x[2],
                     u[0] * k_ * x[0] * x[2] - m2_ * x[1],
                     N_ * m2_ * x[1] - m3_ * x[2]],
                  g=-A_ * x[0] + (1 - u[0]) ** 2,
                  pdefault=[10.0, 0.02, 0.5, 4.4, 0.03, 1500.0, 0.000024, 300.0, 0.05],
                  pnames=["s", "m_1", "m_2", "m_3", "r", "T_max", "k", "N", "A"]))
  # ---- EPIDEMICSEIRN: myriad/systems/lenhart/epidemic_seirn.py:78-92, :94-95 ----
  x = sp.symbols("x0:4", real=True)
  u = sp.symbols("u0:1", real=True)
  A_, b_, d_, c_, e_, g_, a_ = p = sp.symbols("p0:7", real=True)   # A, b, d, c, e, g, a
  seirn_f = [b_ * x[3] - d_ * x[0] - c_ * x[0] * x[2] - u[0] * x[0],
             c_ * x[0] * x[2] - (e_ + d_) * x[1],
             e_ * x[1] - (g_ + a_ + d_) * x[2],
             (b_ - d_) * x[3] - a_ * x[2]]
  out.append(dict(name="EPIDEMICSEIRN", id=10, x=x, u=u, p=p, f=seirn_f, g=A_ * x[2] + u[0] ** 2,
                  pdefault=[0.1, 0.525, 0.5, 0.0001, 0.5, 0.1, 0.2], pnames=["A", "b", "d", "c", "e", "g", "a"]))
  # ---- SEIR: myriad/systems/miscellaneous/seir.py:83-95 (same field, fixed constants of :46-58) ----
  out.append(dict(name="SEIR", id=11, x=x, u=u, p=p, f=seirn_f, g=A_ * x[2] + u[0] ** 2,
                  pdefault=[0.1, 0.525, 0.5, 0.0001, 0.5, 0.1, 0.2], pnames=["A", "b", "d", "c", "e", "g", "a"]))
  # ---- BEARPOPULATIONS: myriad/systems/lenhart/bear_populations.py:72-86, :109-110 (two controls) ----
  x = sp.symbols("x0:3", real=True)
  u = sp.symbols("u0:2", real=True)
  r_, K_, mp_, mf_, cp_, cf_ = p = sp.symbols("p0:6", real=True)   # r, K, m_p, m_f, c_p, c_f
  k1, k2 = r_ / K_, r_ / K_ ** 2
  out.append(dict(name="BEARPOPULATIONS", id=12, x=x, u=u, p=p,
                  f=[r_ * x[0] - k1 * x[0] ** 2 + k1 * mf_ * (1 - x[0] / K_) * x[1] ** 2 - u[0] * x[0],
                     r_ * x[1] - k1 * x[1] ** 2 + k1 * mp_ * (1 - x[1] / K_) * x[0] ** 2 - u[1] * x[1],
                     k1 * (1 - mp_) * x[0] ** 2 + k1 * (1 - mf_) * x[1] ** 2 + k2 * mf_ * x[0] * x[1] ** 2 + k2 * mp_ * x[0] ** 2 * x[1]],
                  g=x[2] + cp_ * u[0] ** 2 + cf_ * u[1] ** 2,
                  pdefault=[0.1, 0.75, 0.5, 0.5, 10000.0, 10.0], pnames=["r", "K", "m_p", "m_f", "c_p", "c_f"]))
  # ---- PENDULUM: myriad/systems/classical_control/pendulum.py:94-108 (dynamics), :114-120 (cost); gym-style clips kept ----
  x = sp.symbols("x0:2", real=True)
  u = sp.symbols("u0:1", real=True)
  g_, m_, L_ = p = sp.symbols("p0:3", real=True)   # g, m, length
  uc = clipf(u[0], -2.0, 2.0)                       # max_torque (:58)
  th = angnormf(x[0])
  dth = clipf(x[1], -8.0, 8.0)                      # max_speed (:57)
  out.append(dict(name="PENDULUM", id=13, x=x, u=u, p=p,
                  f=[dth, (-3 * g_ / (2 * L_) * sp.sin(th) + 3 * uc / (m_ * L_ ** 2)) * sp.Rational(1, 20)],
                  g=angnormf(x[0]) ** 2 + sp.Rational(1, 10) * x[1] ** 2 + sp.Rational(1, 1000) * u[0] ** 2,
                  pdefault=[10.0, 1.0, 1.0], pnames=["g", "m", "length"]))
  # ---- MOUNTAINCAR: myriad/systems/classical_control/mountain_car.py:83-89 (hill_function = x^2/2, :11-13), :100-101 ----
  x = sp.symbols("x0:2", real=True)
  u = sp.symbols("u0:1", real=True)
  p = sp.symbols("p0:2", real=True)   # power, gravity
  out.append(dict(name="MOUNTAINCAR", id=14, x=x, u=u, p=p,
                  f=[x[1], clipf(u[0], -1.0, 1.0) * p[0] - p[1] * x[0]], g=10 * u[0] ** 2,
                  pdefault=[0.0015, 0.0025], pnames=["power", "gravity"]))
  # ---- ROCKETLANDING: myriad/systems/miscellaneous/rocket_landing.py:99-120 (two controls, six states) ----
  x = sp.symbols("x0:6", real=True)
  u = sp.symbols("u0:2", real=True)
  g_, m_, L_ = p = sp.symbols("p0:3", real=True)   # g, m, length
  Fmax = 2210 * 1000                                # max_thrust (:61)
  I_ = m_ * L_ ** 2 / 12                            # :62
  out.append(dict(name="ROCKETLANDING", id=15, x=x, u=u, p=p,
                  f=[x[1], Fmax * u[0] * sp.sin(u[1] + x[4]) / m_,
                     x[3], Fmax * u[0] * sp.cos(u[1] + x[4]) / m_ - g_,
                     x[5], -L_ / 2 * Fmax * u[0] * sp.sin(u[1]) / I_],
                  g=u[0] ** 2 + u[1] ** 2 + 2 * x[5] ** 2,
                  pdefault=[9.8, 100000.0, 50.0], pnames=["g", "m", "length"]))
  # ==== systems with a (linear) terminal cost: applied by the trapezoidal transcription (trapezoidal.py:126-127) and the
  # post-solve rollout (utils.py:295-296); NOT by the reference's Hermite-Simpson objective (hermite_simpson.py:243-257)
  # ---- BACTERIA: myriad/systems/lenhart/bacteria.py:59-63, :77-78, terminal :84-86 ----
  x = sp.symbols("x0:1", real=True)
  u = sp.symbols("u0:1", real=True)
  p = sp.symbols("p0:4", real=True)   # r, A, B, C
  out.append(dict(name="BACTERIA", id=16, x=x, u=u, p=p,
                  f=[p[0] * x[0] + p[1] * u[0] * x[0] - p[2] * u[0] ** 2 * sp.exp(-x[0])], g=u[0] ** 2, gT=-p[3] * x[0],
                  pdefault=[1.0, 1.0, 12.0, 1.0], pnames=["r", "A", "B", "C"]))
  # ---- TUMOUR: myriad/systems/miscellaneous/tumour.py:80-86, cost 0 :100-101, terminal :106-108 ----
  x = sp.symbols("x0:3", positive=True)
  u = sp.symbols("u0:1", real=True)
  xi_, b_, d_, G_, mu_ = p = sp.symbols("p0:5", real=True)   # xi, b, d, G, mu
  out.append(dict(name="TUMOUR", id=17, x=x, u=u, p=p,
                  f=[-xi_ * x[0] * sp.log(x[0] / x[1]), x[1] * (b_ - (mu_ + d_ * x[0] ** sp.Rational(2, 3) + G_ * u[0])), u[0]],
                  g=sp.S.Zero, gT=x[0],
                  pdefault=[0.084, 5.85, 0.00873, 0.15, 0.02], pnames=["xi", "b", "d", "G", "mu"]))
  # ==== systems whose running cost depends on time: the point's time sits in a slot of the parameter vector ====
  tt = sp.Symbol("tt", nonnegative=True)
  # ---- HARVEST: myriad/systems/lenhart/harvest.py:54-62 ----
  x = sp.symbols("x0:1", real=True)
  u = sp.symbols("u0:1", real=True)
  p = sp.symbols("p0:3", real=True)   # A, k, m
  out.append(dict(name="HARVEST", id=18, x=x, u=u, p=p,
                  f=[-(p[2] + u[0]) * x[0]], g=-p[0] * (p[1] * tt / (tt + 1)) * x[0] * u[0] + u[0] ** 2,
                  pdefault=[5.0, 10.0, 0.2], pnames=["A", "k", "m"]))
  # ---- TIMBERHARVEST: myriad/systems/lenhart/timber_harvest.py:62-69, :84-85 ----
  x = sp.symbols("x0:1", real=True)
  u = sp.symbols("u0:1", real=True)
  p = sp.symbols("p0:2", real=True)   # r, k
  out.append(dict(name="TIMBERHARVEST", id=19, x=x, u=u, p=p,
                  f=[p[1] * x[0] * u[0]], g=-sp.exp(-p[0] * tt) * x[0] * (1 - u[0]),
                  pdefault=[0.0, 1.0], pnames=["r", "k"]))
  # ---- PREDATORPREY: myriad/systems/lenhart/predator_prey.py:85-96, cost :113-114, terminal :120-122.  x_T = [None, None, B]
  # pins one terminal state only: the reference's collocation optimisers cannot build that (np.expand_dims / linspace over
  # None), its shooting optimiser and the FBSM secant solver can -- same here (host side).
  x = sp.symbols("x0:3", real=True)
  u = sp.symbols("u0:1", real=True)
  p = sp.symbols("p0:3", real=True)   # d_1, d_2, A
  out.append(dict(name="PREDATORPREY", id=20, x=x, u=u, p=p,
                  f=[(1 - x[1]) * x[0] - p[0] * x[0] * u[0], (x[0] - 1) * x[1] - p[1] * x[1] * u[0], u[0]],
                  g=p[2] * sp.Rational(1, 2) * u[0] ** 2, gT=x[0],
                  pdefault=[0.1, 0.1, 1.0], pnames=["d_1", "d_2", "A"]))
  # ==== ELASTIC twins (no counterpart in the reference: the solver's feasibility-restoration device, DESIGN.md "Elastic mode") ====
  # x' = f(x, u) + s with NS slack "controls" s and the running cost g + rho/2 |s|^2: every state trajectory is feasible for the
  # twin, and its optima approach those of the system as rho grows (quadratic penalty on the dynamics residual).  ids: 100 + id.
  # Under the solver's variable scaling the penalty is rho/2 |s / sc|^2 (sc: the scale of the slack = that of its state).
  base = {S["name"]: S for S in out}
  # (round 4: every system with pinned terminal states -- the ones a solve can jam on -- has its twin; PREDATORPREY pins one state and is
  #  refused by the collocation transcriptions, the reference's behaviour)
  for nm in ("PENDULUM", "ROCKETLANDING", "CARTPOLE", "VANDERPOL", "MOUNTAINCAR"):
    out.append(elastic(base[nm]))
  return out


def elastic(S):
  ns, nu, npar = len(S["x"]), len(S["u"]), len(S["p"])
  s = sp.symbols(f"u{nu}:{nu + ns}", real=True)
  rho = sp.Symbol(f"p{npar}", real=True)
  return dict(name=S["name"] + "_ELASTIC", id=100 + S["id"], x=S["x"], u=tuple(S["u"]) + tuple(s), p=tuple(S["p"]) + (rho,),
              f=[S["f"][i] + s[i] for i in range(ns)], g=S["g"], g_scaled=rho / 2 * sum(v ** 2 for v in s),
              pdefault=list(S["pdefault"]) + [1.0], pnames=list(S["pnames"]) + ["rho"])


def emit_block(assigns, indent="  "):
  """assigns: list of (lhs_string, expr).  CSE over all, emit straight-line code."""
  exprs = [e for _, e in assigns]
  repl, red = sp.cse(exprs, symbols=sp.numbered_symbols("t"), optimizations="basic")
  lines = []
  for s, e in repl:
    lines.append(f"{indent}const double {s} = {PR.doprint(e)};")
  for (lhs, _), e in zip(assigns, red):
    lines.append(f"{indent}{lhs} = {PR.doprint(e)};")
  return "\n".join(lines)


def gen_system(S):
  name, x, u, p = S["name"], list(S["x"]), list(S["u"]), list(S["p"])
  ns, nu, npar = len(x), len(u), len(p)
  w = x + u
  nw = ns + nu
  # Variable scaling (solver-side conditioning): the generated functions take SCALED variables
  # xt = x / sc, ut = u / sc and return the scaled field ft_i = f_i(sc xt, sc ut) / sc_i and the unchanged cost; the NW
  # scale factors follow the NP model parameters in the parameter vector (all 1 unless the solver sets them, in which
  # case every product below is exact and the functions are the reference's).  Derivatives are taken AFTER the
  # substitution, so A, B and the second derivatives are those of the scaled problem.
  sc = list(sp.symbols(f"sc0:{nw}", positive=True))
  isc = list(sp.symbols(f"isc0:{ns}", positive=True))      # 1 / sc_i, precomputed by SysParams::set_scale (no divisions here)
  smap = {v: sc[i] * v for i, v in enumerate(w)}
  f = [sp.sympify(e).subs(smap, simultaneous=True) * isc[i] for i, e in enumerate(S["f"])]
  g = sp.sympify(S["g"]).subs(smap, simultaneous=True)
  # (elastic twins) a term of the running cost written in the SCALED variables: rho/2 |s / sc|^2 weighs every slack against the
  # magnitude of its state, whatever the units; with unit scales (every entry point but the scaled solve) it is rho/2 |s|^2
  g = g + sp.sympify(S.get("g_scaled", 0))
  tt = sp.Symbol("tt", nonnegative=True)                              # time of the point (cost only; the dynamics of all systems are autonomous)
  time_dep = g.has(tt)
  assert not any(e.has(tt) for e in f), "time-dependent dynamics are not generated"
  gT = sp.sympify(S.get("gT", 0)).subs(smap, simultaneous=True)      # terminal cost (systems/base.py:101-111), in scaled variables
  has_term = gT != 0
  gTw = [sp.diff(gT, v) for v in (x + u)]
  assert all(sp.simplify(sp.diff(e, v)) == 0 for e in gTw for v in (x + u)), "terminal cost must be linear (no Hessian term is generated)"
  A = [[sp.diff(f[i], x[j]) for j in range(ns)] for i in range(ns)]
  Bm = [[sp.diff(f[i], u[j]) for j in range(nu)] for i in range(ns)]
  gw = [sp.diff(g, v) for v in w]
  mu = sp.symbols(f"mu0:{ns}", real=True)
  wg = sp.Symbol("wg", real=True)
  lag = wg * g + sum(mu[i] * f[i] for i in range(ns))
  H = [[sp.diff(lag, w[i], w[j]) for j in range(nw)] for i in range(nw)]
  cost_dep_x = any(sp.diff(g, v) != 0 for v in x) or any(sp.diff(gT, v) != 0 for v in x)

  def unpack(indent="  "):
    s = []
    for i, v in enumerate(x):
      s.append(f"{indent}const double {v} = x[{i}];")
    for i, v in enumerate(u):
      s.append(f"{indent}const double {v} = u[{i}];")
    for i, v in enumerate(p):
      s.append(f"{indent}const double {v} = p[{i}];")
    for i, v in enumerate(sc):
      s.append(f"{indent}const double {v} = p[{npar + i}];")
    for i, v in enumerate(isc):
      s.append(f"{indent}const double {v} = p[{npar + nw + i}];")
    s.append(f"{indent}const double tt = p[{npar + nw + ns}];")
    return "\n".join(s) + "\n" + "\n".join(f"{indent}(void){v};" for v in (x + u + p + sc + isc + [tt]))

  o = []
  o.append(f"// ===== {name} (id {S['id']}): ns={ns} nu={nu} np={npar} =====")
  o.append(f"struct Sys{name} {{")
  o.append(f"  static constexpr int ID = {S['id']}, NS = {ns}, NU = {nu}, NP = {npar}, NW = {nw};")
  o.append(f"  static constexpr int NPX = NP + NW + NS + 1;   // model parameters, the NW variable scales, the NS inverse state scales, the time slot")
  o.append(f"  static constexpr int T_SLOT = NP + NW + NS;     // p[T_SLOT]: time of the point being evaluated (set_time(); cost only)")
  o.append(f"  static constexpr bool TIME_DEP = {'true' if time_dep else 'false'};   // running cost g(x,u,t) depends on t")
  o.append(f"  static constexpr bool COST_DEP_X = {'true' if cost_dep_x else 'false'};")
  o.append(f"  static constexpr const char* NAME = \"{name}\";")
  o.append("  static constexpr bool PARAMS_BY_POINTER = false;   // parameters are a handful of scalars: copied to registers")
  # f
  o.append("  // dynamics f(x,u)")
  o.append("  MYR_HD static inline void f(const double* x, const double* u, const double* p, double* fo) {")
  o.append(unpack("    "))
  o.append(emit_block([(f"fo[{i}]", f[i]) for i in range(ns)], "    "))
  o.append("  }")
  # cost
  o.append("  // running cost g(x,u)")
  o.append("  MYR_HD static inline double g(const double* x, const double* u, const double* p) {")
  o.append(unpack("    "))
  o.append("    double r;")
  o.append(emit_block([("r", g)], "    "))
  o.append("    return r;")
  o.append("  }")
  # g, gw only
  o.append("  // running cost and its gradient wrt (x,u)")
  o.append("  MYR_HD static inline void cost_grad(const double* x, const double* u, const double* p, double* go, double* gw) {")
  o.append(unpack("    "))
  o.append(emit_block([("*go", g)] + [(f"gw[{i}]", gw[i]) for i in range(nw)], "    "))
  o.append("  }")
  # f, A, B, g, gw
  o.append("  // f, A=df/dx (row-major ns x ns), B=df/du (ns x nu), g, dg/d(x,u)")
  o.append("  MYR_HD static inline void lin(const double* x, const double* u, const double* p,")
  o.append("                                double* fo, double* A, double* B, double* go, double* gw) {")
  o.append(unpack("    "))
  ass = [(f"fo[{i}]", f[i]) for i in range(ns)]
  ass += [(f"A[{i * ns + j}]", A[i][j]) for i in range(ns) for j in range(ns)]
  ass += [(f"B[{i * nu + j}]", Bm[i][j]) for i in range(ns) for j in range(nu)]
  ass += [("*go", g)] + [(f"gw[{i}]", gw[i]) for i in range(nw)]
  o.append(emit_block(ass, "    "))
  o.append("  }")
  # everything + structurally non-zero second derivatives, and the matching contraction
  d2 = []   # (kind, i, r, c, expr): kind 'f' -> d2 f_i / dw_r dw_c ; kind 'g' -> d2 g
  for i in range(ns):
    for r in range(nw):
      for c in range(r, nw):
        e = sp.simplify(sp.diff(f[i], w[r], w[c]))
        if e != 0:
          d2.append(("f", i, r, c, e))
  for r in range(nw):
    for c in range(r, nw):
      e = sp.simplify(sp.diff(g, w[r], w[c]))
      if e != 0:
        d2.append(("g", 0, r, c, e))
  nnz2 = max(1, len(d2))
  o.append(f"  static constexpr int NNZ2 = {nnz2};   // structurally non-zero second derivatives of (f, g)")
  o.append("  // as lin(), plus D2[NNZ2]: the non-zero second derivatives (pattern known to contract())")
  o.append("  MYR_HD static inline void lin_d2(const double* x, const double* u, const double* p,")
  o.append("                                   double* fo, double* A, double* B, double* go, double* gw, double* D2) {")
  o.append(unpack("    "))
  ass2 = list(ass) + [(f"D2[{q}]", t[4]) for q, t in enumerate(d2)]
  o.append(emit_block(ass2, "    "))
  if not d2:
    o.append("    D2[0] = 0;")
  o.append("  }")
  o.append("  // W (NW x NW row-major, symmetric) = wg * d2g + sum_i mu[i] * d2 f_i   (Hessian of the Lagrangian at one point)")
  o.append("  MYR_HD static inline void contract(const double* D2, const double* mu, double wg, double* W) {")
  o.append("    (void)D2; (void)mu; (void)wg;")
  terms = {}
  for q, (kind, i, r, c, e) in enumerate(d2):
    coef = "wg" if kind == "g" else f"mu[{i}]"
    terms.setdefault((r, c), []).append(f"{coef}*D2[{q}]")
  for r in range(nw):
    for c in range(r, nw):
      rhs = " + ".join(terms.get((r, c), [])) or "0.0"
      o.append(f"    W[{r * nw + c}] = {rhs};")
      if c != r:
        o.append(f"    W[{c * nw + r}] = W[{r * nw + c}];")
  o.append("  }")
  o.append("  // Hessian of the Lagrangian at one point from the data lin_d2() produced (uniform interface with SysNODE)")
  o.append("  MYR_HD static inline void hessian(const double* x, const double* u, const double* p, const double* D2,")
  o.append("                                    const double* mu, double wg, double* W) {")
  o.append("    (void)x; (void)u; (void)p;")
  o.append("    contract(D2, mu, wg, W);")
  o.append("  }")
  o.append(f"  static constexpr bool HAS_TERMINAL = {'true' if has_term else 'false'};   // linear terminal cost (applied by the trapezoidal transcription and the rollout)")
  o.append("  MYR_HD static inline double term(const double* x, const double* u, const double* p) {")
  o.append(unpack("    "))
  o.append("    double r;")
  o.append(emit_block([("r", gT)], "    "))
  o.append("    return r;")
  o.append("  }")
  o.append("  MYR_HD static inline void term_grad(const double* x, const double* u, const double* p, double* gw) {")
  o.append(unpack("    "))
  o.append(emit_block([(f"gw[{i}]", gTw[i]) for i in range(nw)], "    "))
  o.append("  }")
  o.append("  MYR_HD static inline void default_params(double* p) {")
  for i, v in enumerate(S["pdefault"]):
    o.append(f"    p[{i}] = {v!r};  // {S['pnames'][i]}")
  o.append(f"    for (int i = NP; i < T_SLOT; ++i) p[i] = 1.0;   // unit variable scales")
  o.append(f"    p[T_SLOT] = 0.0;")
  o.append("  }")
  o.append("};")
  return "\n".join(o)


def main():
  parts = ["// GENERATED by tools/gen_systems.py (sympy) -- do not edit by hand.",
           "// Closed-form dynamics / cost / derivative code for the control systems on the hot path.",
           "// Restates /root/reference/myriad/systems/{classical_control/cartpole.py:76-87,106-108,",
           "//   miscellaneous/van_der_pol.py:46-60, lenhart/cancer_treatment.py:62-76, lenhart/simple_case.py:46-53}.",
           "#pragma once",
           "#include <math.h>",
           "#ifndef MYR_HD",
           "#if defined(__HIPCC__)",
           "#define MYR_HD __host__ __device__",
           "#else",
           "#define MYR_HD",
           "#endif",
           "#endif",
           "namespace myriad {",
           "// gym-style helpers of the PENDULUM / MOUNTAINCAR fields (jnp.clip, angle_normalize with Python's % semantics)",
           "MYR_HD inline double myr_clip(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }",
           "MYR_HD inline double myr_inbox(double x, double lo, double hi) { return (x >= lo && x <= hi) ? 1.0 : 0.0; }",
           "MYR_HD inline double myr_angnorm(double x) {",
           "  const double two_pi = 6.283185307179586476925286766559, pi = 3.141592653589793238462643383279;",
           "  double t = fmod(x + pi, two_pi);",
           "  if (t < 0.0) t += two_pi;",
           "  return t - pi;",
           "}", ""]
  for S in systems():
    parts.append(gen_system(S))
    parts.append("")
  parts.append("""// Per-thread view of a system's parameters: small parameter sets are copied into registers, large ones (the
// weights of a neural-ODE system) are used in place through a pointer.
template <class Sys>
struct SysParams {
  double buf[Sys::PARAMS_BY_POINTER ? 1 : Sys::NP + Sys::NW + Sys::NS + 1];
  const double* ptr;
  MYR_HD inline void load(const double* params, long b, int stride) {
    if constexpr (Sys::PARAMS_BY_POINTER) {
      ptr = params + b * (long)stride;
    } else {
      ptr = nullptr;
      Sys::default_params(buf);
      if (params) { for (int i = 0; i < Sys::NP; ++i) buf[i] = params[b * (long)stride + i]; }
    }
  }
  // variable scales of the scaled problem the solver works on (closed-form systems only; scale[NW])
  MYR_HD inline void set_scale(const double* scale) {
    if constexpr (!Sys::PARAMS_BY_POINTER) {
      for (int i = 0; i < Sys::NW; ++i) buf[Sys::NP + i] = scale[i];
      for (int i = 0; i < Sys::NS; ++i) buf[Sys::NP + Sys::NW + i] = 1.0 / scale[i];
    }
    else (void)scale;
  }
  MYR_HD inline const double* get() const { if constexpr (Sys::PARAMS_BY_POINTER) return ptr; else return buf; }
};
""")
  parts.append("""// Time of the point whose cost is evaluated next (systems with g(x,u,t): harvest.py:61-62, timber_harvest.py:84-85).  `p`
// is the thread's own parameter buffer (SysParams::buf), so the slot is written through the const view the solver cores hold.
template <class Sys>
MYR_HD inline void set_time(const double* p, double t) {
  if constexpr (Sys::TIME_DEP) const_cast<double*>(p)[Sys::T_SLOT] = t;
  else { (void)p; (void)t; }
}

// Fold a (linear) terminal cost into the running cost of the LAST point of a quadrature with weight w_last:
// w_last (g + gT / w_last) = w_last g + gT, and likewise for the gradient -- every consumer of (g, gw) at that point then
// sees the terminal term without further changes (second derivatives are zero for a linear gT).
template <class Sys>
MYR_HD inline void fold_terminal(const double* x, const double* u, const double* p, double w_last, double& g, double* gw) {
  if constexpr (Sys::HAS_TERMINAL) {
    double tg[Sys::NW];
    Sys::term_grad(x, u, p, tg);
    const double iw = 1.0 / w_last;
    g += Sys::term(x, u, p) * iw;
    if (gw) { for (int c = 0; c < Sys::NW; ++c) gw[c] += tg[c] * iw; }
  } else { (void)x; (void)u; (void)p; (void)w_last; (void)g; (void)gw; }
}
""")
  parts.append("}  // namespace myriad")
  with open(OUT, "w") as fh:
    fh.write("\n".join(parts) + "\n")
  print("wrote", os.path.abspath(OUT))


if __name__ == "__main__":
  main()
