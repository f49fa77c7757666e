"""Independent pin for the oracle: continuous-time optima of two Lenhart systems from
Pontryagin's conditions solved as two-point boundary value problems by scipy.integrate.solve_bvp -- no transcription,
no NLP solver, no code shared with the oracle or the device.  Conditions (minimisation form, H = g + p f):

  SIMPLECASE       (lenhart/simple_case.py:25-62)      x' = -x^2/2 + C u,  g = -A x + B u^2
      u* = -C p / (2 B);   p' = A + x p;   x(0) = x0, p(T) = 0
  CANCERTREATMENT  (lenhart/cancer_treatment.py:40-91) x' = r x ln(1/x) - u delta x,  g = a x^2 + u^2,  0 <= u <= 2
      u* = clip(p delta x / 2, 0, 2);   p' = -(2 a x + p (r ln(1/x) - r - u delta));   x(0) = x0, p(T) = 0

Writes tests/golden/pontryagin_bvp.json: optimal cost (by high-order quadrature of the BVP solution), x(T), u(0), u(T/2)
and the solver's residual.  The Hermite-Simpson optimum at N intervals must approach these with O(h^4)."""
import json
import os

import numpy as np
from scipy.integrate import solve_bvp, simpson

out = {}

# ---- SIMPLECASE: A = 1, B = 1, C = 4, x0 = 1, T = 1 (simple_case.py:27-33)
A, B, C, x0, T = 1.0, 1.0, 4.0, 1.0, 1.0
def f(t, y):
  x, p = y
  u = -C * p / (2 * B)
  return np.vstack([-0.5 * x ** 2 + C * u, A + x * p])
def bc(ya, yb):
  return np.array([ya[0] - x0, yb[1]])
t = np.linspace(0, T, 401)
sol = solve_bvp(f, bc, t, np.vstack([np.ones_like(t), np.zeros_like(t)]), tol=1e-11, max_nodes=200000)
assert sol.success
tt = np.linspace(0, T, 20001); x, p = sol.sol(tt); u = -C * p / (2 * B)
out["SIMPLECASE"] = {"params": {"A": A, "B": B, "C": C, "x0": x0, "T": T}, "cost": float(simpson(-A * x + B * u ** 2, x=tt)),
                     "x_T": float(x[-1]), "u_0": float(u[0]), "u_mid": float(u[len(u) // 2]), "rms_residual": float(np.max(sol.rms_residuals))}

# ---- CANCERTREATMENT: r = 0.3, a = 3, delta = 0.45, x0 = 0.975, T = 20 (cancer_treatment.py:40-49)
r, a, d, x0, T = 0.3, 3.0, 0.45, 0.975, 20.0
def f2(t, y):
  x, p = y
  x = np.maximum(x, 1e-9)
  u = np.clip(p * d * x / 2, 0.0, 2.0)
  return np.vstack([r * x * np.log(1 / x) - u * d * x, -(2 * a * x + p * (r * np.log(1 / x) - r - u * d))])
def bc2(ya, yb):
  return np.array([ya[0] - x0, yb[1]])
# initial guess by continuation in the horizon (the T = 20 problem does not converge from a constant guess)
t = np.linspace(0, 1.0, 201)
guess = np.vstack([x0 * np.ones_like(t), np.zeros_like(t)])
for Th in (1.0, 2.0, 4.0, 7.0, 10.0, 14.0, 17.0, 20.0):
  tn = np.linspace(0, Th, 801)
  gx = np.interp(tn * (t[-1] / Th), t, guess[0]); gp = np.interp(tn * (t[-1] / Th), t, guess[1])
  sol = solve_bvp(f2, bc2, tn, np.vstack([gx, gp]), tol=1e-6 if Th < T else 1e-10, max_nodes=400000)
  assert sol.success, (Th, sol.message)
  t, guess = sol.x, sol.y
assert sol.success
tt = np.linspace(0, T, 40001); x, p = sol.sol(tt); u = np.clip(p * d * x / 2, 0.0, 2.0)
out["CANCERTREATMENT"] = {"params": {"r": r, "a": a, "delta": d, "x0": x0, "T": T}, "cost": float(simpson(a * x ** 2 + u ** 2, x=tt)),
                          "x_T": float(x[-1]), "u_0": float(u[0]), "u_mid": float(u[len(u) // 2]), "rms_residual": float(np.max(sol.rms_residuals))}

path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "pontryagin_bvp.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
