"""Two-level Riccati sweep: numpy model of the interface algebra (tools/dev; not shipped).
A staged QP  min sum_k 1/2 y_k^T Q_k y_k + l_k^T y_k,  y_k = (w_k, q_k),  w_{k+1} = F_k y_k + f_k,  terminal multipliers nuT on pinned rows,
solved (a) by one backward recursion over the augmented form Z over (w; theta), theta = (1, nuT), and (b) in W chunks that each start from
the terminal form  1/2 rho |w_e|^2 + nu^T w_e  (theta_c = (1, nu)), joined by the interface recursion of hs_solver_fused.h: two_level_join."""
import numpy as np

NW, NQ = 5, 2
rng = np.random.default_rng(0)

def chol_floor(A, floor):
  """detail::chol_reg of hs_solver.h: a pivot that is not above `floor` is replaced by max(|pivot|, floor) (unreachable directions of a short chunk)"""
  n = A.shape[0]; L = np.zeros_like(A)
  for j in range(n):
    d = A[j, j] - L[j, :j] @ L[j, :j]
    if not d > floor: d = max(abs(d), floor)
    L[j, j] = np.sqrt(d)
    for i in range(j + 1, n):
      L[i, j] = (A[i, j] - L[i, :j] @ L[j, :j]) / L[j, j]
  return L

def stage_step(Z, Q, l, F, f, nth):
  """Z over (w+; theta) -> Z over (w; theta) and the gains K | kc (q = -K w - kc theta)."""
  ny = NW + NQ
  Hs = np.zeros((ny + nth, ny + nth))
  Hs[:ny, :ny] = Q
  Hs[:ny, ny] = l; Hs[ny, :ny] = l
  T = np.zeros((NW + nth, ny + nth))
  T[:NW, :ny] = F; T[:NW, ny] = f
  T[NW:, ny:] = np.eye(nth)
  M = Hs + T.T @ Z @ T
  iw = list(range(NW)) + list(range(ny, ny + nth)); iq = list(range(NW, ny))
  Mqq = M[np.ix_(iq, iq)]; Mqr = M[np.ix_(iq, iw)]
  piv = np.linalg.eigvalsh(Mqq).min()
  G = np.linalg.solve(Mqq, Mqr)
  Zn = M[np.ix_(iw, iw)] - Mqr.T @ G
  return Zn, G, piv

def sweep(stages, Zterm, nth):
  Z = Zterm.copy(); gains = []; pmin = np.inf
  for (Q, l, F, f) in reversed(stages):
    Z, G, piv = stage_step(Z, Q, l, F, f, nth); gains.append(G); pmin = min(pmin, piv)
  return Z, gains[::-1], pmin

def make(N, convex=True):
  st = []
  for k in range(N):
    A = rng.standard_normal((NW + NQ, NW + NQ)) * 0.3
    Q = A @ A.T + (0.5 if convex else -0.05) * np.eye(NW + NQ)
    l = rng.standard_normal(NW + NQ)
    F = np.zeros((NW, NW + NQ)); F[:, :NW] = np.eye(NW) + 0.1 * rng.standard_normal((NW, NW)); F[:, NW:] = 0.3 * rng.standard_normal((NW, NQ))
    f = 0.1 * rng.standard_normal(NW)
    st.append((Q, l, F, f))
  return st

def terminal(pinned, rho):
  npin = len(pinned); nth = 1 + npin
  Z = np.zeros((NW + nth, NW + nth))
  for i, r in enumerate(pinned):
    Z[r, r] = rho; Z[r, NW + 1 + i] = 1.0; Z[NW + 1 + i, r] = 1.0
  return Z, nth

def first_point_and_nu(Z, nth, nfree0):
  """w_0 = (0 (pinned), u_0 free: the last nfree0 entries); eliminate u_0, then nuT from T rows = 0; returns theta_g, w_0"""
  iu = list(range(NW - nfree0, NW)); it = list(range(NW, NW + nth))
  Muu = Z[np.ix_(iu, iu)]; Mut = Z[np.ix_(iu, it)]
  Ku = np.linalg.solve(Muu, Mut)
  Tt = Z[np.ix_(it, it)] - Mut.T @ Ku
  # d/d nuT of the value = pinned terminal state = 0:  Tt[1:, 0] + Tt[1:, 1:] nu = 0
  nu = np.linalg.solve(Tt[1:, 1:], -Tt[1:, 0]) if nth > 1 else np.zeros(0)
  th = np.concatenate([[1.0], nu])
  w0 = np.zeros(NW); w0[iu] = -Ku @ th
  return th, w0, np.linalg.eigvalsh(Muu).min()

def rollout(stages, gains, w0, th_of_stage):
  w = w0.copy(); ws = [w0.copy()]; qs = []
  for k, (Q, l, F, f) in enumerate(stages):
    r = np.concatenate([w, th_of_stage(k)])
    q = -gains[k] @ r; qs.append(q)
    w = F @ np.concatenate([w, q]) + f; ws.append(w.copy())
  return np.array(ws), np.array(qs)

def two_level(stages, pinned, rho_t, rho, W):
  N = len(stages); nthg = 1 + len(pinned)
  edges = [round(c * N / W) for c in range(W + 1)]
  # chunk sweeps (parallel on the device)
  out = []
  for c in range(W):
    if c == W - 1:
      Zt, nth = terminal(pinned, rho_t)
    else:
      nth = 1 + NW
      Zt = np.zeros((NW + nth, NW + nth)); Zt[:NW, :NW] = rho * np.eye(NW); Zt[:NW, NW + 1:] = np.eye(NW); Zt[NW + 1:, :NW] = np.eye(NW)
    Z, G, pmin = sweep(stages[edges[c]:edges[c + 1]], Zt, nth)
    out.append((Z, G, pmin, nth))
  # interface recursion, last chunk first: true form Zt over (w_a; theta_g)
  Ztrue = out[-1][0]; join = [None] * W; pmin_if = np.inf
  for c in range(W - 2, -1, -1):
    Z = out[c][0]
    P = Z[:NW, :NW]; pc1 = Z[:NW, NW]; E = Z[:NW, NW + 1:]; t1 = Z[NW + 1:, NW]; Tnn = Z[NW + 1:, NW + 1:]; c00 = Z[NW, NW]
    Pp = Ztrue[:NW, :NW]; ptp = Ztrue[:NW, NW:]; Ttp = Ztrue[NW:, NW:]
    M = Pp - rho * np.eye(NW)
    A = np.eye(NW) - Tnn @ M
    S = -Tnn
    L = chol_floor(S, 1e-14 * max(np.abs(np.diag(S)).max(), 1e-300))
    C = np.eye(NW) + L.T @ M @ L
    pmin_if = min(pmin_if, np.linalg.eigvalsh(C).min())
    Ai = np.linalg.inv(A)
    # w_e = Ai (E^T w_a + t1 e1^T theta_g + Tnn ptp theta_g)
    t1g = np.zeros((NW, nthg)); t1g[:, 0] = t1
    We_w = Ai @ E.T; We_t = Ai @ (t1g + Tnn @ ptp)
    # nu = M w_e + ptp theta_g
    Nu_w = M @ We_w; Nu_t = M @ We_t + ptp
    Zn = np.zeros((NW + nthg, NW + nthg))
    Zn[:NW, :NW] = P + E @ Nu_w
    pcg = np.zeros((NW, nthg)); pcg[:, 0] = pc1
    Zn[:NW, NW:] = pcg + E @ Nu_t; Zn[NW:, :NW] = Zn[:NW, NW:].T
    # theta block by the envelope theorem: d/d theta_g of the true value = ptp^T w_e + Ttp theta_g  (+ the chunk's own constant row for "1")
    Tg = ptp.T @ We_t + Ttp
    own = np.zeros((nthg, nthg)); own[0, 0] = c00; own[0, :] += t1 @ Nu_t; 
    Zn[NW:, NW:] = Tg + own
    join[c] = (We_w, We_t, Nu_w, Nu_t)
    Ztrue = Zn
  return out, join, Ztrue, edges, pmin_if

def compare(N=24, W=4, pinned=(0, 1, 3), convex=True, verbose=False):
  """two-level against the plain recursion on one random staged QP: largest differences of the value form, nu_T, the states w and the controls q"""
  pinned = list(pinned)
  if True:
    st = make(N, convex)
    Zt, nth = terminal(pinned, 1e4)
    Z, G, pmin = sweep(st, Zt, nth)
    th, w0, p0 = first_point_and_nu(Z, nth, 1)
    ws, qs = rollout(st, G, w0, lambda k: th)
    out, join, Ztrue, edges, pmin_if = two_level(st, pinned, 1e4, 1e4, W)
    th2, w02, p02 = first_point_and_nu(Ztrue, nth, 1)
    # forward over the interfaces: theta of every chunk
    ths = [None] * W; ths[W - 1] = th2
    wa = w02.copy()
    for c in range(W - 1):
      We_w, We_t, Nu_w, Nu_t = join[c]
      we = We_w @ wa + We_t @ th2; nu = Nu_w @ wa + Nu_t @ th2
      ths[c] = np.concatenate([[1.0], nu]); wa = we
    G2 = sum((o[1] for o in out), [])
    def th_of(k):
      c = max(i for i in range(W) if edges[i] <= k)
      return ths[c]
    ws2, qs2 = rollout(st, G2, w02, th_of)
    if verbose:
      print("convex" if convex else "indefinite", "min pivot sequential %.3g | chunks %s interface %.3g" % (min(pmin, p0), ["%.3g" % o[2] for o in out], pmin_if))
      print("  Z(P,pc) diff %.3e  nuT diff %.3e  w diff %.3e  q diff %.3e  pinned terminal %.2e" % (
        np.abs(Ztrue[:NW, :] - Z[:NW, :]).max(), np.abs(th - th2).max(), np.abs(ws - ws2).max(), np.abs(qs - qs2).max(), np.abs(ws2[-1][pinned]).max()))
    return dict(dZ=np.abs(Ztrue[:NW, :] - Z[:NW, :]).max(), dnu=np.abs(th - th2).max(), dw=np.abs(ws - ws2).max(), dq=np.abs(qs - qs2).max(),
                pinned=np.abs(ws2[-1][pinned]).max() if pinned else 0.0, pmin_seq=min(pmin, p0), pmin_chunks=min(o[2] for o in out), pmin_if=pmin_if,
                scale=max(1.0, np.abs(ws).max(), np.abs(qs).max()))


if __name__ == "__main__":
  for convex in (True, False):
    compare(convex=convex, verbose=True)
