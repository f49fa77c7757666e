#!/usr/bin/env python3
"""Fit the (64,64) sigmoid MLP of BASELINE config 5 to the true CARTPOLE vector field and commit the weights.

The reference trains its NODE with Adam on trajectories (myriad/neural_ode/node_training.py, out of scope); what the
hot path needs is ONE plausible weight set of the right architecture (create_node.py:110-117: hk.Linear(64)+sigmoid,
hk.Linear(64)+sigmoid, hk.Linear(4); y = x @ w + b) so that `solve_with_params(node.params)` plans through the network
("one shared weight set fitted on the host to the true cart-pole field and committed").
Output: myriad_amd/data/node_cartpole_64x64.npz with Haiku's key layout {linear, linear_1, linear_2} x {w, b}.
Run: python tools/fit_node_cartpole.py
"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from myriad_amd.systems import CartPole

torch.manual_seed(2019)
torch.set_default_dtype(torch.float64)
s = CartPole()
lo = torch.tensor([-2.0, -2 * np.pi, -5.0, -10.0, -20.0]); hi = -lo

def field(w):
  th, dx, dth, u = w[:, 1], w[:, 2], w[:, 3], w[:, 4]
  sn, cs = torch.sin(th), torch.cos(th)
  ddx = (s.length * s.m2 * sn * dth ** 2 + u + s.m2 * s.g * cs * sn) / (s.m1 + s.m2 * (1 - cs ** 2))
  ddth = -((s.length * s.m2 * cs * dth ** 2 + u * cs + (s.m1 + s.m2) * s.g * sn) / (s.length * s.m1 + s.length * s.m2 * (1 - cs ** 2)))
  return torch.stack([dx, dth, ddx, ddth], dim=1)

n = 40000
W = lo + (hi - lo) * torch.rand(n, 5)
Y = field(W)
net = torch.nn.Sequential(torch.nn.Linear(5, 64), torch.nn.Sigmoid(), torch.nn.Linear(64, 64), torch.nn.Sigmoid(), torch.nn.Linear(64, 4))
opt = torch.optim.Adam(net.parameters(), lr=3e-3)
for it in range(4000):
  idx = torch.randint(0, n, (2048,))
  loss = ((net(W[idx]) - Y[idx]) ** 2).mean()
  opt.zero_grad(); loss.backward(); opt.step()
  if it % 1000 == 0: print(it, float(loss))
lb = torch.optim.LBFGS(net.parameters(), max_iter=300, history_size=30, line_search_fn="strong_wolfe")
def closure():
  lb.zero_grad(); l = ((net(W) - Y) ** 2).mean(); l.backward(); return l
lb.step(closure)
print("final mse", float(((net(W) - Y) ** 2).mean()), "field var", float(Y.var()))
lin = [m for m in net if isinstance(m, torch.nn.Linear)]
out = {}
for name, m in zip(["linear", "linear_1", "linear_2"], lin):
  out[name + "/w"] = m.weight.detach().numpy().T.copy()      # Haiku stores (in, out)
  out[name + "/b"] = m.bias.detach().numpy().copy()
path = os.path.join(ROOT, "myriad_amd", "data", "node_cartpole_64x64.npz")
np.savez_compressed(path, **out)
print("wrote", path, os.path.getsize(path))
