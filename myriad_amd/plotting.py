"""Minimal presentation helper for run_trajectory_opt (the reference's plotting.py is OUT OF SCOPE, SURVEY.md section 2
row 20; this exists so that the default cfg.plot=True of useful_scripts.run_trajectory_opt produces a figure instead
of an import error).  One panel per state and per control over time, cost and defect in the title."""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np


def plot(hp, system, data: Dict[str, np.ndarray], save_as: Optional[str] = None):
  import matplotlib
  matplotlib.use("Agg")
  import matplotlib.pyplot as plt
  x, u = np.asarray(data['x']), np.asarray(data['u'])
  panels = [("x%d" % i, x[:, i]) for i in range(x.shape[1])] + [("u%d" % i, u[:, i]) for i in range(u.shape[1])]
  fig, axes = plt.subplots(len(panels), 1, figsize=(6, 1.6 * len(panels)), sharex=False)
  for ax, (name, ys) in zip(np.atleast_1d(axes), panels):
    ax.plot(np.linspace(0.0, system.T, len(ys)), ys)
    ax.set_ylabel(name)
    ax.grid(True)
  np.atleast_1d(axes)[-1].set_xlabel("time")
  bits = [hp.system.name]
  if 'cost' in data and data['cost'] is not None:
    bits.append("cost %.6g" % float(data['cost']))
  if data.get('defect') is not None:
    bits.append("|defect| %.3g" % float(np.abs(np.asarray(data['defect'])).max()))
  fig.suptitle(", ".join(bits))
  fig.tight_layout()
  if save_as is not None:
    fig.savefig(save_as if save_as.endswith((".png", ".pdf", ".svg")) else save_as + ".png")
  plt.close(fig)
  return fig
