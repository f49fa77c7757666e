"""Drop-in for /root/reference/run.py: `python run.py --system=CARTPOLE --optimizer=COLLOCATION
--quadrature_rule=HERMITE_SIMPSON --integration_method=RK4 --intervals=100` (README.md:82-85 flag syntax)."""
import random

import numpy as np

from myriad_amd.useful_scripts import run_setup, run_trajectory_opt


def main():
  hp, cfg = run_setup()
  random.seed(hp.seed)             # run.py:25-26
  np.random.seed(hp.seed)
  c, defect = run_trajectory_opt(hp, cfg, save_as='traj_opt_example.pdf')
  print("Cost given by integrating the control trajectory:", c)
  print("Defect:", defect)


if __name__ == '__main__':
  main()
