"""Per-system extragradient step sizes (/root/reference/myriad/defaults.py:5-18; the parameter-guess tables of that
file belong to the sysid experiments and are outside the path)."""
from myriad_amd.systems import SystemType

learning_rates = {
  SystemType.CANCERTREATMENT: {'eta_x': 1e-1, 'eta_v': 1e-3},
  SystemType.CARTPOLE: {'eta_x': 1e-2, 'eta_v': 1e-4},
}
