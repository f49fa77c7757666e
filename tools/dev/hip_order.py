# dev: the library first, torch.cuda afterwards, in one process (the order that used to break torch)
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from myriad_amd import _lib
eng = _lib.Engine("CARTPOLE", "HERMITE_SIMPSON", 4, 2.0)
print("eval f", eng.eval(np.zeros((1, eng.n)))["f"])
import torch
torch.cuda.init()
print("torch sees", torch.cuda.device_count(), "device(s);", torch.zeros(3, device="cuda").sum().item())
