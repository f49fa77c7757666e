import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_node as T
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
hp, node, opt = T._setup(N)
rng = np.random.default_rng(0)
z = opt.guess + 0.3 * rng.standard_normal(opt.guess.size)
print("calling eval", flush=True)
r = opt.engine.eval(z[None], params=opt.system.device_params(), want=("c", "jblk", "f", "gradf"))
print("c", np.abs(r["c"]).max(), "f", r["f"], "jblk", np.abs(r["jblk"]).max(), flush=True)
os.environ["MYRIAD_HIP_LIB"] = ""
