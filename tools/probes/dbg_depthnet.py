import os, sys, torch
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_zoedepth_gpu as T
from oracle import zoedepth_ref as Z
r = T.run_case(Z.ZOED_NK, torch.float16, B=1, H=480, W=640, target_hw=(384, 512), seed=2, precision="accurate", class_modes="wmean", attn_mode="single", neck_mode="full")
tp, to = r["taps_p"], r["taps_o"]
d = tp["depth_net"][0].float().cpu(); ref = to["depth_net"]
e = (d - ref).abs()
print("RELU_OUT", os.environ.get("BS_RELU_OUT"), "depth_net err: mean %.3e max %.3e; frac>1e-3: %.4f" % (e.mean(), e.max(), (e > 1e-3).float().mean()))
idx = (e > 1e-3).nonzero()
print(idx[:20].tolist(), idx.shape)
print("by image:", [float(e[i].mean()) for i in range(e.shape[0])])
