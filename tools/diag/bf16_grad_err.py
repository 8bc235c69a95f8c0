import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np, torch
from conftest import sd_from_npz
from oracle import tcs as otcs
from thunder_speech_amd import train_ops
from thunder_speech_amd.quartznet.blocks import QuartznetBlock
g = np.load(os.path.join(os.path.dirname(__file__), "..", "..", "tests", "golden", "blocks.npz"))
spec = otcs.BlockSpec(in_ch=16, out_ch=32, repeat=3, kernel=11)
sd = sd_from_npz(g, "qn_res_k11/sd/")
x, lengths = torch.from_numpy(g["qn_res_k11/x"]), torch.from_numpy(g["qn_res_k11/lengths"])
cot = torch.randn(g["qn_res_k11/y_train"].shape, generator=torch.Generator().manual_seed(1)).cuda()
def run():
    blk = QuartznetBlock(16, 32, repeat=3, kernel_size=(11,), separable=True); blk.load_state_dict(sd); blk = blk.cuda().train()
    xg = x.clone().cuda().requires_grad_(True)
    y, _ = blk(xg, lengths.cuda()); (y * cot).sum().backward()
    return y.detach(), xg.grad, {k: p.grad for k, p in blk.named_parameters()}
y0, gx0, gp0 = run(); train_ops.set_gemm_precision("bf16"); y1, gx1, gp1 = run()
rel = lambda a, b: float((a - b).abs().max()) / max(float(b.abs().max()), 1e-3)
rms = lambda a, b: float((a - b).pow(2).mean().sqrt()) / max(float(b.pow(2).mean().sqrt()), 1e-6)
print("y", rel(y1, y0), rms(y1, y0)); print("gx", rel(gx1, gx0), rms(gx1, gx0))
for k in gp0: print(k, rel(gp1[k], gp0[k]), rms(gp1[k], gp0[k]), float(gp0[k].abs().max()))
