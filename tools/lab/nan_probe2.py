import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from building_gan_amd import ops
from building_gan_amd.synthetic import make_batch
from building_gan_amd.train import Trainer
cuda = torch.device("cuda:0")
batch = make_batch("small", seed=3, ragged=True).to(cuda)
mode = sys.argv[1]
torch.manual_seed(0)
tr = Trainer(cuda)
tr.capture(batch)
od = tr.opt_d
plan = batch.__dict__["_bg_plan"]
gA, gB, gC, stats = tr._graphs["with_g"]
gen = tr._keep[0]
def cs(t):
    return "%.10e" % float(t.detach().double().sum()) if t is not None else "None"
def dump(tag):
    torch.cuda.synchronize()
    items = {"out": gen[0], "soft_out": gen[1], "mask_hard": gen[2]["hard"], "att_hard": gen[3]["hard"], "d_flat": od.flat, "d_grad": od.grad,
             "d_m": od.m, "g_flat": tr.opt_g.flat, "zeros": ops._ZEROS.get(cuda)}
    for k, v in plan.ops.items():
        if torch.is_tensor(v):
            items["plan.ops[%s]" % (k,)] = v
    for k, v in (tr._cw_cache or {}).items():
        items["cw%s" % (k,)] = torch.cat([v[0], v[1]])
    for k, v in items.items():
        print(tag, "%-46s %s ptr %x" % (k, cs(v), v.data_ptr() if v is not None else 0))
gA.replay()
if mode == "bad":
    torch.cuda.synchronize()
    g0 = od.grad.clone()
    y = bool(torch.isfinite(g0).all())
dump("before B")
gB.replay(); torch.cuda.synchronize()
print("g_adv %.5f" % float(stats["g_adv"]))
