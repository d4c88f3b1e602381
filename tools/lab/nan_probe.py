import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from building_gan_amd.synthetic import make_batch
from building_gan_amd.train import Trainer
cuda = torch.device("cuda:0")
batch = make_batch("small", seed=3, ragged=True).to(cuda)
torch.manual_seed(0)
tr = Trainer(cuda)
tr.capture(batch)
fin = lambda t: bool(torch.isfinite(t).all())
for i in range(11):
    if i % 5 == 0:
        gA, gB, gC, stats = tr._graphs["with_g"]
        gA.replay(); torch.cuda.synchronize()
        print(i, "after A : gp %.5f d_loss %.5f | D grad finite %s | G flat finite %s D flat finite %s" % (float(stats["gp"]), float(stats["d_loss"]), fin(tr.opt_d.grad), fin(tr.opt_g.flat), fin(tr.opt_d.flat)))
        gB.replay(); torch.cuda.synchronize()
        print(i, "after B : g_loss %.5f | G grad finite %s | D flat finite %s" % (float(stats["g_loss"]), fin(tr.opt_g.grad), fin(tr.opt_d.flat)))
        gC.replay(); torch.cuda.synchronize()
        print(i, "after C : G flat finite %s derived finite %s" % (fin(tr.opt_g.flat), all(fin(t) for t in tr._g_derived if t is not None)))
    else:
        gA, gB, stats = tr._graphs["no_g"]
        gA.replay(); torch.cuda.synchronize()
        print(i, "after A2: gp %.5f d_loss %.5f | D grad finite %s" % (float(stats["gp"]), float(stats["d_loss"]), fin(tr.opt_d.grad)))
        gB.replay(); torch.cuda.synchronize()
        print(i, "after B2: D flat finite %s" % fin(tr.opt_d.flat))
