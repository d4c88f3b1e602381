import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from building_gan_amd.synthetic import make_batch
from building_gan_amd.train import Trainer, global_counts
cuda = torch.device("cuda:0")
batch = make_batch("small", seed=3, ragged=True).to(cuda)
for mode in sys.argv[1:]:
    torch.manual_seed(0)
    tr = Trainer(cuda)
    tr.capture(batch)
    od = tr.opt_d
    gA, gB, gC, stats = tr._graphs["with_g"]
    gA.replay()
    if mode == "synconly":
        torch.cuda.synchronize()
    elif mode == "clone_small":
        x = torch.empty(16, device=cuda).fill_(1.0)
    elif mode == "clone_big":
        x = torch.empty(1 << 22, device=cuda).fill_(1.0)
    elif mode == "alloc_only":
        x = torch.empty(1 << 22, device=cuda)
    elif mode == "item":
        y = float(stats["d_loss"])
    elif mode == "rand":
        x = torch.rand(16, device=cuda)
    elif mode == "clone_grad":
        g0 = od.grad.clone()
    elif mode == "clone_flat":
        p0 = od.flat.clone()
    elif mode == "absmax":
        y = float(od.grad.abs().max())
    elif mode == "isfinite":
        y = bool(torch.isfinite(od.grad).all())
    elif mode.startswith("poison"):
        # fill every free block the default pool would hand out for requests of this size with NaN
        n = int(mode[6:])
        xs = [torch.full((n,), float("nan"), device=cuda) for _ in range(24)]
        del xs
    elif mode == "c1":
        torch.cuda.synchronize()
        g0, p0 = od.grad.clone(), od.flat.clone()
    elif mode == "c2":
        g0, p0 = od.grad.clone(), od.flat.clone()
        y = float(g0.abs().max())
    elif mode == "c3":
        g0 = od.grad.clone()
        y = bool(torch.isfinite(g0).all())
    elif mode == "c4":
        g0, p0 = od.grad.clone(), od.flat.clone()
        y = (float(g0.abs().max()), bool(torch.isfinite(g0).all()), float(p0.abs().max()))
    elif mode == "c5":
        y = (float(od.grad.abs().max()), bool(torch.isfinite(od.grad).all()), float(od.flat.abs().max()))
    elif mode == "c6":
        print("x")
        y = (float(od.grad.abs().max()), float(od.flat.abs().max()))
    elif mode.startswith("spoison"):
        torch.cuda.synchronize()
        n = int(mode[7:])
        xs = [torch.full((n,), float("nan"), device=cuda) for _ in range(int(os.environ.get("NPOISON", "24")))]
        del xs
    elif mode == "c7":
        torch.cuda.synchronize()
        g0, p0 = od.grad.clone(), od.flat.clone()
        y = (float(g0.abs().max()), bool(torch.isfinite(g0).all()), float(p0.abs().max()))
    elif mode == "c8":
        g0, p0 = od.grad.clone(), od.flat.clone()
        print("after A: grad max %.4e finite %s | flat max %.4f" % (float(g0.abs().max()), bool(torch.isfinite(g0).all()), float(p0.abs().max())))
    elif mode == "c9":
        torch.cuda.synchronize()
        g0 = od.grad.clone()
        y = bool(torch.isfinite(g0).all())
    elif mode == "c10":
        torch.cuda.synchronize()
        y = float(od.flat.abs().max())
    elif mode == "full":
        torch.cuda.synchronize()
        g0, p0 = od.grad.clone(), od.flat.clone()
        print("after A: grad max %.4e finite %s | flat max %.4f" % (float(g0.abs().max()), bool(torch.isfinite(g0).all()), float(p0.abs().max())))
    gB.replay(); torch.cuda.synchronize()
    print("%-12s g_adv %.5f" % (mode, float(stats["g_adv"])))
    del tr
