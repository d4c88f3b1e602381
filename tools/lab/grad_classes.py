"""Prints, per tensor, the oracle gradient's size relative to its network's largest and the HIP error (test_gpu_sizes)."""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import test_gpu_sizes as T
from building_gan_amd.graph import detach_batch
from building_gan_amd.synthetic import batch_sizes, make_batch
from building_gan_amd.train import Trainer, global_counts
from oracle.objectives import train_iteration
cuda = torch.device("cuda:0")
for preset, graphs, aux in (("small", None, (0.0, 0.0)), ("dataset", 2, (0.0, 0.0)), ("small", None, (1.0, 1.0))):
    cpu = make_batch(preset, seed=61, graphs=graphs, ragged=True if preset == "small" else "jitter")
    sz = batch_sizes(cpu)
    Go, Do, _, _ = T._nets(cuda, seed=61)
    tr = Trainer(cuda, tr_weight=aux[0], far_weight=aux[1])
    tr.generator.load_state_dict(Go.state_dict()); tr.discriminator.load_state_dict(Do.state_dict())
    torch.manual_seed(61)
    noise, pz, vz, u = T._noise(sz), torch.rand(sz["Np"], 32), torch.rand(sz["Nv"], 32), torch.rand(sz["Nv"], 1)
    og, od = torch.optim.SGD(Go.parameters(), lr=0.0), torch.optim.SGD(Do.parameters(), lr=0.0)
    train_iteration(Go, Do, og, od, cpu, detach_batch, 0, pz, vz, noise, u, tr_weight=aux[0], far_weight=aux[1])
    gpu = copy.deepcopy(cpu).to(cuda)
    counts = global_counts(gpu)
    gen, sh = tr._phase_d(gpu, counts, True, pz.to(cuda), vz.to(cuda), T._to(noise, cuda), u.to(cuda))
    d_grads = T._flat_to_named(tr.opt_d, tr.discriminator)
    tr._phase_g(gpu, counts, gen)
    g_grads = T._flat_to_named(tr.opt_g, tr.generator)
    print("==", preset, aux)
    for tag, grads, net in (("D", d_grads, Do), ("G", g_grads, Go)):
        top = max(float(p.grad.abs().max()) for p in net.parameters())
        for n, p in net.named_parameters():
            own = float(p.grad.abs().max()); err = float((grads[n].double() - p.grad.double()).abs().max())
            print("  %s %-42s own/top %.2e  err/own %.2e  err/top %.2e" % (tag, n, own / top, err / max(own, 1e-300), err / top))
