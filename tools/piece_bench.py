"""hipGraph replay time of the pieces of an iteration (tuning aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd.graph import detach_batch
from building_gan_amd.plan import build_plan
from building_gan_amd.synthetic import make_batch, batch_sizes
from building_gan_amd.train import Trainer, global_counts

dev = torch.device("cuda:0")
batch = make_batch("dataset", seed=0).to(dev)
sz = batch_sizes(batch)
torch.manual_seed(0)
tr = Trainer(dev)
plan = build_plan(batch)
counts = global_counts(batch, None)
G, D = tr.generator, tr.discriminator
pz = torch.rand(sz["Np"], 32, device=dev); vz = torch.rand(sz["Nv"], 32, device=dev)

def timeit(name, fn, iters=20):
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): fn()
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): g.replay()
    b.record(); torch.cuda.synchronize()
    print("%-44s %8.3f ms" % (name, a.elapsed_time(b) / iters), flush=True)
    return keep

def g_fwd_nograd():
    with torch.no_grad():
        return G(batch, pz, vz)
timeit("G forward (no_grad)", g_fwd_nograd)
gen = timeit("G forward (autograd graph)", lambda: G(batch, pz, vz))
out, soft = gen[0].detach(), gen[1].detach()
def d_fwd():
    with torch.no_grad():
        return D(batch, out)
timeit("D forward fake (no_grad)", d_fwd)
def d_fb():
    tr.opt_d.zero_grad()
    b, s = D(batch, batch.voxel_label)
    (b.sum() + s.sum()).backward()
timeit("D forward+backward real", d_fb)
def gp():
    tr.opt_d.zero_grad()
    v = tr._gradient_penalty(detach_batch(batch), batch.voxel_label.detach(), soft, None, sz["Nv"])
    v.backward()
timeit("gradient penalty fwd + double backward", gp)
timeit("phase D (G fwd + 3 D passes + backward)", lambda: tr._phase_d(batch, counts, True))
timeit("Adam(D)", lambda: tr.opt_d.step())
def phase_g():
    gen = G(batch, pz, vz)
    return tr._phase_g(batch, counts, gen)
timeit("G fwd + phase G (D(fake) + backward to G)", phase_g)
timeit("Adam(G)", lambda: tr.opt_g.step())
