"""A/B of two builds of the library on the dataset-size gradient case: dumps the generator's outputs and flat
gradients (run once per BG_LIB_PATH), or compares two dumps.
    python tools/lab/tanh_ab.py dump /tmp/a.pt        python tools/lab/tanh_ab.py cmp /tmp/a.pt /tmp/b.pt"""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch

if sys.argv[1] == "dump":
    from building_gan_amd.synthetic import batch_sizes, make_batch
    from building_gan_amd.train import Trainer, global_counts
    from oracle import networks as om
    cuda = torch.device("cuda:0")
    seed = 61
    cpu = make_batch("dataset", seed=seed, graphs=2, ragged="jitter")
    sz = batch_sizes(cpu)
    torch.manual_seed(seed)
    Go, Do = om.Generator(7, 7, 128, 32, 4, 12), om.DiscriminatorVoxel(7, 6, 128, 12)
    tr = Trainer(cuda)
    tr.generator.load_state_dict(Go.state_dict())
    tr.discriminator.load_state_dict(Do.state_dict())
    torch.manual_seed(seed)
    noise = [(torch.empty(sz["Nv"], 2).exponential_(), torch.empty(sz["Ec"]).exponential_()) for _ in range(6)]
    pz, vz, u = torch.rand(sz["Np"], 32), torch.rand(sz["Nv"], 32), torch.rand(sz["Nv"], 1)
    gpu = copy.deepcopy(cpu).to(cuda)
    counts = global_counts(gpu)
    nz = [(a.to(cuda), b.to(cuda)) for a, b in noise]
    gen, sh = tr._phase_d(gpu, counts, True, pz.to(cuda), vz.to(cuda), nz, u.to(cuda))
    sh.update(tr._phase_g(gpu, counts, gen))
    names = [n for n, _ in tr.generator.named_parameters()]
    grads = {n: tr.opt_g.grad[off:off + p.numel()].detach().cpu().clone() for n, p, off in zip(names, tr.opt_g.params, tr.opt_g.offsets)}
    torch.save({"out": gen[0].detach().cpu(), "soft_out": gen[1].detach().cpu(), "mask_soft": gen[2]["soft"].detach().cpu(),
                "mask_hard": gen[2]["hard"].detach().cpu(), "att_soft": gen[3]["soft"].detach().cpu(),
                "att_hard": gen[3]["hard"].detach().cpu(), "idx": gen[4].cpu(), "grads": grads,
                "g_loss": sh["g_loss"].cpu()}, sys.argv[2])
else:
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in ("out", "soft_out", "mask_soft", "mask_hard", "att_soft", "att_hard", "idx", "g_loss"):
        d = (a[k].double() - b[k].double()).abs()
        print("%-10s max |a-b| %.3e   differing entries %d of %d" % (k, float(d.max()), int((d > 0).sum()), d.numel()))
    top = max(float(v.abs().max()) for v in a["grads"].values())
    for n in a["grads"]:
        d = float((a["grads"][n].double() - b["grads"][n].double()).abs().max())
        print("%-45s own max %.3e  |a-b| / own %.3e  / top %.3e" % (n, float(a["grads"][n].abs().max()), d / float(a["grads"][n].abs().max()), d / top))
