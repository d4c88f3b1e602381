"""Diagnosis aid: the generator's one-call forward (optionally step by step: BG_DRIVER_SYNC=1) against the op-by-op path."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import models as hm
from building_gan_amd.synthetic import make_batch, batch_sizes
from building_gan_amd.util_eval import draw_z
dev = torch.device("cuda:0")
torch.manual_seed(0)
G = hm.Generator(7, 7, 128, 32, 4, 12, dev).to(dev)
order = sys.argv[1] if len(sys.argv) > 1 else "one-first"
for seed in (1, 2):
    b = make_batch("small", seed=seed, ragged=True)
    b.to(dev)
    sz = batch_sizes(b)
    print(sz, flush=True)
    with torch.no_grad():
        pz, vz = draw_z(b, 32, dev, 1.0, G)
        for fused in ((True, False) if order == "one-first" else (False, True)):
            hm.FUSE_GENERATOR_FORWARD = fused
            r = G(b, pz, vz)
            torch.cuda.synchronize(); print("fused=%s ok" % fused, float(r[1].sum()), int(r[4].sum()), flush=True)
