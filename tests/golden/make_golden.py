"""Generates tests/golden/ckpt70_tiny.npz (run once in the build container, where /root/reference exists).

Contents: the 21 generator parameter tensors (and the pe buffer) of the released checkpoint runs/iccv2021/checkpoints/70.ckpt
that the released code defines (loaded with weights_only=True; the checkpoint's extra keys are
dropped exactly as inference.py:74-76 does with strict=False), the noise for one synthetic "tiny"
batch (seed 0) and the CPU oracle's generator outputs on it.  The batch itself is regenerated
from its seed by building_gan_amd.synthetic.make_batch("tiny", seed=0).

The reference's own code never runs here (its dependencies are not installed): the expected
outputs come from the oracle, so this fixture pins the oracle against regressions and gives the
HIP path a checkpoint-weights case; it does not pin parity with the reference (parity unpinned).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from building_gan_amd.synthetic import batch_sizes, make_batch  # noqa: E402
from oracle.networks import Generator  # noqa: E402


def main():
    sd = torch.load("/root/reference/runs/iccv2021/checkpoints/70.ckpt", weights_only=True, map_location="cpu")
    G = Generator(7, 7, 128, 32, 4, 12)
    missing = G.load_state_dict(sd, strict=False)
    assert not missing.missing_keys
    used = {k: v.numpy() for k, v in G.state_dict().items()}
    batch = make_batch("tiny", seed=0)
    sz = batch_sizes(batch)
    g = torch.Generator().manual_seed(1234)
    pz = torch.rand(sz["Np"], 32, generator=g)
    vz = torch.rand(sz["Nv"], 32, generator=g)
    noise = [(torch.empty(sz["Nv"], 2).exponential_(generator=g), torch.empty(sz["Ec"]).exponential_(generator=g))
             for _ in range(6)]
    with torch.no_grad():
        out, soft_out, mask, att, idx = G(batch, pz, vz, noise)
    arrays = {"w::" + k: v for k, v in used.items()}
    arrays.update(program_z=pz.numpy(), voxel_z=vz.numpy(), out=out.numpy(), soft_out=soft_out.numpy(),
                  mask_hard=mask["hard"].numpy(), mask_soft=mask["soft"].numpy(), att_hard=att["hard"].numpy(),
                  att_soft=att["soft"].numpy(), max_out_program_index=idx.numpy())
    for i, (a, b) in enumerate(noise):
        arrays["noise_mask_%d" % i] = a.numpy()
        arrays["noise_att_%d" % i] = b.numpy()
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ckpt70_tiny.npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path), "bytes;", len(used), "tensors")


if __name__ == "__main__":
    main()
