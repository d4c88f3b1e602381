import os

import numpy as np
import torch

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ckpt70_tiny.npz")


def load_golden():
    z = np.load(PATH, allow_pickle=False)
    weights = {k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")}
    noise = [(torch.from_numpy(z["noise_mask_%d" % i]), torch.from_numpy(z["noise_att_%d" % i])) for i in range(6)]
    other = {k: torch.from_numpy(z[k]) for k in z.files if not k.startswith("w::") and not k.startswith("noise_")}
    return weights, noise, other
