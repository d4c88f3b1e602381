"""The HIP generator with the released checkpoint's weights against the committed golden vectors."""
import copy

import pytest
import torch

from golden_util import load_golden

pytestmark = pytest.mark.gpu


def test_hip_generator_matches_golden(cuda):
    from building_gan_amd import models as hm
    from building_gan_amd.synthetic import make_batch
    weights, noise, ref = load_golden()
    G = hm.Generator(7, 7, 128, 32, 4, 12, cuda)
    res = G.load_state_dict(weights, strict=False)
    assert not res.missing_keys and not res.unexpected_keys
    G = G.to(cuda)
    batch = copy.deepcopy(make_batch("tiny", seed=0)).to(cuda)
    with torch.no_grad():
        out, soft_out, mask, att, idx = G(batch, ref["program_z"].to(cuda), ref["voxel_z"].to(cuda),
                                          [(a.to(cuda), b.to(cuda)) for a, b in noise])
    assert torch.equal(idx.cpu(), ref["max_out_program_index"])           # bit-exact pointer
    assert torch.equal(mask["hard"].round().cpu(), ref["mask_hard"].round())
    torch.testing.assert_close(soft_out.cpu(), ref["soft_out"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(att["soft"].cpu(), ref["att_soft"], rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(out.cpu(), ref["out"], rtol=1e-4, atol=1e-5)
