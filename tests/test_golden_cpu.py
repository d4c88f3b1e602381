"""The oracle against the committed golden vectors (released-checkpoint weights, tiny batch)."""
import torch

from golden_util import load_golden


def test_oracle_reproduces_golden_outputs():
    from building_gan_amd.synthetic import make_batch
    from oracle.networks import Generator
    weights, noise, ref = load_golden()
    assert len(weights) == 22            # the 21 parameter tensors the released code defines + the `pe` buffer
    G = Generator(7, 7, 128, 32, 4, 12)
    G.load_state_dict(weights)
    batch = make_batch("tiny", seed=0)
    with torch.no_grad():
        out, soft_out, mask, att, idx = G(batch, ref["program_z"], ref["voxel_z"], noise)
    assert torch.equal(idx, ref["max_out_program_index"])
    torch.testing.assert_close(out, ref["out"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(soft_out, ref["soft_out"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(att["soft"], ref["att_soft"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(mask["soft"], ref["mask_soft"], rtol=1e-5, atol=1e-6)
