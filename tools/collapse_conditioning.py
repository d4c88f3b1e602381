"""Calibration of DiscriminatorVoxel.guard_collapse: the discriminator's block weights scaled by 1, 2, 4, 8, 16;
collapsed (polynomial) and layer-by-layer fp32 evaluations against the oracle in float64 on dataset-sized graphs,
next to the measured amplification.  Prints one line per scale."""
import copy, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from building_gan_amd import models as hm
from building_gan_amd.synthetic import make_batch
from oracle import networks as om

dev = torch.device("cuda:0")
cpu = make_batch("dataset", seed=31, graphs=2, ragged="jitter")
cpu64 = copy.deepcopy(cpu); cpu64.voxel_feature = cpu64.voxel_feature.double()
gpu = copy.deepcopy(cpu).to(dev)
for scale in (1, 2, 4, 8, 16):
    torch.manual_seed(31)
    Do = om.DiscriminatorVoxel(7, 6, 128, 12)
    with torch.no_grad():
        for p in Do.voxelGNN_D.block.parameters():
            p.mul_(scale)
    Dh = hm.DiscriminatorVoxel(7, 6, 128, 12).to(dev)
    Dh.load_state_dict(Do.state_dict())
    D64 = copy.deepcopy(Do).double()
    with torch.no_grad():
        v64 = D64.voxelGNN_D(cpu64.voxel_feature, cpu.voxel_label.double(), cpu.voxel_edge, cpu.voxel_feature_batch,
                             cpu.voxel_floor_cluster, cpu.voxel_feature_batch)
        b64, s64 = D64(cpu64, cpu.voxel_label.double())
        errs = {}
        for name, flag in (("collapsed", True), ("stack", False)):
            hm.COLLAPSE_D = flag
            gpu.__dict__.pop("_bg_plan", None)
            from building_gan_amd.plan import build_plan
            v = Dh.voxelGNN_D(gpu.voxel_feature, gpu.voxel_label, build_plan(gpu)).cpu().double()
            bh, sh = Dh(gpu, gpu.voxel_label)
            e_v = float(((v - v64).abs().amax(0) / v64.abs().amax(0)).max())
            e_o = max(float((bh.cpu().double() - b64).abs().max() / b64.abs().max()),
                      float((sh.cpu().double() - s64).abs().max() / s64.abs().max()))
            errs[name] = (e_v, e_o)
        hm.COLLAPSE_D = True
        amp = Dh.guard_collapse(gpu)
    print("scale %2d: amplification %.3e (eps*amp = %.2e, collapse_ok=%s) | v_L error per column max: collapsed %.2e stack %.2e | "
          "outputs: collapsed %.2e stack %.2e" % (scale, amp, amp * 2.0 ** -24, Dh.voxelGNN_D.collapse_ok,
                                                   errs["collapsed"][0], errs["stack"][0], errs["collapsed"][1], errs["stack"][1]), flush=True)
