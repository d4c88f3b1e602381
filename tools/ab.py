"""Same-process, same-box A/B of the stream step: alternating blocks of steps with a switch on / off, medians of the
blocks (boxes of the pool differ by several per cent, so only this kind of comparison resolves small effects).
    python tools/ab.py models.FUSE_PSTACK [models.FUSE_GSTACK ...] | prefetch"""
import argparse, os, statistics, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from building_gan_amd import models, ops, plan as planmod

args = argparse.Namespace(gpus=1, steps=40, warmup=10, batches=8, graphs=8, preset="dataset", no_replay=True,
                          no_cpu_baseline=True, no_microbench=True, seed=0)
w = bench.HipWorkload(args, 0, 1)
i = [0]


saved = [None]


def block(n=40):
    # every block starts from the same parameters / optimizer state and the same iteration index, so the blocks
    # differ only by the switch (and the trajectory never drifts to where the collapsed critic's guard trips)
    if saved[0] is None:
        saved[0] = (w.trainer.opt_g.state(), w.trainer.opt_d.state())
    else:
        w.trainer.opt_g.restore(saved[0][0]); w.trainer.opt_d.restore(saved[0][1]); w.trainer.refresh_derived()
    i[0] = 40
    w.sync()
    t = time.perf_counter()
    for _ in range(n):
        w.step(i[0]); i[0] += 1
    w.sync()
    return (time.perf_counter() - t) / n * 1e3


def setsw(name, on):
    if name == "prefetch":
        w.prefetch_plan = planmod.prefetch_plan if on else w.build_plan
        return
    mod, attr = name.split(".")
    setattr({"models": models, "ops": ops, "plan": planmod}[mod], attr, on)


for _ in range(2):
    block(20)
for name in sys.argv[1:]:
    res = {True: [], False: []}
    for rep in range(4):
        for on in (True, False):
            setsw(name, on)
            block(10)                      # settle (allocator, caches) after the switch
            res[on].append(block(40))
    setsw(name, True)
    a, b = statistics.median(res[True]), statistics.median(res[False])
    D = w.trainer.discriminator
    print("   [i=%d collapse_ok=%s amplification=%.1f finite=%s]" % (i[0], D.voxelGNN_D.collapse_ok, D.guard_collapse(w.batches[0]),
          bool(torch.isfinite(w.trainer.opt_d.flat).all() and torch.isfinite(w.trainer.opt_g.flat).all())), flush=True)
    print("%-28s on %.3f ms/step  off %.3f ms/step  (%+.1f %%)  blocks on %s off %s" % (
        name, a, b, (b - a) / b * 100, ["%.3f" % x for x in res[True]], ["%.3f" % x for x in res[False]]), flush=True)
