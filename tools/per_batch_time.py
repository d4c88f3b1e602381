"""GPU time of a critic-only and a generator-update iteration on each batch of the bench's stream (eager, plan kept)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
args = argparse.Namespace(gpus=1, steps=20, warmup=5, batches=8, graphs=8, preset="dataset", no_replay=True,
                          no_cpu_baseline=True, no_microbench=True, seed=0)
w = bench.HipWorkload(args, 0, 1)
for i in range(16):
    w.trainer.train_iteration(w.batches[i % 8], 1)
torch.cuda.synchronize()
for bi, b in enumerate(w.batches):
    out = []
    for it in (1, 5):
        for _ in range(3):
            w.trainer.train_iteration(b, it)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            w.trainer.train_iteration(b, it)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / 10 * 1e3)
    print("batch %d: Nv %6d Np %4d S %3d  critic-only %.3f ms  with generator update %.3f ms" % (
        bi, b.voxel_feature.shape[0], b.program_class_feature.shape[0] if hasattr(b, "program_class_feature") else -1,
        b._bg_plan.S if hasattr(b, "_bg_plan") else -1, out[0], out[1]), flush=True)
