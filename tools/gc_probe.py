"""Does Python's cyclic GC show in the eager stream step?  Same process: blocks of 60 steps with the collector on / off."""
import argparse, gc, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

args = argparse.Namespace(gpus=1, steps=60, warmup=10, batches=8, graphs=8, preset="dataset", no_replay=True,
                          no_cpu_baseline=True, no_microbench=True, seed=0)
w = bench.HipWorkload(args, 0, 1)
for i in range(40):
    w.step(i)
w.sync()
i = 40
for rep in range(4):
    for on in (True, False):
        gc.enable() if on else gc.disable()
        gc.collect()
        w.sync()
        t0 = time.perf_counter()
        worst = 0.0
        for _ in range(60):
            t1 = time.perf_counter()
            w.step(i); i += 1
            worst = max(worst, time.perf_counter() - t1)
        w.sync()
        dt = (time.perf_counter() - t0) / 60
        print("gc %-3s %.3f ms/step   slowest host-side step call %.2f ms   gc counts %s" % ("on" if on else "off", dt * 1e3, worst * 1e3, gc.get_count()), flush=True)
gc.enable()
gc.collect()
gc.freeze()          # everything alive now (modules, plans' code objects, the trainer) leaves the collector's generations
for rep in range(3):
    w.sync()
    t0 = time.perf_counter()
    for _ in range(60):
        w.step(i); i += 1
    w.sync()
    print("gc on after gc.freeze(): %.3f ms/step" % ((time.perf_counter() - t0) / 60 * 1e3), flush=True)
