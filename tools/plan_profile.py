"""Where the time of one plan build goes (run on the GPU box): host issue, device execution, read-back, wrapping."""
import sys
import time

import torch

sys.path.insert(0, ".")
from building_gan_amd import plan as P                      # noqa: E402
from building_gan_amd.synthetic import make_batch            # noqa: E402


def main():
    dev = torch.device("cuda:0")
    preset = sys.argv[1] if len(sys.argv) > 1 else "small"
    batches = [make_batch(preset, seed=i, ragged="jitter").to(dev) for i in range(8)]
    for one_call in (True, False):
        P.ONE_CALL_BUILD = one_call
        for rep in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for g in batches:
                g.__dict__.pop("_bg_plan", None)
                P.build_plan(g)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / len(batches)
        S = [int(g.voxel_floor_cluster.max()) + 1 for g in batches]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g, s in zip(batches, S):
            g.__dict__.pop("_bg_plan", None)
            P.build_plan(g, num_stories=s)
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / len(batches)
        print(f"one_call={one_call}: {dt * 1e3:.3f} ms per plan; with the number of stories known {dt_s * 1e3:.3f} ms")
    # the composite alone: host issue time and device time
    import ctypes as C
    from building_gan_amd import _lib
    lib = _lib.load()
    real = lib.bg_plan_build
    acc = {"issue": 0.0, "n": 0}

    def timed(*a):
        t = time.perf_counter()
        rc = real(*a)
        acc["issue"] += time.perf_counter() - t
        acc["n"] += 1
        return rc
    P.ONE_CALL_BUILD = True
    lib.bg_plan_build = timed
    try:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot_dev = 0.0
        for g in batches:
            g.__dict__.pop("_bg_plan", None)
            torch.cuda.synchronize()
            ev0.record()
            P.build_plan(g)
            ev1.record()
            torch.cuda.synchronize()
            tot_dev += ev0.elapsed_time(ev1)
    finally:
        lib.bg_plan_build = real
    print(f"bg_plan_build host issue {acc['issue'] / acc['n'] * 1e6:.1f} us; events around build_plan {tot_dev / len(batches) * 1e3:.1f} us")


if __name__ == "__main__":
    main()
