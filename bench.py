#!/usr/bin/env python3
"""Benchmark of the Building-GAN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one G+D training iteration (reference train.py:73-114) over one synthetic batch of 8
"dataset-sized" buildings per GPU (BASELINE.json configs[1]); the index i of the iteration keeps
running through warm-up and timed steps, so every 5th step carries the generator update.
value = voxel-graphs/s over the whole job.  Extra objects on the same JSON line:
  roofline     -- the dominant kernel of the step (fp32 MFMA GEMM of the generator block),
                  timed live with HIP events on the launch stream;
  microbench   -- configs[2]: one message-passing gather-mean layer at 10^6 edges (HBM roofline);
  cpu_baseline -- the CPU oracle (a "port": the reference itself cannot run) timed
                  on this host on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak


def event_time_ms(fn, iters, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def pmc_traffic(key):
    """HBM bytes per launch from the committed PMC pass (profiles/r01_traffic.json; collected with
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950), or None when that file does not cover the kernel."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def roofline_dominant_gemm(dev, batch, plan, hidden=128):
    """The kernel with the largest share of the step (profiles/r01_bench_kernel_stats*): the generator
    block's collapsed Linear as the step launches it, [v | mean_j v_j] (K = 2H) -> H starting from the
    position + bias term (evaluated once per forward), the per-story term gathered in the epilogue,
    LeakyReLU and the residual.  Algorithmic flops = 2 M N K."""
    from building_gan_amd import ops
    H, Nv = hidden, int(batch.voxel_feature.shape[0])
    torch.manual_seed(1)
    xs = [torch.rand(Nv, H, device=dev) * 2 - 1, torch.rand(Nv, H, device=dev) * 2 - 1]
    W = torch.rand(H, 2 * H, device=dev) * 2 - 1
    posb = torch.rand(Nv, H, device=dev) * 2 - 1
    term = torch.rand(plan.S, H, device=dev)
    rows = plan.story_gather.csr.col
    run = lambda: ops.gemm_raw(xs, W, False, True, act=ops.ACT_LEAKY, slope=0.01, pre_add=term, pre_rows=rows,
                               residual=xs[0], acc_init=posb)
    # launch-bound in eager mode at this size: time the kernel inside a hipGraph of 20 launches
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        run()
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        for _ in range(20):
            run()
    ms = event_time_ms(g.replay, 10) / 20
    flops = 2.0 * Nv * (2 * H) * H
    ach = flops / (ms * 1e-3) / 1e12
    return {"kernel": "k_gemm<64,64,64> (generator block Linear, M=%d N=%d K=%d from the position + bias term, "
                      "fused gather + LeakyReLU + residual epilogue)" % (Nv, H, 2 * H),
            "bound": "mfma", "achieved": round(ach, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": pmc_traffic("gemm_generator_block"),
            "avg_launch_us": round(ms * 1e3, 2), "algorithmic_flops_per_launch": flops,
            # operands [Nv, 2H] + weight + accumulator start [Nv, H] + result [Nv, H] + story term + row index
            # (the residual re-reads the first operand)
            "algorithmic_bytes_per_launch": 4.0 * (Nv * 2 * H + H * 2 * H + 2 * Nv * H + plan.S * H + Nv)}


def microbench_message_passing(dev):
    """configs[2]: gather-mean over a 10^6-edge voxel graph, H = 128 fp32 (SURVEY.md 8d bytes formula)."""
    from building_gan_amd import ops
    from building_gan_amd.plan import build_plan
    from building_gan_amd.synthetic import batch_sizes, make_batch
    batch = make_batch("microbench", seed=0)
    sz = batch_sizes(batch)
    batch.to(dev)
    plan = build_plan(batch)
    res = {"workload": "8 x (20x33x33) voxel graphs", "Nv": sz["Nv"], "Ev": sz["Ev"]}
    for H in (128, 256):
        x = torch.rand(sz["Nv"], H, device=dev) * 2 - 1
        ms = event_time_ms(lambda: ops.spmm_raw(plan.vox_mean, x), 50)
        nbytes = sz["Ev"] * (4 * H + 4) + sz["Nv"] * (4 * H + 4 + 4)
        gbs = nbytes / (ms * 1e-3) / 1e9
        res["H%d" % H] = {"edges_per_s": round(sz["Ev"] / (ms * 1e-3), 1), "avg_launch_us": round(ms * 1e3, 2),
                          "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": pmc_traffic("spmm_H%d" % H),
                                       "algorithmic_bytes_per_launch": nbytes}}
    return res


def cpu_baseline(graphs=2):
    """CPU oracle on `graphs` dataset-sized buildings: iteration 0 (D+G update) and iteration 1
    (D update only); a 5-iteration cycle costs t0 + 4*t1."""
    from building_gan_amd.graph import detach_batch
    from building_gan_amd.synthetic import batch_sizes, make_batch
    from oracle.networks import DiscriminatorVoxel, Generator
    from oracle.objectives import train_iteration
    cores = min(16, os.cpu_count() or 1)
    try:
        cores = min(cores, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    torch.set_num_threads(cores)
    batch = make_batch("dataset", seed=0, graphs=graphs)
    sz = batch_sizes(batch)
    torch.manual_seed(0)
    G, D = Generator(7, 7, 128, 32, 4, 12), DiscriminatorVoxel(7, 6, 128, 12)
    og = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999))
    od = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.999))
    times = []
    for i in (5, 0, 1):      # first call = untimed warm-up (allocator, thread pool)
        pz, vz = torch.rand(sz["Np"], 32), torch.rand(sz["Nv"], 32)
        t = time.perf_counter()
        train_iteration(G, D, og, od, batch, detach_batch, i, pz, vz, None, None)
        times.append(time.perf_counter() - t)
    cycle = times[1] + 4 * times[2]
    return {"value": round(5 * graphs / cycle, 4), "unit": "voxel-graphs/s", "cores": cores, "kind": "port",
            "sample": "%d dataset-sized graphs (Nv=%d, Ev=%d); iteration 0 (D+G) %.2fs + 4 x iteration 1 (D) %.2fs; "
                      "pure-PyTorch CPU oracle, literal per-edge formulation, torch %s"
                      % (graphs, sz["Nv"], sz["Ev"], times[1], times[2], torch.__version__)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--preset", default="dataset")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-microbench", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of hipGraph replay")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    local = local % max(torch.cuda.device_count(), 1)      # rehearsal: several ranks may share one card
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BG_DIST_BACKEND", "nccl")  # "gloo" only to rehearse the N > 1 path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from building_gan_amd.plan import build_plan
    from building_gan_amd.synthetic import batch_sizes, make_batch
    from building_gan_amd.train import Trainer

    batch = make_batch(args.preset, seed=rank)
    sz = batch_sizes(batch)
    batch.to(dev)
    torch.manual_seed(0)
    trainer = Trainer(dev)
    trainer.broadcast_parameters(0)
    torch.manual_seed(1000 + rank)
    build_plan(batch)                     # first build also loads the library and warms the allocator
    batch.__dict__.pop("_bg_plan", None)
    torch.cuda.synchronize()
    t_plan = time.perf_counter()
    build_plan(batch)                     # per-batch index preprocessing (CSR by target/source, segment tables)
    torch.cuda.synchronize()
    plan_ms = round((time.perf_counter() - t_plan) * 1e3, 3)

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.no_graph:
        step_fn = lambda it: trainer.train_iteration(batch, it)
    else:
        trainer.capture(batch)
        step_fn = trainer.train_iteration_graphed
    i = 0
    for _ in range(args.warmup):
        step_fn(i)
        i += 1
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step_fn(i)
        i += 1
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        graphs = sz["B"] * world * args.steps
        line = {
            "metric": "voxel-graphs/sec (G+D train step)", "value": round(graphs / dt, 3), "unit": "voxel-graphs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "G+D WGAN-GP train step, %d synthetic 6types-shaped graphs per GPU "
                                   "(10 stories x 12x12 voxels, 4 program nodes/story)" % sz["B"],
                       "preset": args.preset, "graphs_per_gpu": sz["B"], "Nv": sz["Nv"], "Ev": sz["Ev"], "Ec": sz["Ec"],
                       "Np": sz["Np"], "hidden": 128, "voxel_layer": 12, "program_layer": 4,
                       "n_critic_g": 5, "parallelism": "dp%d" % world,
                       "launch": "eager" if args.no_graph else "hipGraph segments + eager RCCL all-reduce"},
        }
        line["roofline"] = roofline_dominant_gemm(dev, batch, build_plan(batch))
        line["plan_build_ms"] = plan_ms
        if world == 1 and not args.no_microbench:
            line["microbench"] = microbench_message_passing(dev)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
