#!/usr/bin/env python3
"""Benchmark of the Building-GAN hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one G+D training iteration (reference train.py:73-114) over one synthetic batch of 8
"dataset-sized" buildings per GPU (BASELINE.json configs[1]); the index i of the iteration keeps
running through warm-up and timed steps, so every 5th step carries the generator update.

``value`` is what a training loop over a stream of batches gets: the timed steps rotate over
``--batches`` (default 8) DISTINCT batches of different shapes, every step rebuilds the batch's index
plan inside the timed region (as for a batch fresh from the loader) and launches its kernels eagerly
-- a captured hipGraph is tied to one batch's shapes.  The hipGraph replay of ONE fixed batch (round
1's headline) is reported beside it as ``replay_best_case``.

With ``--gpus N`` (N > 1) and no WORLD_SIZE in the environment this process only LAUNCHES N ranks
(children of its own, one per GPU, before it makes any GPU call) and exits with their status; under
torchrun (WORLD_SIZE set) it is one of the ranks.

Extra objects on the JSON line rank 0 prints:
  roofline         -- the dominant kernel of the step (fp32 MFMA GEMM of the generator block),
                      timed live with HIP events on the launch stream;
  microbench       -- configs[2]: one message-passing gather-mean layer at 10^6 edges (HBM roofline);
  cpu_baseline     -- the CPU oracle (a "port": the reference itself cannot run) timed
                      on this host on a bounded sample of the same workload (rank 0, N=1 only);
  replay_best_case -- hipGraph replay of one fixed batch, plan build outside the timed region.
"""
import argparse
import importlib
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: dense fp32 matrix peak
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E spec peak
TRAFFIC_FILE = os.path.join("profiles", "r03_traffic.json")


# ------------------------------------------------------------------------------------ launcher
def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launch_ranks(n, argv):
    """Start n ranks of this script as child processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set) and
    wait for them.  The parent has not touched the GPU (it never does: nothing here imports torch) and is
    not replaced by another program; rank 0 prints the JSON line on the inherited stdout.  Returns the exit
    status: 0 only when every rank returned 0; the first failure ends the others."""
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    status = 0
    pending = set(range(n))
    while pending:
        for r in list(pending):
            rc = procs[r].poll()
            if rc is None:
                continue
            pending.discard(r)
            if rc != 0 and status == 0:
                status = rc if rc > 0 else 1
                print("bench.py: rank %d exited with status %d; stopping the other ranks" % (r, rc), file=sys.stderr)
                for q in pending:
                    procs[q].terminate()
        if pending:
            time.sleep(0.05)
    return status


# ------------------------------------------------------------------------------------ helpers
def event_time_ms(fn, iters, warm=3):
    import torch
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
    """HBM bytes per launch from the committed PMC pass (TRAFFIC_FILE; collected with rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
    gfx950), or None when that file does not cover the kernel.  A constant from a committed profile: PMC
    counters cannot be collected from inside this process."""
    try:
        with open(os.path.join(ROOT, TRAFFIC_FILE)) as f:
            return json.load(f).get(key)
    except (OSError, ValueError):
        return None


def roofline_dominant_gemm(dev, batch, plan, hidden=128):
    """The generator block's collapsed Linear as the step launches it: [v | mean_j v_j] (K = 2H) -> H starting from the
    position + bias term (evaluated once per forward), LeakyReLU and the residual, with the per-story term
    P[s] + mean over the building's stories of Q[s'] FORMED WHILE THE PRODUCT LOADS IT (bg_gemm_nbr_f32, round 3: the
    launch that used to write that [S, H] matrix is gone).  ``product_alone`` is the same product reading a finished
    story term (round 2's roofline kernel; what the step still launches where the fused form declines the shape).
    Algorithmic flops = 2 M N K for both."""
    import torch
    from building_gan_amd import ops
    H, Nv = hidden, int(batch.voxel_feature.shape[0])
    torch.manual_seed(1)
    xs = [torch.rand(Nv, H, device=dev) * 2 - 1, torch.rand(Nv, H, device=dev) * 2 - 1]
    W = torch.rand(H, 2 * H, device=dev) * 2 - 1
    posb = torch.rand(Nv, H, device=dev) * 2 - 1
    pq = torch.rand(plan.S, 2 * H, device=dev)
    sb = plan.ops["story_building_mean"]
    term = ops.spmm_add_raw(sb, pq[:, H:], pq[:, :H])
    rows = plan.story_gather.csr.col
    out = torch.empty(Nv, H, device=dev)
    alone = lambda: ops.gemm_raw(xs, W, False, True, act=ops.ACT_LEAKY, slope=0.01, pre_add=term, pre_rows=rows,
                                 residual=xs[0], acc_init=posb, out=out)
    fused = lambda: ops.gemm_nbr_raw(xs, W, pq, sb, rows, act=ops.ACT_LEAKY, slope=0.01, residual=xs[0], acc_init=posb, out=out)
    took = fused() is not None

    def graph_us(run):           # launch-bound in eager mode at this size: the kernel inside a hipGraph of 20 launches
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            run()
        torch.cuda.current_stream().wait_stream(s)
        with torch.cuda.graph(g, capture_error_mode="thread_local"):     # (a process group's watchdog thread may poll events meanwhile)
            for _ in range(20):
                run()
        return event_time_ms(g.replay, 10) / 20 * 1e3
    flops = 2.0 * Nv * (2 * H) * H
    us_alone = graph_us(alone)
    us = graph_us(fused) if took else us_alone
    ach = flops / (us * 1e-6) / 1e12
    key = "gemm_generator_block_fused" if took else "gemm_generator_block"
    return {"kernel": "k_gemm_panel (16x16x4 f32 MFMA, 48-row x 64-column workgroups): generator block Linear, M=%d N=%d "
                      "K=%d from the position + bias term, %s + LeakyReLU + residual epilogue"
                      % (Nv, H, 2 * H, "story term formed on load from the [S, 2H] story pairs and the building's story list"
                         if took else "gathered story term"),
            "bound": "mfma", "achieved": round(ach, 3), "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / MFMA_F32_PEAK_TFLOPS, 4), "traffic": pmc_traffic(key),
            "traffic_source": TRAFFIC_FILE,
            "avg_launch_us": round(us, 2), "algorithmic_flops_per_launch": flops,
            # operands [Nv, 2H] + weight + accumulator start [Nv, H] + result [Nv, H] + story pairs + row index
            # (the residual re-reads the first operand)
            "algorithmic_bytes_per_launch": 4.0 * (Nv * 2 * H + H * 2 * H + 2 * Nv * H + plan.S * 2 * H + Nv),
            "product_alone": {"avg_launch_us": round(us_alone, 2),
                              "achieved": round(flops / (us_alone * 1e-6) / 1e12, 3),
                              "frac": round(flops / (us_alone * 1e-6) / 1e12 / MFMA_F32_PEAK_TFLOPS, 4),
                              "traffic": pmc_traffic("gemm_generator_block"),
                              "what": "the same product reading a finished [S, H] story term (round 2's roofline kernel); the "
                                      "fused launch replaces it AND the launch that wrote the term (~4.8 us in the step)"}}


def microbench_message_passing(dev, pairs=4):
    """configs[2]: gather-mean over a 10^6-edge voxel graph, fp32.  The timed launches rotate over ``pairs``
    operand / result pairs (H=128: 4 x 178 MB) so that no launch finds its operand in the 256 MiB Infinity
    Cache.  roofline.frac is PHYSICAL: HBM bytes per launch (PMC FETCH_SIZE x 2 + WRITE_SIZE from the committed
    pass over this same rotation, else the compulsory bytes operand + result + index read once) / time / peak;
    the logical rate of SURVEY.md 8(d) (every neighbour row counted once per edge) is quoted beside it."""
    import torch
    from building_gan_amd import ops
    from building_gan_amd.plan import build_plan
    from building_gan_amd.synthetic import batch_sizes, make_batch
    batch = make_batch("microbench", seed=0)
    sz = batch_sizes(batch)
    batch.to(dev)
    plan = build_plan(batch)
    res = {"workload": "8 x (20x33x33) voxel graphs", "Nv": sz["Nv"], "Ev": sz["Ev"], "operand_result_pairs": pairs}
    for H in (128, 256):
        xs = [torch.rand(sz["Nv"], H, device=dev) * 2 - 1 for _ in range(pairs)]
        outs = [torch.empty(sz["Nv"], H, device=dev) for _ in range(pairs)]
        k = [0]

        def run():
            ops.spmm_raw(plan.vox_mean, xs[k[0] % pairs], out=outs[k[0] % pairs])
            k[0] += 1
        ms = event_time_ms(run, 48, warm=4)
        logical = sz["Ev"] * (4 * H + 4) + sz["Nv"] * (4 * H + 4 + 4)
        compulsory = 2 * sz["Nv"] * 4 * H + sz["Ev"] * 4 + sz["Nv"] * 8
        traffic = pmc_traffic("spmm_H%d" % H)
        phys = traffic if traffic else compulsory
        gbs = phys / (ms * 1e-3) / 1e9
        res["H%d" % H] = {"edges_per_s": round(sz["Ev"] / (ms * 1e-3), 1), "avg_launch_us": round(ms * 1e3, 2),
                          "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": traffic,
                                       "traffic_source": TRAFFIC_FILE if traffic else None,
                                       "bytes_basis": "pmc" if traffic else "compulsory (operand + result + index once)",
                                       "compulsory_bytes_per_launch": compulsory,
                                       "logical_bytes_per_launch": logical,
                                       "logical_GBs": round(logical / (ms * 1e-3) / 1e9, 1)}}
        del xs, outs
    return res


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_share():
    """Host threads this job may use: the affinity mask, capped by the box's share (OMP_NUM_THREADS is set to
    the share on the GPU box; os.cpu_count() there shows the whole machine)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:
        n = min(n, int(os.environ.get("OMP_NUM_THREADS", n)))
    except ValueError:
        pass
    return max(n, 1)


def cpu_baseline(graphs=8, budget_s=40.0):
    """CPU oracle (pure-PyTorch restatement, literal per-edge formulation) on the SAME workload shape as the GPU
    figure: one batch of ``graphs`` (8) dataset-sized buildings, iterations in the reference's schedule (every 5th
    carries the generator update), SURVEY.md 8(d), on all the host threads of this job.  Bounded sample: one warm-up
    iteration, then three timed ones (two critic-only, one with the generator update);
    a 5-iteration cycle costs t(D+G) + 4 t(D), medians per kind.  The 1-thread figure is taken on ONE graph (a
    batch-8 iteration takes ~40 s on one thread) and says so."""
    import torch
    from building_gan_amd.graph import detach_batch
    from building_gan_amd.synthetic import batch_sizes, make_batch
    from oracle.networks import DiscriminatorVoxel, Generator
    from oracle.objectives import train_iteration

    def run(threads, n_graphs, first, count, budget):
        batch = make_batch("dataset", seed=0, graphs=n_graphs)
        sz = batch_sizes(batch)
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        G, D = Generator(7, 7, 128, 32, 4, 12), DiscriminatorVoxel(7, 6, 128, 12)
        og = torch.optim.Adam(G.parameters(), lr=1e-4, betas=(0.5, 0.999))
        od = torch.optim.Adam(D.parameters(), lr=1e-4, betas=(0.5, 0.999))
        t_g, t_d = [], []
        start = time.perf_counter()
        # iteration indices first .. : the warm-up one is critic-only, the timed ones hold one generator update
        for k, i in enumerate(range(first, first + 1 + count)):
            pz, vz = torch.rand(sz["Np"], 32), torch.rand(sz["Nv"], 32)
            t = time.perf_counter()
            train_iteration(G, D, og, od, batch, detach_batch, i, pz, vz, None, None)
            dt = time.perf_counter() - t
            if k >= 1:
                (t_g if i % 5 == 0 else t_d).append(dt)
            if t_g and t_d and time.perf_counter() - start > budget:
                break
        cycle = statistics.median(t_g) + 4 * statistics.median(t_d)
        return 5 * n_graphs / cycle, statistics.median(t_g), statistics.median(t_d), len(t_g) + len(t_d), sz

    prev = torch.get_num_threads()
    cores = _cpu_share()
    try:
        # warm-up = iteration 3, timed = 4, 5 (generator update), 6  (~5 s each on 16 EPYC threads)
        v_all, tg, td, n_all, sz = run(cores, graphs, 3, 3, budget_s)
        v_one, tg1, td1, n_one, sz1 = run(1, 1, 3, 2, budget_s / 2)
    finally:
        torch.set_num_threads(prev)
    return {"value": round(v_all, 4), "unit": "voxel-graphs/s", "cores": cores, "kind": "port",
            "sample_graphs_per_batch": graphs, "sample_is_batch_1": graphs == 1,
            "one_thread_value": round(v_one, 4), "one_thread_sample_graphs_per_batch": 1,
            "cpu_model": _cpu_model(), "torch": torch.__version__,
            "timed_iterations": n_all, "timed_iterations_one_thread": n_one,
            "sample": "one batch of %d dataset-sized graphs (Nv=%d, Ev=%d, Ec=%d: the GPU figure's batch shape); reference "
                      "schedule, 1 warm-up + %d timed iterations on %d threads: iteration with G update %.3fs, critic-only "
                      "%.3fs; 1 thread on ONE graph (Nv=%d): %.3fs / %.3fs (%d timed); pure-PyTorch CPU oracle, literal "
                      "per-edge formulation (the reference itself cannot run: SURVEY.md 8c)"
                      % (graphs, sz["Nv"], sz["Ev"], sz["Ec"], n_all, cores, tg, td, sz1["Nv"], tg1, td1, n_one)}


def inference_figures(dev):
    """BASELINE.md workload (1), configs[0] / configs[4]: generator inference as /root/reference/inference.py:74-82 ->
    util_eval.evaluate runs it (forward, check_connectivity, get_program_ratio per batch) on ONE dataset-sized
    building (batch 1), and the variation batch of a large building (8 copies of 32 stories x 64 x 64 = 1,048,576
    voxels; reference util_eval.py:139-157).  Random-init weights of the released architecture (the timing does not
    depend on the values; the released checkpoint's parity is tests/test_golden_gpu.py); synthetic graphs."""
    import torch
    from building_gan_amd.graph import Batch
    from building_gan_amd.models import Generator
    from building_gan_amd.plan import build_plan
    from building_gan_amd.synthetic import batch_sizes, make_batch
    from building_gan_amd.util_eval import check_connectivity, generate_multiple_outputs_from_batch
    from building_gan_amd.util_graph import get_program_ratio
    res = {}
    torch.manual_seed(0)
    G = Generator(7, 7, 128, 32, 4, 12, dev).to(dev).eval()
    b = make_batch("dataset", seed=3, graphs=1).to(dev)
    sz = batch_sizes(b)

    def one(fresh_plan):
        if fresh_plan:
            b.__dict__.pop("_bg_plan", None)
        with torch.no_grad():
            pz, vz = torch.rand(sz["Np"], 32, device=dev), torch.rand(sz["Nv"], 32, device=dev)
            out, soft, mask, att, idx = G(b, pz, vz)
            check_connectivity(b, idx, mask["hard"])
            get_program_ratio(b, att["hard"], mask["hard"], 6)

    def wall(fn, n, warm=3):
        for _ in range(warm):
            fn()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / n * 1e3
    ms_fresh, ms_kept = wall(lambda: one(True), 30), wall(lambda: one(False), 30)
    res["batch1"] = {"workload": "1 dataset-sized building (Nv=%d, Ev=%d, Ec=%d): generator forward + check_connectivity + "
                                 "get_program_ratio, as util_eval.evaluate" % (sz["Nv"], sz["Ev"], sz["Ec"]),
                     "ms_per_design": round(ms_fresh, 3), "designs_per_s": round(1e3 / ms_fresh, 1),
                     "ms_per_design_plan_kept": round(ms_kept, 3)}
    del b
    Gl = Generator(7, 7, 128, 32, 4, 12, dev, max_len=32).to(dev).eval()      # 32 stories: a 32-row position table
    variations = 8
    big = Batch.from_data_list([make_batch("large", seed=0)] * variations).to(dev)
    szl = batch_sizes(big)
    torch.cuda.synchronize()
    t = time.perf_counter()
    build_plan(big)
    torch.cuda.synchronize()
    plan_ms = (time.perf_counter() - t) * 1e3
    ms = wall(lambda: generate_multiple_outputs_from_batch(big, variations, Gl, None, None, None, [dev]), 3, warm=2)
    res["variation_batch_1M_voxels"] = {
        "workload": "%d variations of a 32-story 64x64 building in one batch (Nv=%d, Ev=%d, Ec=%d): generator forward + "
                    "program ratios, as util_eval.generate_multiple_outputs_from_batch" % (variations, szl["Nv"], szl["Ev"], szl["Ec"]),
        "ms_per_batch": round(ms, 2), "voxels_per_s": round(szl["Nv"] / (ms * 1e-3), 0), "designs_per_s": round(variations / (ms * 1e-3), 1),
        "plan_build_ms": round(plan_ms, 1), "peak_mem_GiB": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2)}
    del big
    torch.cuda.empty_cache()
    return res


# ------------------------------------------------------------------------------------ the workload
class HipWorkload:
    """BASELINE.json configs[1] on one rank: a stream of distinct batches through train.Trainer."""

    def __init__(self, args, rank, world):
        import torch
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if os.environ.get("BG_DIST_BACKEND", "nccl") == "nccl" and world > torch.cuda.device_count():
            raise SystemExit("bench.py: %d ranks asked for, %d GPUs visible" % (world, torch.cuda.device_count()))
        local = local % max(torch.cuda.device_count(), 1)      # gloo rehearsal: several ranks may share one card
        torch.cuda.set_device(local)
        self.dev = dev = torch.device("cuda", local)
        self.world, self.rank, self.args = world, rank, args
        self.force_dist = bool(getattr(args, "force_dist", False)) and world == 1
        if self.force_dist:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            from building_gan_amd import train as _train
            _train.FORCE_COLLECTIVES = True
        if world > 1:
            backend = os.environ.get("BG_DIST_BACKEND", "nccl")  # "gloo" only to rehearse the N > 1 path on one GPU
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        from building_gan_amd.plan import build_plan, prefetch_plan
        from building_gan_amd.synthetic import batch_sizes, make_batch
        from building_gan_amd.train import Trainer
        self.build_plan = build_plan
        self.prefetch_plan = prefetch_plan if not os.environ.get("BG_BENCH_NO_PREFETCH") else build_plan
        # the stream: distinct batches of different shapes (stories, grid, program nodes vary per graph around
        # the dataset preset), resident in HBM before the timed region starts
        self.batches = [make_batch(args.preset, seed=1000 * rank + k, ragged="jitter") for k in range(args.batches)]
        self.sizes = [batch_sizes(b) for b in self.batches]
        for b in self.batches:
            b.to(dev)
        self.fixed = make_batch(args.preset, seed=rank)          # the fixed batch of the replay figure / roofline
        self.fixed_sz = batch_sizes(self.fixed)
        self.fixed.to(dev)
        torch.manual_seed(0)
        self.trainer = Trainer(dev)
        self.trainer.grad_exchange = getattr(args, "grad_exchange", "all_reduce")
        self.trainer.broadcast_parameters(0)
        torch.manual_seed(1000 + rank)
        build_plan(self.fixed)                    # first build also loads the library and warms the allocator
        self.graphs_per_step = self.sizes[0]["B"]

    def step(self, i):
        """One iteration on batch i of the stream plus ONE plan build: the plan of the NEXT batch, rebuilt from its
        index vectors as for a batch fresh from the loader.  Its launches go to a side stream BEFORE this iteration is
        issued (plan.prefetch_plan -- the loader-side prefetch of a training loop; train.Trainer.fit does the same), so
        its flags have reached the host long before the next step reads them.  Only the very first step finds no plan
        waiting and builds its own first."""
        n = len(self.batches)
        b = self.batches[i % n]
        if "_bg_plan" not in b.__dict__:
            self.build_plan(b)                    # finishes the plan the previous step started (or builds the first one)
        nxt = self.batches[(i + 1) % n]
        nxt.__dict__.pop("_bg_plan", None)        # a fresh batch: its index plan is part of the step
        nxt.__dict__.pop("_bg_plan_pending", None)
        getattr(nxt.voxel_feature, "__dict__", {}).pop("_bg_split", None)      # (and nothing else derived from it is kept)
        if self.prefetch_plan is not self.build_plan:
            self.trainer.prefetch(nxt)            # plan (+ with several ranks the global counts) on a side stream
        self.trainer.train_iteration(b, i)
        if self.prefetch_plan is self.build_plan:
            self.build_plan(nxt)

    def sync(self):
        import torch
        import torch.distributed as dist
        torch.cuda.synchronize()
        if self.world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(self, dt):
        import torch
        import torch.distributed as dist
        if self.world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=self.dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def joined(self):
        import torch.distributed as dist
        return dist.get_world_size() if self.world > 1 else 1

    def plan_build_ms(self):
        import torch
        b = self.batches[0]
        ts = []
        for _ in range(5):
            b.__dict__.pop("_bg_plan", None)
            torch.cuda.synchronize()
            t = time.perf_counter()
            self.build_plan(b)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t) * 1e3)
        return round(statistics.median(ts), 3)

    def settle_host(self):
        self.trainer.settle_host()

    def replay_best_case(self, steps, warmup):
        """hipGraph replay of the captured iteration on ONE fixed batch, plan build outside (all ranks)."""
        tr = self.trainer
        tr.capture(self.fixed)
        i = 0
        for _ in range(warmup):
            tr.train_iteration_graphed(i)
            i += 1
        self.sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            tr.train_iteration_graphed(i)
            i += 1
        self.sync()
        dt = self.max_over_ranks(time.perf_counter() - t0)
        return {"value": round(self.fixed_sz["B"] * self.world * steps / dt, 3), "unit": "voxel-graphs/s",
                "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps,
                "what": "hipGraph replay of one fixed batch (Nv=%d), plan build outside the timed region"
                        % self.fixed_sz["Nv"]}

    def config(self):
        a = self.args
        n = len(self.sizes)
        mean = lambda k: round(sum(s[k] for s in self.sizes) / n, 1)
        return {"workload": "G+D WGAN-GP train step over a stream of %d distinct synthetic 6types-shaped batches per "
                            "GPU, %d graphs each (stories 8-11, grid 10-14 x 10-14 per graph, 2-6 program nodes per "
                            "story), one plan build per step (the next batch's, prefetched on a side stream) + eager launches inside the timed "
                            "region" % (n, self.sizes[0]["B"]),
                "preset": a.preset, "graphs_per_gpu": self.sizes[0]["B"], "distinct_batches": n,
                "Nv_mean": mean("Nv"), "Ev_mean": mean("Ev"), "Ec_mean": mean("Ec"), "Np_mean": mean("Np"),
                "Nv_min": min(s["Nv"] for s in self.sizes), "Nv_max": max(s["Nv"] for s in self.sizes),
                "hidden": 128, "voxel_layer": 12, "program_layer": 4, "n_critic_g": 5,
                "parallelism": "dp%d" % self.world,
                "launch": "eager kernels + eager RCCL " + ("all-gather + local sum in rank order (--grad-exchange gather_sum)"
                                                           if self.trainer.grad_exchange == "gather_sum" else "all-reduce")
                          + " of the flat gradient buffers",
                "collectives": ("issued on a 1-rank RCCL group (--force-dist)" if self.force_dist else
                                ("RCCL, %d ranks" % self.world if self.world > 1 else "none (one rank)"))}

    def extras(self, line):
        a = self.args
        plan = self.build_plan(self.fixed)
        line["roofline"] = roofline_dominant_gemm(self.dev, self.fixed, plan)
        line["plan_build_ms"] = self.plan_build_ms()
        if self.world == 1 and not a.no_microbench:
            line["microbench"] = microbench_message_passing(self.dev)
        if self.world == 1 and not a.no_inference:
            line["inference"] = inference_figures(self.dev)
        if self.world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()

    def close(self):
        import torch.distributed as dist
        if self.world > 1 or self.force_dist:
            dist.barrier()
            dist.destroy_process_group()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)      # 20 five-iteration cycles: a 20-step window moves by +-5 % from run to run
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--preset", default="dataset")
    ap.add_argument("--batches", type=int, default=8, help="distinct batches the timed steps rotate over")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-microbench", action="store_true")
    ap.add_argument("--no-inference", action="store_true", help="skip the generator-inference figures (configs[0] / configs[4])")
    ap.add_argument("--force-dist", action="store_true",
                    help="with --gpus 1: create a 1-rank process group (RCCL) and issue every collective of the "
                         "data-parallel path anyway (train.FORCE_COLLECTIVES); the line says so")
    ap.add_argument("--grad-exchange", choices=("all_reduce", "gather_sum"), default="all_reduce",
                    help="how the ranks' flat gradient buffers are summed: RCCL's all-reduce (default), or one all-gather and "
                         "a local sum in rank order (FlatAdam.all_reduce: the one-shot form for these ~1 MB messages)")
    ap.add_argument("--no-replay", action="store_true", help="skip the fixed-batch hipGraph replay figure")
    ap.add_argument("--workload", default=None, metavar="MODULE:CLASS",
                    help="replace the HIP workload by MODULE:CLASS(args, rank, world) with the same methods "
                         "(tests/ rehearse the launcher and the timing harness on CPU ranks with it)")
    argv = list(sys.argv[1:] if argv is None else argv)
    args = ap.parse_args(argv)

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, argv)              # before anything in this process touches a GPU

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.workload:
        mod, cls = args.workload.split(":")
        wl = getattr(importlib.import_module(mod), cls)(args, rank, world)
    else:
        wl = HipWorkload(args, rank, world)
    joined = wl.joined()
    if joined != args.gpus:
        raise SystemExit("bench.py: %d ranks joined, --gpus %d" % (joined, args.gpus))

    i = 0
    for _ in range(args.warmup):
        wl.step(i)
        i += 1
    if hasattr(wl, "settle_host"):
        wl.settle_host()              # what Trainer.fit does before its loop: collect once, gc.freeze() (collector stays on)
    wl.sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(i)
        i += 1
    wl.sync()
    dt = wl.max_over_ranks(time.perf_counter() - t0)

    replay = None
    if not args.no_replay and hasattr(wl, "replay_best_case"):
        replay = wl.replay_best_case(args.steps, args.warmup)

    if rank == 0:
        graphs = wl.graphs_per_step * joined * args.steps
        line = {
            "metric": "voxel-graphs/sec (G+D train step)", "value": round(graphs / dt, 3), "unit": "voxel-graphs/s",
            "n_gpus": joined, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": wl.config(),
        }
        if replay is not None:
            line["replay_best_case"] = replay
        wl.extras(line)
        print(json.dumps(line), flush=True)
    wl.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
