"""A CPU stand-in for bench.HipWorkload (TEST INFRASTRUCTURE): the same methods, gloo ranks, the Trainer's host
logic over the oracle networks (the HIP networks need a GPU).  `bench.py --gpus 2 --workload
tests.bench_cpu_workload:CpuWorkload` rehearses the launcher, the rendezvous, the barriers, the max-over-ranks
timing and the JSON line without hardware (tests/test_bench_launcher.py)."""
import os

import numpy as np
import torch
import torch.distributed as dist


class CpuWorkload:
    def __init__(self, args, rank, world):
        from building_gan_amd.graph import Batch
        from building_gan_amd.synthetic import make_graph
        from building_gan_amd.train import Trainer
        from oracle.networks import DiscriminatorVoxel, Generator
        torch.set_num_threads(1)
        self.world, self.rank = world, rank
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        rng = np.random.default_rng(100 + rank)
        shapes = [(2, 3, 2, [2, 3]), (1, 2, 2, [1])]
        self.batches = [Batch.from_data_list([make_graph(rng, f, x, y, pps, data_id=i)
                                              for i, (f, x, y, pps) in enumerate(shapes)]) for _ in range(args.batches)]
        torch.manual_seed(77)
        G, D = Generator(7, 7, 32, 8, 1, 3), DiscriminatorVoxel(7, 6, 32, 3)
        self.trainer = Trainer(torch.device("cpu"), noise_dim=8, generator=G, discriminator=D)
        from oracle.pointer import get_program_ratio
        self.trainer.program_ratio_fn = get_program_ratio      # the HIP pipeline needs a GPU plan
        self.trainer.broadcast_parameters(0)
        self.graphs_per_step = 2

    def step(self, i):
        from building_gan_amd.train import global_counts
        b = self.batches[i % len(self.batches)]
        tr = self.trainer
        counts = global_counts(b)
        gen, _ = tr._phase_d(b, counts, True, g_update=i % tr.n_critic_g == 0)
        tr.opt_d.all_reduce()
        if i % tr.n_critic_g == 0:
            tr._phase_g(b, counts, gen)
            tr.opt_g.all_reduce()

    def sync(self):
        if self.world > 1:
            dist.barrier()

    def max_over_ranks(self, dt):
        if self.world > 1:
            t = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    def joined(self):
        return dist.get_world_size() if self.world > 1 else 1

    def config(self):
        return {"workload": "CPU rehearsal of the bench harness (oracle networks, gloo)", "parallelism": "dp%d" % self.world}

    def extras(self, line):
        line["grad_checksum"] = float(self.trainer.opt_d.grad.double().abs().sum())

    def close(self):
        if self.world > 1:
            dist.barrier()
            dist.destroy_process_group()
