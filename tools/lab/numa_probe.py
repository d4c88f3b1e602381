"""Is the run-to-run spread of the stream step (about 1.82 or about 1.86 ms on one box) a matter of where the host thread
runs?  Prints the process's CPU set, the GPU's NUMA node and the nodes' CPU lists, then times the step with the process
bound to (a) its default set, (b) the CPUs of the GPU's node, (c) the CPUs of another node -- each before the GPU is
touched, in a child process.
    python tools/lab/numa_probe.py"""
import glob, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import argparse, os, sys, time
sys.path.insert(0, %r)
cpus = os.environ.get("PROBE_CPUS")
if cpus:
    os.sched_setaffinity(0, {int(c) for c in cpus.split(",")})
import torch
import bench
args = argparse.Namespace(gpus=1, steps=20, warmup=5, batches=8, graphs=8, preset="dataset", no_replay=True,
                          no_cpu_baseline=True, no_microbench=True, seed=0, force_dist=False)
w = bench.HipWorkload(args, 0, 1)
for i in range(40):
    w.step(i)
res = []
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(40, 340):
        w.step(i)
    torch.cuda.synchronize(); res.append((time.perf_counter() - t0) / 300 * 1e3)
print(os.environ.get("PROBE_TAG"), "cpus", len(os.sched_getaffinity(0)), "ms/step", " ".join("%%.3f" %% r for r in res), flush=True)
''' % ROOT


def parse(lst):
    out = []
    for part in lst.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


allowed = sorted(os.sched_getaffinity(0))
print("allowed cpus: %d (%s..%s)" % (len(allowed), allowed[0], allowed[-1]), flush=True)
nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    try:
        nodes[int(d.rsplit("node", 1)[1])] = parse(open(d + "/cpulist").read())
    except OSError:
        pass
print("numa nodes:", {k: "%d cpus" % len(v) for k, v in nodes.items()}, flush=True)
gpu_nodes = []
for d in sorted(glob.glob("/sys/class/drm/card*/device/numa_node")):
    try:
        gpu_nodes.append((d.split("/")[4], int(open(d).read())))
    except (OSError, ValueError):
        pass
print("gpu numa nodes:", gpu_nodes, flush=True)
vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
print("visible devices:", vis, flush=True)


def run(tag, cpus):
    env = dict(os.environ, PROBE_TAG=tag)
    if cpus:
        env["PROBE_CPUS"] = ",".join(str(c) for c in cpus)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=170)
    print((r.stdout.strip().splitlines() or ["(no output) " + r.stderr[-300:]])[-1], flush=True)


for rep in range(2):
    run("default", None)
    for n, cpus in nodes.items():
        use = [c for c in cpus if c in allowed][:16]
        if len(use) >= 4:
            run("node%d" % n, use)
