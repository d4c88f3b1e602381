"""Inference driver restating /root/reference/inference.py:66-82: load a generator checkpoint by key
name (strict=False, as the reference does for runs/iccv2021/checkpoints/70.ckpt), generate designs
for a list of raw-JSON data ids, write the result JSONs, and sample ``variation_num`` variations.

    python -m building_gan_amd.inference --ckpt 70.ckpt --raw_dir Data/6types-raw_data --output_dir out --ids 0 1 2
"""
from __future__ import annotations

import argparse
import functools
import os
from typing import List, Sequence

import torch

from .graph import Batch, DEFAULT_FOLLOW_BATCH
from .json_io import load_graph_jsons, save_output
from .models import Generator
from .util_eval import evaluate, generate_multiple_outputs_from_batch


def load_generator(ckpt_path: str, device, hidden_dim=128, noise_dim=32, program_layer=4, voxel_layer=12,
                   max_len: int = 20) -> Generator:
    gen = Generator(7, 7, hidden_dim, noise_dim, program_layer, voxel_layer, device, max_len).to(device)
    state = torch.load(ckpt_path, map_location="cpu", weights_only=True)   # never unpickle arbitrary objects
    gen.load_state_dict(state, strict=False)                               # extra checkpoint keys are dropped
    return gen.eval()


def run(ckpt_path: str, raw_dir: str, output_dir: str, ids: Sequence[int], device=None, batch_size: int = 20,
        variation_num: int = 25, variation_ids: Sequence[int] = ()) -> float:
    device = device or torch.device("cuda", 0)
    gen = load_generator(ckpt_path, device)
    graphs = [load_graph_jsons(i, raw_dir) for i in ids]
    loader: List[Batch] = [Batch.from_data_list(graphs[k:k + batch_size], DEFAULT_FOLLOW_BATCH)
                           for k in range(0, len(graphs), batch_size)]
    os.makedirs(output_dir, exist_ok=True)

    def writer(out_dir, batch, cw, pw, far, idx, out, new_data_id_strs=None):
        n = len(batch.data_id_str)
        return save_output(n, batch, cw, pw, far, idx, out, DEFAULT_FOLLOW_BATCH, raw_dir, out_dir, new_data_id_strs)

    acc, _ = evaluate(loader, gen, raw_dir, output_dir, DEFAULT_FOLLOW_BATCH, [device],
                      number_of_batches=len(loader), writer=functools.partial(writer, os.path.join(output_dir, "output")))
    for k, vid in enumerate(variation_ids):
        data = load_graph_jsons(vid, raw_dir)
        batch = Batch.from_data_list([data for _ in range(variation_num)], DEFAULT_FOLLOW_BATCH)
        generate_multiple_outputs_from_batch(batch, variation_num, gen, raw_dir, output_dir, DEFAULT_FOLLOW_BATCH,
                                             [device], writer=functools.partial(writer, os.path.join(output_dir, "var_output%d" % (k + 1))))
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ckpt", required=True)
    ap.add_argument("--raw_dir", required=True)
    ap.add_argument("--output_dir", required=True)
    ap.add_argument("--ids", type=int, nargs="+", required=True)
    ap.add_argument("--variation_ids", type=int, nargs="*", default=[])
    ap.add_argument("--variation_num", type=int, default=25)
    ap.add_argument("--cuda", type=int, default=0)
    a = ap.parse_args()
    run(a.ckpt, a.raw_dir, a.output_dir, a.ids, torch.device("cuda", a.cuda), variation_num=a.variation_num,
        variation_ids=a.variation_ids)


if __name__ == "__main__":
    main()
