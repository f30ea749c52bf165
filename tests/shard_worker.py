"""One rank of the world_size-2 gloo test (tests/test_sharding_cpu.py): the multi-GPU data path with the GPU engine replaced by the CPU oracle.
Each rank takes its round-robin shard of the chunk list (bench.shard_chunks, the rule bench.py and a node-level service use), transcribes it,
and the per-chunk results are gathered on rank 0 in chunk order -- exactly what the hot path does across GPUs: no collective touches the data,
only the result gather and the timing reduction use torch.distributed."""
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import binding as orc  # noqa: E402
from speaksense_amd import synth  # noqa: E402


def main():
    model_path, n_chunks, out_path = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = bench.shard_chunks(n_chunks, world, rank)
    orc.set_thread_cap(2)
    om = orc.OracleModel(model_path)
    P = orc.default_params(language="en", temperature_inc=0.0)

    def step():
        out = {}
        for cid in mine:
            r = om.new_state(orc.MODE_GGML_F16).full(synth.speech_like(100 + cid, 16000 * 3), P)
            out[cid] = [int(t) for t in r["tokens"]]
        return out

    res = {}
    dt, _ = bench.timed_steps(lambda: res.update(step()), 1, 0, dist, lambda: None)
    gathered = [None] * world
    dist.all_gather_object(gathered, res)
    if rank == 0:
        merged = {}
        for g in gathered:
            assert not (set(g) & set(merged)), "a chunk was transcribed by two ranks"
            merged.update(g)
        json.dump({"tokens": {str(k): v for k, v in sorted(merged.items())}, "seconds_max_over_ranks": dt, "shards": [bench.shard_chunks(n_chunks, world, r) for r in range(world)]},
                  open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
