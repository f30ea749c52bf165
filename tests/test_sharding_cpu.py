"""CPU, world_size 2 over gloo: the multi-GPU path of bench.py (round-robin chunk sharding, barrier-bracketed timing,
MAX-over-ranks reduction).  Chunks are independent, so there is no data-path collective to test."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_chunks_round_robin():
    sys.path.insert(0, ROOT)
    import bench
    parts = [bench.shard_chunks(64, 8, r) for r in range(8)]
    assert parts[0][:3] == [0, 8, 16] and parts[7][-1] == 63
    assert sorted(sum(parts, [])) == list(range(64))           # every chunk exactly once
    assert all(len(p) == 8 for p in parts)                      # weak scaling: fixed work per GPU
    assert bench.shard_chunks(5, 2, 1) == [1, 3]                # ragged tail


def test_two_rank_dry_run_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run", "--batch", "4"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout           # only rank 0 prints
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["chunks_rank0"] == [0, 2, 4, 6]
    assert 0 < j["value"] < 2 * 4 * 30 / 0.04 * 1.01   # 8 chunks per 40 ms step at most


def test_eight_rank_dry_run_gloo_self_launched():
    """`python bench.py --gpus 8` launched bare becomes its own torchrun launcher (the line the driver would use on an 8-GPU node): eight gloo
    ranks on this CPU box, 64 chunks per step sharded round-robin, one JSON line from rank 0, every rank pinned to its own slice of the cores
    when there are at least eight."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["steps"] == 2 and j["chunks_rank0"] == [0, 8, 16, 24, 32, 40, 48, 56]
    assert 0 < j["value"] < 64 * 30 / 0.08 * 1.01      # 64 chunks per 80 ms step at most


def test_two_ranks_transcribe_disjoint_shards_gloo(toy_ml_path, tmp_path):
    """The data path across ranks with a real transcriber in place of the GPU engine (the CPU oracle; tests/shard_worker.py): two ranks, 7 chunks
    (ragged), every chunk transcribed exactly once, results gathered in chunk order and identical to a single-process run."""
    from oracle import binding as orc
    from speaksense_amd import synth
    out = str(tmp_path / "gathered.json")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "tests", "shard_worker.py"), toy_ml_path, "7", out]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541"))
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.load(open(out))
    assert j["shards"] == [[0, 2, 4, 6], [1, 3, 5]] and sorted(int(k) for k in j["tokens"]) == list(range(7))
    om = orc.OracleModel(toy_ml_path)
    P = orc.default_params(language="en", temperature_inc=0.0)
    for cid in range(7):
        ref = om.new_state(orc.MODE_GGML_F16).full(synth.speech_like(100 + cid, 16000 * 3), P)
        assert j["tokens"][str(cid)] == [int(t) for t in ref["tokens"]], cid
    assert j["seconds_max_over_ranks"] > 0
    om.close()
