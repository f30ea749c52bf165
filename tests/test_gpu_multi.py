"""GPU: the multi-GPU halves of BASELINE.json's configs, as far as ONE GPU can carry them (no 8-GPU node is available to this build; SURVEY.md
section 8e: replicas + round-robin, no collective).

* `bench.py --gpus 2` with the REAL engine in every rank (two ranks sharing GPU 0 over gloo): one JSON line, 16 chunks per step;
* `ss_pool_*` with four engines ([0, 0, 0, 0] stands for four GPUs) carrying 64 concurrent streams -- configs[3]'s concurrency;
* configs[3] at its own size on one GPU: 64 concurrent gRPC-style streams on the real large-v3 shape, every response equal to what the
  stream gets alone (serial engine)."""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from speaksense_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_two_ranks_with_the_hip_engine():
    """The driver's N > 1 launch line with the real engine behind it: `python bench.py --gpus 2` becomes its own torchrun launcher (one rank per
    GPU; here SS_BENCH_DEVICE pins both ranks to GPU 0 and gloo carries the barrier / MAX reduction), chunks are sharded round-robin, rank 0
    prints one line whose value counts the chunks of BOTH ranks."""
    env = dict(os.environ, SS_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--model", "base.en", "--steps", "3", "--warmup", "1", "--inflight", "2", "--lanes", "2",
           "--dist-backend", "gloo", "--no-cpu-baseline", "--no-steady"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["chunks_per_step"] == 16 and j["scaling"] == "weak" and j["steps"] == 3
    assert j["value"] > 0 and abs(j["value"] - 16 * 3 * 30.0 / (j["ms_per_step"] * 3e-3)) < 1e-2 * j["value"]     # whole-job aggregate over both ranks
    assert j["config"]["parallelism"].startswith("dp2") and "cpu_baseline" not in j


def test_bench_eight_ranks_one_gpu_64_chunks_per_step():
    """BASELINE configs[3] / [4]'s node shape made boring before the hardware exists: the driver's `--gpus 8` launch line with the real
    engine in all EIGHT ranks (pinned to GPU 0, gloo for the barrier / MAX reduction): 64 chunks per step, one JSON line, whole-job aggregate, every
    rank on its own slice of the host's cores, host CPU seconds per chunk reported (what an 8 x 3-lane node needs from its CPUs)."""
    env = dict(os.environ, SS_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--model", "base.en", "--steps", "3", "--warmup", "1", "--inflight", "2", "--lanes", "2",
           "--device-batch", "16", "--dist-backend", "gloo", "--no-cpu-baseline", "--no-steady", "--headline-only"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == 8 and j["config"]["chunks_per_step"] == 64 and j["scaling"] == "weak" and j["steps"] == 3
    assert j["value"] > 0 and abs(j["value"] - 64 * 3 * 30.0 / (j["ms_per_step"] * 3e-3)) < 1e-2 * j["value"]
    assert j["config"]["parallelism"].startswith("dp8")
    hc = j["host_cost"]
    assert 0.0 < hc["cpu_s_per_chunk_mean_over_ranks"] <= hc["cpu_s_per_chunk_max_over_ranks"] < 5.0
    if hc["cores_visible"] and hc["cores_visible"] >= 8:
        assert hc["cores_this_rank"] == hc["cores_visible"] // 8 and hc["affinity_rank0"].startswith("cores ")
    from conftest import report
    report(f"bench.py --gpus 8 on ONE GPU (base.en, gloo): {j['value']:.0f} xRT aggregate, host CPU {hc['cpu_s_per_chunk_mean_over_ranks'] * 1e3:.1f} ms per chunk "
           f"(max over ranks {hc['cpu_s_per_chunk_max_over_ranks'] * 1e3:.1f} ms), {hc['cores_this_rank']} of {hc['cores_visible']} cores per rank")


@pytest.mark.parametrize("n_eng", [4, 8])
def test_pool_of_engines_64_concurrent_streams(toy_ml_path, n_eng):
    """ss_pool_* as the in-library form of "chunks sharded round-robin across the GPUs" at configs[3]'s concurrency: 64 sessions submit at once
    to a pool of four / eight engines ([0] * 8 stands for the eight GPUs of the node); every engine gets its share, every result equals the
    single-engine result."""
    from speaksense_amd import binding
    single = binding.Engine(toy_ml_path, max_batch=8)
    pool = binding.Pool(toy_ml_path, [0] * n_eng, max_batch=8)
    assert pool.n_engines == n_eng
    P = binding.default_params(language="en", temperature_inc=0.0)
    pcms = [synth.speech_like(300 + k % 16, 16000 * 5) for k in range(64)]
    want = {}
    for k in range(16):
        want[k] = single.new_session().transcribe(pcms[k], P)
    ses = [pool.new_session() for _ in pcms]
    out = [None] * 64

    def client(lo, hi):          # 8 client threads, as 8 gRPC streams' tokio tasks would
        tk = [(i, ses[i].submit(pcms[i], P)) for i in range(lo, hi)]
        for i, t in tk:
            out[i] = ses[i].wait(t)
    th = [threading.Thread(target=client, args=(8 * c, 8 * c + 8)) for c in range(8)]
    [t.start() for t in th]
    [t.join() for t in th]
    for i in range(64):
        assert list(out[i]["tokens"]) == list(want[i % 16]["tokens"]) and len(out[i]["tokens"]) > 0, i
        assert [s["text"] for s in out[i]["segments"]] == [s["text"] for s in want[i % 16]["segments"]]
    used = [s.last_engine() for s in ses]
    assert sorted(set(used)) == list(range(n_eng))
    assert max(used.count(e) for e in range(n_eng)) - min(used.count(e) for e in range(n_eng)) <= 8     # least-loaded routing keeps the engines level
    pool.close(); single.close()


def test_large_v3_64_concurrent_streams_equal_serial():
    """configs[3] ("ggml-large-v3 gRPC stream, 64 concurrent") at its own size on one GPU: 64 streams of 11 s (two 5 s chunks + the final flush
    each; every chunk costs a full 1500-position encoder pass, /root/reference/src/asr/whisper.rs:68,144) through the batch former of the real
    large-v3 engine, Mode F 12 decode steps per chunk.  Responses of every stream must equal what the same stream gets from a serial engine."""
    sys.path.insert(0, ROOT)
    import bench
    from speaksense_amd import asr, binding, ggml_io, stream
    path = bench.model_path_for("large-v3")
    if not os.path.exists(path):
        ggml_io.write_model(path + ".tmp", "large-v3", seed=0)
        os.replace(path + ".tmp", path)
    n_streams, seconds = 64, 11
    msgs = [stream.client_messages(synth.speech_like(500 + i % 8, 16000 * seconds)) for i in range(n_streams)]     # 8 distinct streams, 8 copies each

    def norm(resps):
        return [(r.end, r.text, [(s.start, s.end, s.text) for s in r.segments]) for r in resps]
    serial = asr.WhisperAsr(path, max_batch=8)
    serial.params_hook = lambda p: setattr(p, "fixed_steps", 12)
    want = []
    for i in range(8):
        ses = stream.GrpcStreamSession(serial)
        out = []
        for m, end in msgs[i]:
            out.extend(ses.feed(m, end, f"stream-{i}"))
        want.append(norm(out))
    serial.engine.close()
    conc = asr.WhisperAsr(path, max_batch=16, batch_across_callers=True, batch_wait_us=3000)
    conc.params_hook = lambda p: setattr(p, "fixed_steps", 12)
    got = stream.serve_streams(conc, msgs)
    tot = conc.engine.totals()
    conc.engine.close()
    assert len(got) == n_streams
    for i in range(n_streams):
        assert norm(got[i]) == want[i % 8], f"stream {i}"
        assert len(got[i]) >= 1 and got[i][-1].end == 1
    assert tot["decoder_rows"] / max(1, tot["decoder_passes"]) > 4.0      # the chunks of different streams really shared decoder passes
