#!/usr/bin/env python
"""bench.py -- headline benchmark of the MI355X Whisper path (BASELINE.json: audio-sec/s (xRT) + p50 chunk latency,
Whisper large-v3, 30 s chunks, batch 8 per GPU).

A "step" is one pass of the hot path over one batch of synthetic input: B independent 30 s chunks (16 kHz mono f32,
already resident in HBM) -> log-mel -> encoder -> cross-KV -> KV-cached decoder with on-device logits rules -> token ids
and segments on the host.  Mode F: one encoder window + prompt + exactly --fixed-steps greedy steps,
EOT suppressed, so FLOPs/bytes per chunk are deterministic and weight-value independent (weights are seeded random
in ggml format; no real checkpoints exist offline).

Multi-GPU: one process per GPU (torchrun), chunks assigned round-robin to ranks, no data-path collective (chunks are
independent: SURVEY.md §8e); only the timing barrier and the MAX-over-ranks reduction use torch.distributed (RCCL).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CHUNK_SEC = 30.0
MFMA_PEAK_TFLOPS = 2500.0   # dense bf16/f16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def shard_chunks(n_total: int, world: int, rank: int) -> list[int]:
    """Round-robin assignment of independent chunk ids to ranks (north_star: 'sharded round-robin across the 8 GPUs')."""
    return list(range(rank, n_total, world))


def dist_env():
    return int(os.environ.get("RANK", 0)), int(os.environ.get("LOCAL_RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))


def timed_steps(step_fn, steps: int, warmup: int, dist, sync):
    """W untimed warmup steps, then EXACTLY K steps bracketed by barrier + device sync; returns (max-over-ranks seconds, per-step seconds)."""
    import torch
    for _ in range(warmup):
        step_fn()
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    per = []
    for _ in range(steps):
        ts = time.perf_counter()
        step_fn()
        per.append(time.perf_counter() - ts)
    sync()
    if dist is not None:
        dist.barrier()
    sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if (torch.cuda.is_available() and dist.get_backend() == "nccl") else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt, per


def model_path_for(name: str, seed: int = 0) -> str:
    d = os.environ.get("SS_MODEL_DIR", "/tmp/ss_models")
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, f"ggml-{name}-synthetic-s{seed}.bin")


def ensure_model(name: str, local_rank: int, dist) -> str:
    from speaksense_amd import ggml_io
    path = model_path_for(name)
    real = os.environ.get("ASR_MODEL_PATH")   # a real ggml-*.bin if the box has one (the reference's env var, src/lib.rs:24)
    if real and os.path.exists(real):
        return real
    if local_rank == 0 and not os.path.exists(path):
        t = time.time()
        ggml_io.write_model(path + ".tmp", name, seed=0)
        os.replace(path + ".tmp", path)
        print(f"[bench] wrote synthetic {name} ggml model ({os.path.getsize(path) / 1e9:.2f} GB) in {time.time() - t:.1f}s", file=sys.stderr)
    if dist is not None:
        dist.barrier()
    return path


def algorithmic_work(hp, batch: int, n_steps: int, n_prompt: int):
    """FLOPs per chunk (SURVEY.md §8d conventions: 2*MAC, attention 2*T^2*d each for QK^T and AV) and decoder bytes per step."""
    T, d, L, dt, Lt, V = hp.n_audio_ctx, hp.n_audio_state, hp.n_audio_layer, hp.n_text_state, hp.n_text_layer, hp.n_vocab
    conv = 2 * (2 * T) * d * 3 * hp.n_mels + 2 * T * d * 3 * d
    enc_layer = 2 * T * d * d * 4 + 2 * 2 * T * T * d + 2 * T * d * 4 * d * 2
    cross = Lt * 2 * 2 * T * d * dt
    dec_tok = Lt * (2 * dt * dt * 4 + 2 * dt * dt * 2 + 2 * 2 * T * dt + 2 * dt * 4 * dt * 2) + 2 * dt * V
    flops_chunk = conv + L * enc_layer + cross + (n_steps + n_prompt) * dec_tok
    dec_weight_bytes = 2 * (Lt * (4 * dt * dt + 2 * dt * dt + 8 * dt * dt) + dt * V)
    cross_kv_bytes = 2 * Lt * 2 * T * dt
    return dict(flops_chunk=float(flops_chunk), enc_flops=float(conv + L * enc_layer + cross), dec_tok_flops=float(dec_tok),
                dec_bytes_step=float(dec_weight_bytes + batch * cross_kv_bytes))


def pmc_traffic(model: str, batch: int, dtype: str):
    """HBM-side bytes per FC1 launch from the committed rocprofv3 --pmc passes of this command (profiles/pmc_traffic.json):
    (2 x FETCH_SIZE + WRITE_SIZE) KiB, the x2 being the gfx950 FETCH_SIZE correction for 16-B/lane streams
    (MI355X_MICROARCH.md, HBM section; validated here on the cross-attention kernel: 62.2 MB measured vs 61.4 MB algorithmic)."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        j = json.load(open(p))
        k = j.get(f"{model}/batch{batch}/{dtype}")
        return None if k is None else float(k["bytes_per_launch"])
    except Exception:
        return None


def cpu_baseline(path: str, hp, n_steps: int, n_prompt: int):
    """The CPU restatement (oracle, kind 'port') timed on the box's host cores on a bounded sample of the same workload."""
    from oracle import binding as orc
    from speaksense_amd import synth
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    cores = max(1, min(64, ncpu))
    om = orc.OracleModel(path)
    n_enc, n_cross, n_dec = 2, 2, 6
    t = om.time_sample(synth.speech_like(0), orc.MODE_GGML_F16, n_enc, n_cross, n_dec, cores)
    om.close()
    chunk_s = t["mel_s"] + t["stem_s"] + hp.n_audio_layer * t["enc_layer_s"] + hp.n_text_layer * t["cross_layer_s"] + (n_steps + n_prompt) * t["dec_step_s"]
    return {"value": round(CHUNK_SEC / chunk_s, 4), "unit": "audio-sec/s", "cores": cores, "kind": "port",
            "sample": f"1 chunk: log-mel + conv stem + {n_enc}/{hp.n_audio_layer} encoder layers + {n_cross}/{hp.n_text_layer} cross-KV layers + "
                      f"{n_dec} decode steps timed, extrapolated to {hp.n_audio_layer} layers and {n_steps + n_prompt} decoder positions "
                      f"(est. {chunk_s:.1f} s per 30 s chunk); oracle/whisper_oracle.cpp ggml-f16 mode, {cores} OpenMP threads"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--batch", type=int, default=8, help="30 s chunks per GPU per step")
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16"])
    ap.add_argument("--fixed-steps", type=int, default=96, help="Mode F decode steps per chunk; 0 = Mode N (natural EOT, full whisper.cpp rules)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dry-run", action="store_true", help="CPU test of the sharding/timing plumbing: stub workload, gloo backend")
    ap.add_argument("--dist-backend", default=None, help="override (default nccl on GPU); 'gloo' + SS_BENCH_DEVICE=0 lets several ranks share one GPU for testing")
    args = ap.parse_args()

    rank, local_rank, world = dist_env()
    if os.environ.get("SS_BENCH_DEVICE"):   # test hook: all ranks on one GPU
        local_rank_dev = int(os.environ["SS_BENCH_DEVICE"])
    else:
        local_rank_dev = local_rank
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = args.dist_backend or ("gloo" if args.dry_run or not torch.cuda.is_available() else "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank_dev)
        dist_mod.init_process_group(backend=backend, rank=rank, world_size=world)
        dist = dist_mod
    n_gpus = world
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE {world}; using WORLD_SIZE", file=sys.stderr)

    my_chunks = shard_chunks(n_gpus * args.batch, world, rank)

    if args.dry_run:
        def step():
            time.sleep(0.01 * len(my_chunks))
        dt, per = timed_steps(step, args.steps, args.warmup, dist, lambda: None)
        if rank == 0:
            print(json.dumps({"metric": "dry-run", "value": n_gpus * args.batch * args.steps * CHUNK_SEC / dt, "unit": "audio-sec/s",
                              "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "chunks_rank0": my_chunks}))
        if dist is not None:
            dist.destroy_process_group()
        return

    from speaksense_amd import binding, ggml_io, synth
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the HIP path has no CPU fallback); use --dry-run to test the plumbing")
    torch.cuda.set_device(local_rank_dev)
    path = ensure_model(args.model, local_rank, dist)
    hp = ggml_io.PRESETS.get(args.model)
    eng = binding.Engine(path, device=local_rank_dev, dtype=binding.DTYPE_F16 if args.dtype == "f16" else binding.DTYPE_BF16, max_batch=args.batch)
    if hp is None:
        hp = ggml_io.HParams(eng.n_vocab, eng.n_audio_ctx, eng.n_audio_state, eng.n_audio_head, eng.n_audio_layer, eng.n_text_ctx,
                             eng.n_text_state, eng.n_text_head, eng.n_text_layer, eng.n_mels, eng.ftype)
    # synthetic audio, one seed per global chunk id, uploaded before the timed region
    pcm = torch.stack([torch.from_numpy(synth.speech_like(cid)) for cid in my_chunks]).cuda()
    ptrs = [(pcm[i].data_ptr(), pcm.shape[1]) for i in range(len(my_chunks))]
    sessions = [eng.new_session() for _ in my_chunks]
    P = binding.default_params(language="en", fixed_steps=args.fixed_steps)
    n_prompt = 3 if eng.n_vocab >= 51865 else 1
    tok_counts = []
    timings = []

    def step():
        res = eng.transcribe_batch(sessions, None, P, device_ptrs=ptrs)
        tok_counts.append(sum(len(r["tokens"]) for r in res))
        timings.append(eng.last_timing())

    dt, per = timed_steps(step, args.steps, args.warmup, dist, torch.cuda.synchronize)
    audio_sec = n_gpus * args.batch * args.steps * CHUNK_SEC
    value = audio_sec / dt

    if rank == 0:
        tl = timings[-args.steps:]
        n_steps_dec = args.fixed_steps if args.fixed_steps > 0 else int(np.mean(tok_counts[-args.steps:]) / max(1, len(my_chunks)))
        work = algorithmic_work(hp, args.batch, n_steps_dec, n_prompt)
        enc_ms = float(np.mean([t["encode_ms"] for t in tl])); dec_ms = float(np.mean([t["decode_ms"] for t in tl]))
        # roofline of the dominant kernel: the encoder MFMA GEMM (gemm_kernel), measured with HIP events on the engine's stream
        gemm_ms, gemm_flops = eng.probe_gemm(args.batch, 20)
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12
        out = {
            "metric": "audio-sec/s (xRT) + p50 chunk latency, Whisper large-v3 30s chunks @1/8 GPU",
            "value": round(value, 2), "unit": "audio-sec/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"ggml-{args.model} batch={args.batch}x30s chunks per GPU, "
                                   + (f"Mode F: 1 encoder window + {n_prompt}-token prompt + {args.fixed_steps} greedy steps" if args.fixed_steps > 0
                                      else "Mode N: natural EOT, whisper.cpp fallback rules"),
                       "weights": "seeded random, ggml legacy format" if "synthetic" in path else path,
                       "chunks_per_step": n_gpus * args.batch, "parallelism": f"dp{n_gpus} (independent chunks, no collective)"},
            "p50_chunk_latency_ms": round(1e3 * float(np.median(per)), 2),
            "phase_ms": {"mel": round(float(np.mean([t["mel_ms"] for t in tl])), 3), "encode_cross_kv": round(enc_ms, 2), "decode": round(dec_ms, 2)},
            "roofline": {"bound": "mfma", "kernel": "gemm256_kernel<T, EPI_GELU_T> (encoder FC1: M=batch*1500, N=4d, K=d, fused bias+GELU)",
                         "achieved": round(achieved, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                         "traffic": pmc_traffic(args.model, args.batch, args.dtype), "algorithmic_bytes": 2.0 * (args.batch * hp.n_audio_ctx * hp.n_audio_state + 4 * hp.n_audio_state * hp.n_audio_state + 4 * args.batch * hp.n_audio_ctx * hp.n_audio_state),
                         "avg_launch_ms": round(gemm_ms, 4)},
            "phase_roofline": {
                "encoder_phase_tflops": round(args.batch * work["enc_flops"] / (enc_ms * 1e-3) / 1e12, 1),
                "encoder_phase_frac_mfma": round(args.batch * work["enc_flops"] / (enc_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS, 4),
                "decode_step_ms": round(dec_ms / max(1, n_steps_dec + 1), 4),
                "decode_hbm_gbs": round(work["dec_bytes_step"] * (n_steps_dec + 1) / (dec_ms * 1e-3) / 1e9, 1),
                "decode_frac_hbm": round(work["dec_bytes_step"] * (n_steps_dec + 1) / (dec_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "whole_chunk_tflops": round(n_gpus * args.batch * work["flops_chunk"] * args.steps / dt / 1e12, 1)},
        }
        if n_gpus == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(path, hp, n_steps_dec, n_prompt)
            except Exception as e:  # never fabricate a number
                out["cpu_baseline"] = {"value": None, "unit": "audio-sec/s", "cores": 0, "kind": "port", "sample": f"unavailable: {e}"}
        print(json.dumps(out))
    eng.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
