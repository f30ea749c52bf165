"""Config #4's shape on ONE GPU (BASELINE.json configs[3] puts 8 of the 64 streams on each of 8 GPUs): N concurrent gRPC-style streams, each
replaying the client's wire format (PCM16 -> base64 -> 32 KiB messages) through speaksense_amd.stream.GrpcStreamSession -- 5 s chunks with 0.5 s
overlap, denoise on the device, transcribe on the shared engine whose batch former groups the chunks of different streams.  Every 5 s chunk
costs a full 1500-position encoder pass (audio_ctx = 0), exactly as in the reference.
Prints one JSON line: aggregate audio-seconds per second and the p50 / p95 latency of a chunk (submit -> responses).

    python tools/stream_bench.py --streams 64 --seconds 30 [--model large-v3] [--max-batch 8]
"""
import argparse, json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speaksense_amd import asr, stream, synth  # noqa: E402
import bench  # noqa: E402  (model cache helper)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--model", default="large-v3")
    ap.add_argument("--max-batch", type=int, default=8)
    ap.add_argument("--batch-wait-us", type=int, default=3000)
    ap.add_argument("--dtype", default="f16", choices=["f16", "bf16", "fp8"])
    ap.add_argument("--fixed-steps", type=int, default=24, help="Mode F decode length per chunk (random weights never reach a natural EOT without walking the fallback ladder); 0 = natural")
    a = ap.parse_args()
    path = bench.ensure_model(a.model, 0, None)
    from speaksense_amd import binding
    eng = asr.WhisperAsr(path, dtype={"f16": binding.DTYPE_F16, "bf16": binding.DTYPE_BF16, "fp8": binding.DTYPE_FP8}[a.dtype], max_batch=a.max_batch,
                         batch_across_callers=True, batch_wait_us=a.batch_wait_us)
    if a.fixed_steps > 0:
        eng.params_hook = lambda p: setattr(p, "fixed_steps", a.fixed_steps)
    msgs = [stream.client_messages(synth.speech_like(100 + i, int(16000 * a.seconds))) for i in range(a.streams)]
    # warm-up: one short stream (kernel attributes, graphs)
    stream.serve_streams(eng, [stream.client_messages(synth.speech_like(1, 16000 * 6))])
    lat = []
    lock = threading.Lock()
    orig_feed = stream.GrpcStreamSession.feed

    def timed_feed(self, audio_b64, end=0, device_id=""):
        t0 = time.perf_counter()
        n_before = len(self.buf)
        out = orig_feed(self, audio_b64, end, device_id)
        dt = time.perf_counter() - t0
        if dt > 1e-3:      # a message that triggered a chunk (or the final flush)
            with lock:
                lat.append(dt)
        return out

    stream.GrpcStreamSession.feed = timed_feed
    t0 = time.perf_counter()
    res = stream.serve_streams(eng, msgs)
    wall = time.perf_counter() - t0
    stream.GrpcStreamSession.feed = orig_feed
    n_resp = sum(len(r) for r in res)
    lat = np.array(sorted(lat)) * 1e3
    print(json.dumps({"dtype": a.dtype, "workload": f"{a.streams} concurrent streams x {a.seconds:.0f} s, 5 s chunks + final flush, {a.model}, max_batch {a.max_batch}, " + (f"Mode F {a.fixed_steps} decode steps per chunk" if a.fixed_steps else "natural EOT"),
                      "audio_sec_per_sec": round(a.streams * a.seconds / wall, 1), "wall_s": round(wall, 3), "chunks": int(len(lat)), "responses": n_resp,
                      "chunk_latency_ms_p50": round(float(np.percentile(lat, 50)), 1), "chunk_latency_ms_p95": round(float(np.percentile(lat, 95)), 1)}))
    eng.engine.close()


if __name__ == "__main__":
    main()
