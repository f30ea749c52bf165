"""GPU: the gRPC stream handler mirror (speaksense_amd/stream.py; /root/reference/src/grpc/handlers/asr.rs:146-280) end to end.
Expected responses come from the same message schedule with every transcription done by the CPU oracle (GGML_F16 mode) on the same
denoised samples; the device denoiser has its own parity test (test_gpu_denoise.py)."""
import numpy as np
import pytest

from oracle import binding as orc
from speaksense_amd import asr, stream, synth

pytestmark = pytest.mark.gpu


class OracleAsr(asr.WhisperAsr):
    """Same trait surface, transcription by the oracle; `engine` (the device engine) is kept for denoise_audio only."""

    def __init__(self, om, gpu_asr):
        self.om, self.engine = om, gpu_asr.engine
        self.n_fail = 0          # windows that fell back to temperature sampling (draws near a CDF boundary may then differ legitimately)

    def create_state(self):
        return self.om.new_state(orc.MODE_GGML_F16)

    def transcribe_with_state(self, state, audio, user_params):
        bp = self.build_params(user_params)
        p = orc.default_params(language=bp.language, no_context=bp.no_context, tdrz_enable=bp.tdrz_enable, single_segment=bp.single_segment,
                               temperature_inc=bp.temperature_inc)
        res = state.full(np.asarray(audio, np.float32), p)
        self.n_fail += res["n_fail"]
        return self._collect(res, user_params)


def _run(session, msgs, device_id="dev-7"):
    out = []
    for m, end in msgs:
        out.extend(session.feed(m, end, device_id))
    return [(r.end, r.text, r.device_id, [(s.start, s.end, s.text) for s in r.segments]) for r in out]


@pytest.mark.parametrize("seconds,msg_bytes", [(13.0, 32 * 1024), (5.2, 7001), (3.0, 32 * 1024)])
def test_stream_session_matches_oracle_driven_schedule(toy_ml_path, seconds, msg_bytes):
    om = orc.OracleModel(toy_ml_path)
    gpu = asr.WhisperAsr(toy_ml_path, max_batch=4)
    pcm = synth.speech_like(11, int(16000 * seconds))
    msgs = stream.client_messages(pcm, msg_bytes)
    got = _run(stream.GrpcStreamSession(gpu), msgs)
    oa = OracleAsr(om, gpu)
    want = _run(stream.GrpcStreamSession(oa), msgs)
    # inputs chosen (tools/find_nofallback_seeds.py) so that no window leaves temperature 0: nothing is sampled, the comparison is exact
    assert oa.n_fail == 0, "fixture drifted: the oracle fell back to sampling; re-pick the seed with tools/find_nofallback_seeds.py"
    assert got == want
    n_chunks = 0
    buf = int(16000 * seconds) * 2
    # chunk count of the reference loop: at most one 5 s chunk per request message (asr.rs:185 is an `if`, not a `while`)
    b = 0
    for m, _ in msgs:
        b += len(__import__("base64").b64decode(m))
        if b >= stream.CHUNK_SIZE:
            n_chunks += 1
            b -= stream.CHUNK_SIZE - stream.OVERLAP_SIZE
    assert buf > 0 and got[-1][0] == 1 and got[-1][2] == "dev-7"          # the end==1 flush answers with end = 1
    assert sum(1 for g in got if g[0] == 0) <= max(n_chunks, 0) * 8       # chunk responses only while chunks were cut
    # times are absolute milliseconds and never run backwards (asr.rs:45-54)
    ends = [s[1] for g in got for s in g[3]]
    starts = [s[0] for g in got for s in g[3]]
    assert all(e >= s for s, e in zip(starts, ends)) and all(starts[i + 1] >= ends[i] for i in range(len(ends) - 1))
    gpu.engine.close()


def test_stream_bad_base64_is_skipped(toy_ml_path):
    gpu = asr.WhisperAsr(toy_ml_path, max_batch=2)
    s = stream.GrpcStreamSession(gpu)
    assert s.feed(b"!!!not base64!!!", 0, "d") == []
    assert len(s.buf) == 0
    pcm = synth.speech_like(5, 16000 * 2)
    out = []
    for m, end in stream.client_messages(pcm):
        out += s.feed(m, end, "d")
    assert s.done and s.feed(b"", 1, "d") == []
    gpu.engine.close()


def test_concurrent_streams_equal_serial(toy_ml_path):
    """8 concurrent streams through the batch former give each stream exactly what it gets alone."""
    serial = asr.WhisperAsr(toy_ml_path, max_batch=4)
    streams = [stream.client_messages(synth.speech_like(40 + i, 16000 * 11)) for i in range(8)]
    want = [_run(stream.GrpcStreamSession(serial), m, f"stream-{i}") for i, m in enumerate(streams)]
    serial.engine.close()
    conc = asr.WhisperAsr(toy_ml_path, max_batch=4, batch_across_callers=True, batch_wait_us=5000)
    got = stream.serve_streams(conc, streams)
    got = [[(r.end, r.text, r.device_id, [(s.start, s.end, s.text) for s in r.segments]) for r in g] for g in got]
    assert got == want
    conc.engine.close()
