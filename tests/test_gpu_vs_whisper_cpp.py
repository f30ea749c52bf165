"""GPU: parity against the REFERENCE's own arithmetic -- whisper.cpp (whisper-rs 0.11.1 -> whisper-rs-sys 0.9.0, /root/reference/Cargo.lock:3888-3907;
call site /root/reference/src/asr/whisper.rs:75) on real ggml weights -- whenever a box has them.

Neither whisper.cpp nor any ggml-*.bin exists in the build image or on the offline GPU box, so the oracle this repo is
tested against is "parity unpinned".  This test closes that automatically, with no manual step, on any machine where both are present:

    WHISPER_CPP_MAIN = path of whisper.cpp's `main` / `whisper-cli` binary (v1.5.x CPU build)
    ASR_MODEL_PATH   = the ggml-*.bin the service loads (the reference's own variable, /root/reference/src/lib.rs:24)
    WHISPER_CPP_WAVS = optional, os.pathsep-separated 16 kHz 16-bit WAV files (default: whisper.cpp's samples/jfk.wav next to the binary)

It runs tools/compare_with_whisper_cpp.py: whisper.cpp (`-t 16 -bo 5 -ojf`, the reference's parameters) and the MI355X engine on the same
samples; token ids, segment boundaries and text must be identical, or every divergence must be proven a near tie by a forced replay on the
oracle (--oracle).  Elsewhere it SKIPS -- loudly: a skipped pin is not a green one."""
import os
import shutil
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def whisper_cpp_setup():
    """(main binary, model path, wavs) or (None, reason)."""
    main = os.environ.get("WHISPER_CPP_MAIN") or shutil.which("whisper-cli") or shutil.which("whisper-cpp")
    model = os.environ.get("ASR_MODEL_PATH")
    if not main or not os.path.exists(main):
        return None, "no whisper.cpp binary (set WHISPER_CPP_MAIN)"
    if not model or not os.path.exists(model):
        return None, "no real ggml model (set ASR_MODEL_PATH, /root/reference/src/lib.rs:24)"
    wavs = [w for w in os.environ.get("WHISPER_CPP_WAVS", "").split(os.pathsep) if w]
    if not wavs:
        for cand in (os.path.join(os.path.dirname(main), "samples", "jfk.wav"), os.path.join(os.path.dirname(main), "..", "samples", "jfk.wav"),
                     os.path.join(os.path.dirname(main), "..", "..", "samples", "jfk.wav")):
            if os.path.exists(cand):
                wavs = [cand]
                break
    if not wavs or not all(os.path.exists(w) for w in wavs):
        return None, "no WAV input (set WHISPER_CPP_WAVS)"
    return (main, model, wavs), ""


def test_token_ids_identical_to_whisper_cpp_on_real_weights():
    setup, why = whisper_cpp_setup()
    if setup is None:
        pytest.skip("PARITY STAYS UNPINNED ON THIS BOX: " + why + " -- the oracle is anchored to HF transformers only (DESIGN.md section 2)")
    main, model, wavs = setup
    cmd = [sys.executable, os.path.join(ROOT, "tools", "compare_with_whisper_cpp.py"), "--main", main, "--model", model, "--oracle"] + wavs
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=3600)
    sys.stdout.write(r.stdout[-6000:])
    sys.stderr.write(r.stderr[-3000:])
    assert r.returncode != 2, "the engine's DEFAULT behaviour does not match this whisper.cpp but another SS_COMPAT variant does (report above; DESIGN.md 2a)"
    assert r.returncode == 0, "the MI355X engine and whisper.cpp disagree beyond proven near ties under every variant tried (see the report above)"
