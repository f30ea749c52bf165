"""CPU: model-file I/O, the host-side mirror of the reference's src/asr interface, and the C-ABI surface
(the library must load without a GPU and export every symbol include/*.h declares; compute needs a GPU and must fail loudly)."""
import ctypes
import os
import re

import numpy as np
import pytest

from speaksense_amd import asr, ggml_io

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ggml_roundtrip(tmp_path):
    p = str(tmp_path / "m.bin")
    hp = ggml_io.write_model(p, "toy", seed=3)
    hp2, filt, vocab, t = ggml_io.read_model(p)
    assert hp2 == hp and filt.shape == (128, 201)
    assert len(vocab) == 50257 and vocab[220] == b" "
    names = [n for n, _, _ in ggml_io.tensor_specs(hp)]
    assert set(names) == set(t)
    assert t["encoder.conv1.weight"].shape == (128, 128, 3)
    assert t["decoder.blocks.1.cross_attn.key.weight"].shape == (128, 128)
    assert "decoder.blocks.0.attn.key.bias" not in t          # whisper has no key bias
    w = t["decoder.blocks.0.mlp.0.weight"]
    assert np.array_equal(w, w.astype(np.float16).astype(np.float32))  # 2-D weights are stored f16
    # same seed -> same bytes
    p2 = str(tmp_path / "m2.bin")
    ggml_io.write_model(p2, "toy", seed=3)
    assert open(p, "rb").read() == open(p2, "rb").read()


def test_presets_match_survey():
    lv3 = ggml_io.PRESETS["large-v3"]
    assert (lv3.n_vocab, lv3.n_audio_state, lv3.n_audio_head, lv3.n_audio_layer, lv3.n_mels) == (51866, 1280, 20, 32, 128)
    n_params = sum(int(np.prod(s)) for _, s, _ in ggml_io.tensor_specs(lv3))
    assert 1.50e9 < n_params < 1.60e9   # 1541 M


def test_presets_of_the_download_script(tmp_path):
    """The other models /root/reference/script/download-ggml-model.sh:36-48 fetches: parameter counts of the published checkpoints, and the
    special-token table of the 51865-token vocabulary (99 languages: every id after the language block sits one lower than in large-v3)."""
    from oracle import binding as orc
    def n_params(name):
        return sum(int(np.prod(s)) for _, s, _ in ggml_io.tensor_specs(ggml_io.PRESETS[name]))
    assert 0.80e9 < n_params("large-v3-turbo") < 0.82e9          # 809 M
    assert 0.76e9 < n_params("medium") < 0.78e9                  # 769 M
    assert 0.24e9 < n_params("small.en") < 0.25e9                # 244 M
    t = ggml_io.PRESETS["large-v3-turbo"]
    assert (t.n_audio_layer, t.n_text_layer) == (32, 4)
    path = str(tmp_path / "m.bin")
    ggml_io.write_model(path, ggml_io.HParams(51865, 1500, 128, 2, 1, 448, 128, 2, 1, 80, 1), seed=2)
    om = orc.OracleModel(path)
    assert (om.eot, om.sot, om.translate, om.transcribe, om.solm, om.prev, om.nosp, om.not_, om.beg) == (50257, 50258, 50358, 50359, 50360, 50361, 50362, 50363, 50364)
    om.close()


def test_punctuation_and_promo_filter():
    # /root/reference/src/asr/whisper.rs:175-201
    assert asr.add_punctuation("你好吗") == "你好吗？"
    assert asr.add_punctuation("太棒了") == "太棒了！"
    assert asr.add_punctuation("hello") == "hello "
    assert asr.add_punctuation("结束。") == "结束。"
    assert asr.add_punctuation("为何这样真好") == "为何这样真好？"   # question wins over exclamation
    # whisper.rs:41-43
    assert asr.is_promotional_text("感谢观看 请不吝点赞 订阅")
    assert not asr.is_promotional_text("today's weather")


def test_collect_semantics():
    """whisper.rs:77-128: promo segments dropped, speaker id increments on speaker_turn_next of the PREVIOUS segment,
    stream mode keeps only the last segment, timestamps passed through as f64 centiseconds."""
    w = asr.WhisperAsr.__new__(asr.WhisperAsr)
    res = dict(segments=[
        dict(text="第一段".encode(), t0=0, t1=250, speaker_turn_next=True),
        dict(text="请不吝点赞".encode(), t0=250, t1=300, speaker_turn_next=False),
        dict(text="second".encode(), t0=300, t1=512, speaker_turn_next=False),
    ])
    out = w._collect(res, asr.AsrParams(stream_mode=False))
    assert [s.text for s in out.segments] == ["第一段 ", "second "]
    # the promo segment `continue`s before the speaker check (whisper.rs:87-97), so segment 0's turn flag is lost -- mirrored
    assert [s.speaker_id for s in out.segments] == [0, 0]
    res2 = dict(segments=[dict(text=b"a", t0=0, t1=1, speaker_turn_next=True), dict(text=b"b", t0=1, t1=2, speaker_turn_next=True),
                          dict(text=b"c", t0=2, t1=3, speaker_turn_next=False)])
    assert [s.speaker_id for s in w._collect(res2, asr.AsrParams()).segments] == [0, 1, 2]
    assert out.segments[1].start == 300.0 and out.segments[1].end == 512.0
    assert out.full_text == "第一段 second "
    out = w._collect(res, asr.AsrParams(stream_mode=True))
    assert len(out.segments) == 1 and out.full_text == "second "
    bad = dict(segments=[dict(text=b"\xff\xfe", t0=0, t1=1, speaker_turn_next=False)])
    with pytest.raises(UnicodeDecodeError):   # strict UTF-8, as full_get_segment_text (whisper.rs:85)
        w._collect(bad, asr.AsrParams())


def _declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    return sorted(set(re.findall(r"\b((?:ss|whisper)_[a-z0-9_]+)\s*\(", src)) - {"whisper_new_segment_callback"})


@pytest.mark.parametrize("header", ["speaksense.h", "whisper_compat.h"])
def test_library_exports_every_declared_symbol(header):
    from speaksense_amd import binding
    assert os.path.exists(binding.LIB_PATH), "libspeaksense_hip.so not built (python __graft_entry__.py)"
    L = ctypes.CDLL(binding.LIB_PATH)
    fns = _declared_functions(header)
    assert len(fns) >= 20
    for f in fns:
        assert hasattr(L, f), f"{f} declared in include/{header} but not exported"


def test_no_cpu_fallback(toy_ml_path):
    """Without a GPU the product path must fail loudly (SS_ERR_DEVICE), never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from speaksense_amd import binding
    with pytest.raises(binding.SpeakSenseError) as e:
        binding.Engine(toy_ml_path)
    assert e.value.code == -4


def _abi_layout(tmp_path, defines=()):
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / ("layout" + "".join(defines)))
    r = subprocess.run(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror"] + ["-D" + d for d in defines] + ["-I", os.path.join(root, "include"),
                        os.path.join(root, "tests", "c_harness", "layout.c"), "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    out = subprocess.run([exe], capture_output=True, text=True).stdout
    return out, {(a, b): (int(c), int(d)) for a, b, c, d in (l.split() for l in out.strip().splitlines())}


def test_params_struct_layout_matches_header(tmp_path):
    """Both by-value parameter structs, field by field: what a C11 compiler makes of the public headers == the committed table
    (tests/golden/abi_layout.txt; the whisper_full_params part was derived by hand from whisper.h v1.5.4 for LP64 -- a maintainer diffs it
    against bindgen's layout test of the vendored header) == the ctypes mirror the Python binding passes."""
    import os
    from speaksense_amd import binding
    out, lay = _abi_layout(tmp_path)
    golden = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "abi_layout.txt")).read()
    assert out == golden, "include/*.h no longer lay out as the committed ABI table"
    assert lay[("whisper_full_params", "sizeof")][0] == 256 and lay[("whisper_full_params", "language")] == (88, 8)
    assert lay[("whisper_full_params", "greedy")] == (128, 4) and lay[("whisper_full_params", "grammar_penalty")] == (248, 4)
    for name, ctype in binding.Params._fields_:
        f = getattr(binding.Params, name)
        assert lay[("ss_params", name)] == (f.offset, f.size), name
    assert lay[("ss_params", "sizeof")][0] == ctypes.sizeof(binding.Params)
    assert len([k for k in lay if k[0] == "ss_params"]) == len(binding.Params._fields_) + 1
    for name, ctype in binding.EngineOpts._fields_:
        f = getattr(binding.EngineOpts, name)
        assert lay[("ss_engine_opts", name)] == (f.offset, f.size), name
    p = binding.default_params()
    assert (p.best_of, p.no_context, p.suppress_blank, p.language, p.n_max_text_ctx, p.offset_ms, p.prompt_n_tokens) == (5, 1, 1, b"en", 16384, 0, 0)
    assert abs(p.temperature_inc - 0.2) < 1e-7 and abs(p.entropy_thold - 2.4) < 1e-6 and p.logprob_thold == -1.0
    assert ctypes.sizeof(binding.EngineOpts) == 32


def test_whisper_h_post_1_5_4_layout(tmp_path):
    """The second, build-selectable snapshot of whisper.h (include/whisper_compat.h, -DSS_WHISPER_H_POST_1_5_4 = v1.5.5: gpu_device + DTW fields in
    whisper_context_params, t_dtw in whisper_token_data): field by field against its own committed table; everything else is the v1.5.4 table; and the
    library built with that macro (libspeaksense_whisper_post154.so) exports the whole whisper_* list and binds its own whisper_* calls to itself."""
    import os, subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    out, lay = _abi_layout(tmp_path, ("SS_WHISPER_H_POST_1_5_4",))
    assert out == open(os.path.join(here, "golden", "abi_layout_post_1_5_4.txt")).read()
    base, _ = _abi_layout(tmp_path)
    keep = lambda t: [l for l in t.splitlines() if not l.startswith(("whisper_context_params", "whisper_token_data"))]
    assert keep(out) == keep(base)
    assert lay[("whisper_context_params", "sizeof")][0] == 48 and lay[("whisper_context_params", "dtw_aheads")] == (24, 16)
    assert lay[("whisper_token_data", "t_dtw")] == (40, 8) and lay[("whisper_token_data", "sizeof")][0] == 56
    from speaksense_amd import build
    build.build()
    assert os.path.exists(build.LIB_W155)
    sym = subprocess.run(["nm", "-D", "--defined-only", build.LIB_W155], capture_output=True, text=True).stdout
    have = {l.split()[-1] for l in sym.splitlines() if " T " in l}
    main = subprocess.run(["nm", "-D", "--defined-only", build.LIB], capture_output=True, text=True).stdout
    want = {l.split()[-1] for l in main.splitlines() if " T whisper_" in l}
    assert want and want <= have, sorted(want - have)
    dyn = subprocess.run(["readelf", "-d", build.LIB_W155], capture_output=True, text=True).stdout
    assert "SYMBOLIC" in dyn and "libspeaksense_hip.so" in dyn


def _build_c_harness(tmp_path):
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from speaksense_amd import build
    lib = build.build()
    exe = str(tmp_path / "harness")
    cmd = ["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(root, "include"), os.path.join(root, "tests", "c_harness", "harness.c"), "-o", exe,
           "-L", os.path.dirname(lib), "-lspeaksense_hip", "-lm", f"-Wl,-rpath,{os.path.dirname(lib)}", "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def test_public_headers_compile_as_c_and_link(tmp_path):
    """include/speaksense.h and include/whisper_compat.h are consumed by a C11 translation unit (what bindgen / cgo / a C service would do) and
    every symbol it uses resolves against libspeaksense_hip.so.  Running it needs a GPU (tests/test_gpu_variants.py)."""
    import subprocess
    exe = _build_c_harness(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def test_tokenizer_matches_regex_restatement(tmp_path):
    """whisper_tokenize (initial_prompt): the product's hand-written GPT-2 pre-split (csrc/model.cpp) against the oracle, which restates
    whisper.cpp's tokenize() with the same std::regex pattern -- two independent implementations, host only."""
    import random
    from oracle import binding as orc
    from speaksense_amd import binding, ggml_io
    path = str(tmp_path / "toy.bin")
    ggml_io.write_model(path, "toy", seed=1)
    om = orc.OracleModel(path)
    rnd = random.Random(1)
    texts = [b"", b"  ", b"a  b   c    ", b"'re're 'll x'd", "什么 好 ".encode(), b"it's 12  x\t\n y"]
    for _ in range(200):
        t = b"".join(om.token_str(rnd.randint(0, 50000)) for _ in range(rnd.randint(1, 12)))
        if rnd.random() < 0.3:
            t = t.replace(b" ", b"  ")
        if rnd.random() < 0.2:
            t += b"'s 12 x"
        texts.append(t)
    n_long = 0
    for t in texts:
        a, b = om.tokenize(t), binding.model_tokenize(path, t)
        assert a == b, t
        n_long += len(a) > 2
    assert n_long > 100
    om.close()


def test_tokenize_buffer_convention(tmp_path):
    """Both tokenize entry points share ONE convention: count >= 0; SS_ERR_BUFFER (-6, nothing written) when n_max is too small -- never a
    value that collides with another SS_ERR_* code; (NULL, 0) is the size query."""
    import ctypes as C
    from speaksense_amd import binding, ggml_io
    path = str(tmp_path / "toy.bin")
    ggml_io.write_model(path, "toy", seed=1)
    L = binding.lib()
    text = b"it's a long way to go"
    need = L.ss_model_tokenize(path.encode(), text, None, 0)
    assert need == len(binding.model_tokenize(path, text)) and need >= 4
    small = np.full(need - 1, -7, np.int32)
    assert L.ss_model_tokenize(path.encode(), text, small.ctypes.data_as(C.c_void_p), len(small)) == -6
    assert (small == -7).all() and b"tokens needed" in L.ss_last_error()
    for k in range(need):           # the old -(needed) convention returned -1 == SS_ERR_ARG when one token was needed, -2..-5 likewise
        buf = np.zeros(max(k, 1), np.int32)
        assert L.ss_model_tokenize(path.encode(), text, buf.ctypes.data_as(C.c_void_p), k) == -6
    assert L.ss_model_tokenize(path.encode(), None, None, 0) == -1


def test_pool_routing_rule():
    """ss_pool_pick: least-loaded engine, ties go round-robin from the cursor (north_star: chunks 'sharded round-robin across the 8 GPUs')."""
    from speaksense_amd import binding
    n = 8
    # idle pool: pure round-robin
    assert [binding.pool_pick([0] * n, c) for c in range(2 * n)] == [c % n for c in range(2 * n)]
    # 64 chunks arriving at once (BASELINE configs[3]: 64 concurrent streams over 8 GPUs): every engine ends up with exactly 8
    load = [0] * n
    for c in range(64):
        k = binding.pool_pick(load, c)
        load[k] += 1
    assert load == [8] * n
    # a busy engine is skipped even when it is next in turn; the least-loaded one wins wherever the cursor stands
    assert binding.pool_pick([5, 0, 0, 0], 0) == 1
    assert binding.pool_pick([3, 2, 1, 2], 0) == 2 and binding.pool_pick([3, 2, 1, 2], 3) == 2
    assert binding.pool_pick([1, 1, 0, 0], 3) == 3 and binding.pool_pick([1, 1, 0, 0], 0) == 2
    assert binding.pool_pick([7], 123) == 0


@pytest.mark.parametrize("ft", ["q5_0", "q5_1", "q8_0"])
def test_quantised_model_files(tmp_path, ft):
    """The `*-q5_0` / `*-q5_1` files of script/download-ggml-model.sh:28-51 (and q8_0): ggml block format written by ggml_io (numpy restatement of
    quantize_row_q*_reference), dequantised by the oracle's C++ loader.  The oracle on the quantised file must equal -- to the bit -- the oracle on
    an f16 file holding numpy's dequantisation of the same blocks (the product loader shares the dequantiser; its GPU test is test_gpu_variants)."""
    from oracle import binding as orc
    from speaksense_amd import synth
    rng = np.random.default_rng(0)
    x = rng.standard_normal(32 * 257).astype(np.float32)
    q = ggml_io.FTYPE_TO_QTYPE[ggml_io.FTYPE_BY_NAME[ft]]
    blocks = ggml_io.quantize_blocks(x, q)
    assert len(blocks) == 257 * ggml_io.Q_BLOCK_BYTES[q]
    y = ggml_io.dequantize_blocks(blocks, q, len(x))
    assert np.abs(x - y).max() < {"q5_0": 0.05, "q5_1": 0.03, "q8_0": 0.005}[ft] * np.abs(x).max()
    assert ggml_io.quantize_blocks(y, q) == blocks or ft != "q8_0"        # q8_0 is idempotent on its own grid
    pq, pf = str(tmp_path / "q.bin"), str(tmp_path / "deq.bin")
    ggml_io.write_model(pq, "toy", seed=1, ftype=ft)
    hp, filt, vocab, t = ggml_io.read_model(pq)
    assert hp.ftype == 2000 + ggml_io.FTYPE_BY_NAME[ft]
    assert os.path.getsize(pq) < 0.62 * 16517 * 1024
    ggml_io.write_model(pf, "toy", seed=1, tensors=t)
    a, b = orc.OracleModel(pq), orc.OracleModel(pf)
    pcm = synth.speech_like(3, 16000 * 10)
    P = orc.default_params(language="en", temperature_inc=0.0)
    ra, rb = a.new_state(orc.MODE_GGML_F16).full(pcm, P), b.new_state(orc.MODE_GGML_F16).full(pcm, P)
    assert list(ra["tokens"]) == list(rb["tokens"]) and np.array_equal(ra["plog"], rb["plog"]) and len(ra["tokens"]) > 0
    a.close(); b.close()


def test_whisper_cpp_comparison_runner_parses_ojf(tmp_path):
    """tools/compare_with_whisper_cpp.py (the box-side runner that pins parity against a real whisper.cpp + weights): its parser of whisper.cpp's
    `-ojf` JSON and its WAV reader, on a hand-made sample in that format."""
    import wave
    sys_path = os.path.join(ROOT, "tools")
    import importlib.util
    spec = importlib.util.spec_from_file_location("cmp_wcpp", os.path.join(sys_path, "compare_with_whisper_cpp.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    j = {"transcription": [
        {"timestamps": {"from": "00:00:00,000", "to": "00:00:02,500"}, "offsets": {"from": 0, "to": 2500}, "text": " hello",
         "tokens": [{"text": "[_BEG_]", "id": 50365, "p": 0.9}, {"text": " hello", "id": 7751, "p": 0.8}, {"text": "[_TT_125]", "id": 50490, "p": 0.7}]},
        {"timestamps": {"from": "00:00:02,500", "to": "00:00:04,000"}, "offsets": {"from": 2500, "to": 4000}, "text": " world",
         "tokens": [{"text": " world", "id": 1002, "p": 0.6}]}]}
    ids, segs = m.parse_whisper_json_full(j)
    assert ids == [50365, 7751, 50490, 1002] and segs == [(0, 250, " hello"), (250, 400, " world")]
    assert m.first_divergence([1, 2, 3], [1, 2, 3]) is None and m.first_divergence([1, 2, 3], [1, 9, 3]) == 1 and m.first_divergence([1, 2], [1, 2, 3]) == 2
    p = str(tmp_path / "a.wav")
    x = (np.sin(np.arange(16000) * 0.05) * 12000).astype("<i2")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(x.tobytes())
    pcm = m.read_wav_16k_mono(p)
    assert pcm.dtype == np.float32 and len(pcm) == 16000 and abs(float(np.abs(pcm).max()) - 12000 / 32768) < 1e-3


def test_decoder_weight_layout_is_a_permutation():
    """The fragment-major layout the decode-step GEMVs read (kernels.h dec_wpack_off, exported as ss_dec_weight_offset): for N % 16 == 0 and
    K % 32 == 0 every (row, column) maps to a distinct offset in [0, N K); the 16 x 32 block one MFMA 16x16x32 consumes is one contiguous
    kilobyte (512 elements) in lane order, 8 consecutive k per lane; 4 consecutive k of a row stay contiguous (the embedding gather and the
    LayerNorm store rely on it)."""
    from speaksense_amd import binding
    L = ctypes.CDLL(binding.LIB_PATH)
    L.ss_dec_weight_offset.restype = ctypes.c_int64
    L.ss_dec_weight_offset.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32]
    off = lambda n, k, K: int(L.ss_dec_weight_offset(n, k, K))
    for N, K in ((16, 32), (48, 128), (32, 1280), (160, 384)):
        seen = np.zeros(N * K, np.int32)
        for n in range(N):
            for k in range(K):
                o = off(n, k, K)
                assert 0 <= o < N * K
                seen[o] += 1
        assert (seen == 1).all(), (N, K)
    K = 1280
    for t, b in ((0, 0), (3, 7), (9, 39)):                       # fragment (tile t, 32-k block b) = elements [base, base + 512)
        base = (t * (K // 32) + b) * 512
        for lane in range(64):
            n, kk = 16 * t + (lane & 15), 32 * b + 8 * (lane >> 4)
            assert [off(n, kk + e, K) for e in range(8)] == list(range(base + 8 * lane, base + 8 * lane + 8))
    assert off(5, 3, 100) == -1 and off(-1, 0, 32) == -1 and off(0, 32, 32) == -1     # K not a multiple of 32 / out of range


def test_session_wait_withholds_results_while_another_ticket_is_outstanding():
    """binding.Session bookkeeping (no GPU): results live on the session and are "valid until its next transcribe/submit" (include/speaksense.h); with
    a second ticket outstanding wait() returns None instead of reading what the engine may be writing."""
    from speaksense_amd import binding

    class FakeLib:
        def ss_wait(self, t): return 0
    s = binding.Session.__new__(binding.Session)
    s.L, s.h, s._pending = FakeLib(), 1, 0
    s.result = lambda: {"tokens": [1, 2, 3]}
    s._pending = 2                       # two tickets submitted
    assert s.wait(object()) is None      # the other one is still outstanding
    assert s.wait(object()) == {"tokens": [1, 2, 3]}


def test_language_table_matches_hf_order():
    """Language i's token is sot + 1 + i: the ORDER of the table is part of the path.  tests/golden/hf_languages.txt is transformers' LANGUAGES (= openai/whisper
    tokenizer.py) in id order; the product's table (whisper_lang_id / whisper_lang_str / whisper_lang_str_full of the whisper.h shim, the Python binding's
    copy) and the oracle's must be that list, full names included (whisper_lang_id accepts them)."""
    import ctypes as C
    import os
    from speaksense_amd import binding
    from oracle import binding as orc
    here = os.path.dirname(os.path.abspath(__file__))
    langs = [l.split(" ", 1) for l in open(os.path.join(here, "golden", "hf_languages.txt")).read().strip().splitlines()]
    assert len(langs) == 100 and langs[0] == ["en", "english"] and langs[99] == ["yue", "cantonese"]
    L = C.CDLL(binding.LIB_PATH)
    L.whisper_lang_id.argtypes = [C.c_char_p]
    L.whisper_lang_str.restype = C.c_char_p
    L.whisper_lang_str_full.restype = C.c_char_p
    assert L.whisper_lang_max_id() == 99
    for i, (code, name) in enumerate(langs):
        assert L.whisper_lang_id(code.encode()) == i, code
        assert L.whisper_lang_id(name.encode()) == i, name
        assert L.whisper_lang_str(i) == code.encode() and L.whisper_lang_str_full(i) == name.encode()
        assert binding.lang_code(i) == code
        assert orc.lib().orc_lang_code_to_id(code.encode()) == i
    assert orc.lib().orc_lang_code_to_id(b"xx") == -1
