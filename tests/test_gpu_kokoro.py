"""GPU tests of the Kokoro TTS path (BASELINE.json configs[4]): the synthesiser behind include/skw_tts.h against
oracle/skw_kokoro_oracle.cpp (the same network, include/skw_kokoro_net.h, every operator a CPU loop), the node libkokoro.so through the plugin C ABI, and the voice-agent chain
30 s clip -> libwhisper.so (Whisper-small) -> Transcription -> Text -> libkokoro.so -> 24 kHz frames.
PARITY UNPINNED for the synthesiser's arithmetic (include/skw_tts.h): the checker is this repository's own restatement, not the reference's
(Kokoro-82M inside onnxruntime, absent offline); the node's text front end IS pinned, by the reference's own vectors (tests/test_cpu_kokoro.py)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import kokoro_lib
from streamkit_amd import minihost
from streamkit_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KOKORO = os.path.join(ROOT, "streamkit_amd", "libkokoro.so")

# Every contraction is the same k-ascending f32 chain on both sides (MFMA 16x16x4 f32 on the GPU, fmaf on the CPU), exp is skw_expf on both and the f64 statistics are summed in
# the contract's order (include/skw_kokoro_net.h), so everything up to the decoder's output — ALBERT, durations, F0 / energy curves, text encoder, decoder — must be BIT-IDENTICAL.
# From the generator on, Snake's sinf and the source's sin / the STFT's atan2 / the inverse STFT's sin, cos are the platform's, a few ulps apart: tolerances relative to the RMS.
# One discontinuity is the architecture's own: the source's STFT PHASE is a network input, and a near-empty bin whose phase sits at +-pi can wrap the other way on an ulp (measured
# once, round 4, when the F0 curves still differed by 1e-6: three wraps in 650 000 bins, each disturbing ~600 spectrum rows).  Rows downstream of such a wrap are left out.
TOL_GEN = 2e-5
TOL_WAVE = 5e-5


def _wrap_mask(har_gpu, har_cpu, span=1500):
    """rows of the spectrum that no +-pi wrap of the source's phase can have reached (the generator's receptive field is under `span` rows either side)"""
    dp = np.abs(har_gpu[:, 11:] - har_cpu[:, 11:])
    rows = np.unique(np.nonzero(dp > 3.0)[0])
    keep = np.ones(har_cpu.shape[0], bool)
    for r in rows:
        keep[max(0, r - span):r + span] = False
    return keep, rows.size


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b * b)) + 1e-30))


def _wave_ok(y, ref):
    """the node-level check: same length, and at least 80 % of the 0.1 s blocks within TOL_WAVE of the checker's (a phase wrap, see above, disturbs at most a few blocks)"""
    if y.size != ref.size:
        return False
    n = (y.size // 2400) * 2400
    if n == 0:
        return _rel(y, ref) < 2e-3
    rms = np.sqrt(np.mean(ref.astype(np.float64) ** 2)) + 1e-30
    eb = np.sqrt(np.mean((y[:n].astype(np.float64) - ref[:n]).reshape(-1, 2400) ** 2, axis=1)) / rms
    return float(np.mean(eb < TOL_WAVE)) >= 0.8


TEXTS = ["Hello world. This is a test of the synthesiser, 1 2 3!", "Short one.", "你好。 Mixed: café, naïve?",
         "A considerably longer sentence, with commas; semicolons: and colons - so that the token count moves the style row and the frame count grows well past a few hundred frames."]


@pytest.mark.parametrize("size", ["micro", "small"])
def test_synthesiser_matches_oracle_stage_by_stage(size):
    d = kokoro_lib.synth_kokoro_dir(size)
    tts = kokoro_lib.Tts(d); orc = kokoro_lib.OracleTts(d)
    assert tts.L.skw_tts_sample_rate(tts.h) == 24000 and tts.L.skw_tts_num_speakers(tts.h) == 103
    tts.taps(True)
    for k, text in enumerate(TEXTS):
        sid, speed = (50, 1.0) if k % 2 == 0 else (7, 1.25)
        assert tts.tokenize(text).tolist() == kokoro_lib.tokenize(text, d)                       # the product's tokeniser == the Python restatement
        y, rate = tts.generate(text, sid, speed)
        r = orc.synth(text, sid, speed)
        assert rate == 24000
        assert np.array_equal(tts.tap(0).astype(np.int32), r["dur"]), (size, k)                  # durations: integers, exact
        F = int(r["dur"].sum())
        assert y.size == 600 * F == r["y"].size                                                  # 600 samples per predicted frame (2 x 10 x 6 x hop 5)
        for what, name in ((5, "bert"), (6, "d_en"), (7, "t_en"), (1, "f0"), (2, "en"), (3, "dec")):
            assert np.array_equal(tts.tap(what).view(np.uint32), r[name].ravel().view(np.uint32)), (size, k, name, _rel(tts.tap(what), r[name].ravel()))
        keep, wraps = _wrap_mask(tts.tap(8).reshape(-1, 22), r["har"])
        assert wraps <= 3 and keep.mean() > 0.5, (wraps, keep.mean())
        keep_y = np.repeat(keep[:-1], 5)                                                          # spectrum row p covers samples 5 p .. 5 p + 4
        e = {"source": _rel(tts.tap(8).reshape(-1, 22)[keep, :11], r["har"][keep, :11]), "spec": _rel(tts.tap(4).reshape(-1, 22)[keep], r["post"][keep]), "wave": _rel(y[keep_y], r["y"][keep_y])}
        print("kokoro %s text %d: %d tokens, %d frames, %.2f s of audio in %.2f ms on the GPU; bit-identical through the decoder; %d phase wrap(s); rel rms err %s"
              % (size, k, r["ids"].size, F, y.size / 24000.0, tts.last_ms(), wraps, {a: "%.2g" % b for a, b in e.items()}))
        assert e["source"] < TOL_GEN and e["spec"] < TOL_GEN and e["wave"] < TOL_WAVE, e
        assert np.isfinite(y).all() and 1e-3 < float(np.sqrt((y ** 2).mean())) < 0.5
    with pytest.raises(RuntimeError, match="speaker id"):
        tts.generate("Hello.", 1000, 1.0)
    with pytest.raises(RuntimeError, match="speed must be positive"):
        tts.generate("Hello.", 0, 0.0)
    with pytest.raises(RuntimeError, match="no symbol of the text"):
        tts.generate("☃☃", 0, 1.0)
    tts.taps(False)
    y2, _ = tts.generate(TEXTS[1], 7, 1.25)                                                      # taps off: the same audio, and the previous taps stay readable
    assert np.array_equal(y2.view(np.uint32), tts.generate(TEXTS[1], 7, 1.25)[0].view(np.uint32))      # deterministic call to call
    tts.close()


@pytest.mark.parametrize("size", ["micro", "small"])
def test_kernel_variants_are_bit_identical(size):
    """k_tts_conv (one gather per lane) and k_tts_conv_t (128-step slabs through LDS), k_tts_lstm (one workgroup per direction) and k_tts_lstm_mw (H / 32 workgroups per direction,
    weights resident in LDS, h exchanged through memory) evaluate the same k-ascending chains: forced one way and the other, every tap and the waveform must agree bit for bit
    (the automatic choice mixes them by launch size, so the checker comparison above covers the mix)."""
    d = kokoro_lib.synth_kokoro_dir(size)
    tts = kokoro_lib.Tts(d); tts.taps(True)
    got = {}
    try:
        for mode in (1, 2):
            tts.L.skw_tts_debug_conv_mode(mode); tts.L.skw_tts_debug_lstm_mode(mode)
            y, _ = tts.generate(TEXTS[3], 7, 1.25)
            got[mode] = [y] + [tts.tap(k) for k in range(9)]
    finally:
        tts.L.skw_tts_debug_conv_mode(0); tts.L.skw_tts_debug_lstm_mode(0)
    for a, b in zip(got[1], got[2]):
        assert a.size == b.size and np.array_equal(a.view(np.uint32), b.view(np.uint32))
    tts.close()


def test_synthesiser_edges_shortest_longest_and_too_long():
    """The ends of the input range, against the checker: the shortest utterance the tokeniser lets through (one symbol between the pads: 3 tokens, every launch a fraction of a
    tile), the longest (510 tokens incl. pads: the position table's and voices.bin's last row), and a call whose predicted frame count passes the engine's cap — refused with the
    same message by both, before any waveform work."""
    d = kokoro_lib.synth_kokoro_dir("micro")
    tts = kokoro_lib.Tts(d); orc = kokoro_lib.OracleTts(d); tts.taps(True)
    rng = np.random.default_rng(11)
    for ids, speed in (([0, 20, 0], 1.0), (np.concatenate([[0], rng.integers(1, 70, 508), [0]]), 2.0)):
        ids = np.asarray(ids, np.int32)
        y, _ = tts.generate(None, 3, speed, ids=ids)
        r = orc.synth(None, 3, speed, ids=ids)
        assert np.array_equal(tts.tap(0).astype(np.int32), r["dur"]) and y.size == r["y"].size == 600 * int(r["dur"].sum())
        for what, name in ((5, "bert"), (1, "f0"), (3, "dec")):
            assert np.array_equal(tts.tap(what).view(np.uint32), r[name].ravel().view(np.uint32)), (ids.size, name)
        keep, wraps = _wrap_mask(tts.tap(8).reshape(-1, 22), r["har"])
        keep_y = np.repeat(keep[:-1], 5)
        assert wraps <= 3 and _rel(y[keep_y], r["y"][keep_y]) < TOL_WAVE, (ids.size, wraps)
        print("kokoro micro edge: %d tokens -> %d frames, %.2f s of audio in %.2f ms" % (ids.size, y.size // 600, y.size / 24000.0, tts.last_ms()))
    long_ids = np.concatenate([[0], rng.integers(1, 70, 508), [0]]).astype(np.int32)
    with pytest.raises(RuntimeError, match=r"Generated audio too long \(\d+ frames; at most 3000\)"):
        tts.generate(None, 3, 0.05, ids=long_ids)                                                    # durations are divided by the speed: 20 x as many frames
    with pytest.raises(RuntimeError, match=r"Generated audio too long \(\d+ frames; at most 3000\)"):
        orc.synth(None, 3, 0.05, ids=long_ids)
    with pytest.raises(RuntimeError, match="token id outside the embedding table"):
        tts.generate(None, 3, 1.0, ids=np.array([0, 5000, 0], np.int32))
    y, _ = tts.generate(None, 3, 1.0, ids=np.array([0, 20, 0], np.int32))                             # the engine is usable after a refused call
    assert y.size > 0 and np.isfinite(y).all()
    tts.close()


def test_kokoro_82m_geometry_speed():
    """The synthesiser at Kokoro-82M's real widths (tools/make_synth_kokoro.py --size kokoro82m: 12 ALBERT passes at 768, 512-wide predictor / text encoder, 1024-wide decoder,
    512 -> 256 -> 128 generator; seeded weights): ~30 s of speech in one call, timed with GPU events.  Too large for the CPU checker within a test run — the arithmetic is the
    same code the micro / small sizes check against it; here the output is only checked for shape, finiteness and determinism."""
    d = kokoro_lib.synth_kokoro_dir("kokoro82m")
    tts = kokoro_lib.Tts(d)
    rng = np.random.default_rng(5)
    ids = np.concatenate([[0], rng.integers(1, 60, 300), [0]]).astype(np.int32)
    y, _ = tts.generate(None, 50, 1.0, ids=ids)
    frames = y.size // 600
    speed = float(np.clip(frames / 1200.0, 0.3, 3.0))                                          # aim at ~1200 frames = 30 s
    best = 1e9
    for _ in range(3):
        y, rate = tts.generate(None, 50, speed, ids=ids); best = min(best, tts.last_ms())
    secs = y.size / 24000.0
    y2, _ = tts.generate(None, 50, speed, ids=ids)
    assert np.isfinite(y).all() and y.size % 600 == 0 and 10.0 < secs <= 75.0 and np.array_equal(y.view(np.uint32), y2.view(np.uint32))
    print("kokoro82m geometry: %d tokens -> %d frames, %.1f s of 24 kHz audio in %.1f ms on the GPU (best of 3): %.0fx real time" % (ids.size, y.size // 600, secs, best, secs * 1000.0 / best))
    assert secs * 1000.0 / best > 20.0
    tts.close()


def test_kokoro_node_through_the_plugin_abi():
    """process (Text and Binary), sentence splitting, one 24 kHz mono f32 frame per sentence, tts.start / tts.done telemetry, update_params, flush —
    kokoro_node.rs:444-652 through the C ABI, audio checked against the oracle's synthesiser."""
    d = kokoro_lib.synth_kokoro_dir("micro")
    p = minihost.Plugin(KOKORO); L = minihost.lib()
    L.mh_process_binary.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]; L.mh_output_audio_format.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.POINTER(C.c_uint16)]
    orc = kokoro_lib.OracleTts(d)
    node = p.create_node({"model_dir": d, "speaker_id": 9, "emit_telemetry": True, "telemetry_preview_chars": 12, "min_sentence_length": 10})
    texts = ["Hello there, how are you? I am fine. thanks", "tiny", "<b>and</b>   more \U0001F600 text"]
    assert node.process_text(texts[0]) == 0, node.last_error()
    assert len(node.outputs()) == 1                                                             # "thanks." (7 bytes < min_sentence_length) stays buffered
    assert node.process_text(texts[1]) == 0 and len(node.outputs()) == 2                        # ... until "tiny." joins it: "thanks.tiny."
    b = texts[2].encode(); assert L.mh_process_binary(node.h, b, len(b)) == 0, node.last_error()  # Binary packets are UTF-8 text too (kokoro_node.rs:449-452)
    want = kokoro_lib.node_sentences(texts, 10, flush=False)
    outs = node.outputs()
    assert len(outs) == len(want) == 3 and want == ["Hello there, how are you? I am fine.", "thanks.tiny.", "bandb more text."]      # '<', '>', '/' and the emoji are dropped, not spaced
    for i, (o, sent) in enumerate(zip(outs, want)):
        rate, ch = C.c_uint32(), C.c_uint16(); L.mh_output_audio_format(node.h, i, C.byref(rate), C.byref(ch))
        assert o[0] == "out" and o[1] == 0 and (rate.value, ch.value) == (24000, 1)            # RawAudio{24000, 1, F32} on pin "out"
        y = np.frombuffer(o[2], np.float32); r = orc.synth(sent, 9, 1.0)
        assert _wave_ok(y, r["y"]), (i, sent)
    tel = node.telemetry()
    assert [t[0] for t in tel] == ["tts.start", "tts.done"] * 3
    s0, d0 = tel[0][1], tel[1][1]
    assert s0 == {"execution_provider": "cpu", "speaker_id": 9, "speed": 1.0, "text_length": len(want[0].encode()), "text_preview": "Hello there,..."}
    n0 = np.frombuffer(outs[0][2], np.float32).size
    assert {k: v for k, v in d0.items() if k != "latency_ms"} == dict(s0, audio_samples=n0, audio_duration_ms=(n0 * 1000 + 12000) // 24000)
    assert isinstance(d0["latency_ms"], int) and 0 <= d0["latency_ms"] < 5000
    assert list(d0.keys()) == sorted(d0.keys())                                                # serde_json::json! without preserve_order: keys in alphabetical order
    # update_params: the whole config is parsed (model_dir required), only speaker_id and speed are taken over (kokoro_node.rs:494-506)
    assert node.update_params({"speaker_id": 3}) == -1 and "missing field `model_dir`" in node.last_error()
    assert node.update_params({"model_dir": "/does/not/matter", "speaker_id": 3, "speed": 1.5, "min_sentence_length": 1}) == 0
    assert node.process_text("x") == 0 and len(node.outputs()) == 3                            # "x." still shorter than the ORIGINAL min_sentence_length
    assert node.flush() == 0, node.last_error()                                                 # flush speaks what is buffered (kokoro_node.rs:508-533)
    outs = node.outputs()
    assert len(outs) == 4
    r = orc.synth("x.", 3, 1.5); y = np.frombuffer(outs[3][2], np.float32)
    assert _wave_ok(y, r["y"])
    assert node.telemetry()[-1][1]["speaker_id"] == 3 and node.telemetry()[-1][1]["speed"] == 1.5
    assert node.flush() == 0 and len(node.outputs()) == 4                                       # nothing left
    # errors (kokoro_node.rs:447-455)
    assert node.process_audio(np.zeros(16, np.float32)) != 0 and node.last_error() == "Only accepts Text or Binary packets"
    node.destroy()
    n2 = p.create_node({"model_dir": d})                                                        # the engine is cached per (model_dir, threads, provider)
    assert any("CACHE HIT" in l for l in n2.logs())
    bad = b"\xff\xfe"; assert L.mh_process_binary(n2.h, bad, 2) != 0 and n2.last_error().startswith("Failed to decode binary data as UTF-8")
    n2.destroy()


def test_config5_whisper_small_to_kokoro_voice_agent_chain(small_model_path):
    """BASELINE.json configs[4]: a 30 s clip -> libwhisper.so (Whisper-small; the reference pipeline's STT node) -> the Transcription -> Text step of
    samples/pipelines/dynamic/voice-agent-openai.yaml:86-95 -> libkokoro.so -> 24 kHz mono frames, on one MI355X; every frame equals the oracle's
    synthesis of the sentence the reference's front end would have cut."""
    d = kokoro_lib.synth_kokoro_dir("small")
    wp = minihost.Plugin(); kp = minihost.Plugin(KOKORO); L = minihost.lib()
    L.mh_forward_transcription_as_text.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    stt = wp.create_node({"model_path": small_model_path, "vad_mode": "always", "flush_tail": True, "precision": "f16_mfma"})
    tts = kp.create_node({"model_dir": d, "emit_telemetry": True})
    pcm = synth.clip(17, 480000)
    import time
    t0 = time.perf_counter()
    for i in range(0, pcm.size, 960):
        assert stt.process_audio(pcm[i:i + 960]) == 0, stt.last_error()
    assert stt.flush() == 0
    souts = stt.outputs()
    assert len(souts) >= 1 and all(o[1] == 3 for o in souts)
    texts = []
    for i, o in enumerate(souts):
        rc = L.mh_forward_transcription_as_text(stt.h, i, tts.h)
        assert rc in (0, 1), tts.last_error()
        t = json.loads(o[2].decode())["text"].strip()
        if t:
            texts.append(t)
    assert tts.flush() == 0, tts.last_error()
    wall = time.perf_counter() - t0
    frames = tts.outputs()
    want = kokoro_lib.node_sentences(texts, 10, flush=True)
    assert len(frames) == len(want) >= 1
    orc = kokoro_lib.OracleTts(d); total = 0
    for o, sent in zip(frames, want):
        y = np.frombuffer(o[2], np.float32); total += y.size
        assert o[1] == 0 and np.isfinite(y).all()
        if len(sent) <= 400:                                                                    # (the oracle's direct convolutions take a few seconds per long sentence)
            r = orc.synth(sent, 50, 1.0)
            assert _wave_ok(y, r["y"]), sent[:40]
    done = [t[1] for t in tts.telemetry() if t[0] == "tts.done"]
    assert len(done) == len(frames) and sum(x["audio_samples"] for x in done) == total
    print("configs[4]: 30 s clip -> Whisper-small (f16_mfma) -> %d sentence(s), %d chars -> Kokoro-shaped synthesiser -> %.1f s of 24 kHz audio; chain wall time %.2f s (packet feeding included), TTS latency %s ms"
          % (len(want), sum(len(s) for s in want), total / 24000.0, wall, [x["latency_ms"] for x in done]))
    stt.destroy(); tts.destroy()


# ------------------------------------------------------------------ against the INDEPENDENT checker (tests/kokoro_torch_ref.py)
def test_synthesiser_matches_the_torch_fixture_stage_by_stage():
    """libskw_tts.so against tests/golden/kokoro_torch_micro.json — known answers of a torch.nn restatement of the published Kokoro modules (ALBERT: transformers' AlbertModel)
    that shares no code with include/skw_kokoro_net.h, on the seeded micro model (loaded strict=True there): durations exactly; ALBERT, bert_encoder, text encoder, the F0 / N
    curves and the decoder's output at 96 seeded positions each, within 1e-4 of the stage's largest magnitude (torch sums in its own order: ~3e-6 measured against the CPU checker);
    the waveform's length exactly and its RMS within 5 % (free-running, everything after the curves depends on the integral of F0: the next test hands the curves over).
    PARITY UNPINNED still — no trained model, no sherpa-onnx output — but the WIRING is no longer checked only against itself."""
    fx = json.load(open(os.path.join(ROOT, "tests", "golden", "kokoro_torch_micro.json")))
    d = kokoro_lib.synth_kokoro_dir("micro")
    tts = kokoro_lib.Tts(d); tts.taps(True)
    worst = {}
    for c in fx["cases"]:
        ids = np.asarray(c["ids"], np.int32)
        assert tts.tokenize(c["text"]).tolist() == c["ids"]
        y, rate = tts.generate(None, c["sid"], c["speed"], ids=ids)
        assert rate == 24000 and tts.tap(0).astype(np.int64).tolist() == c["durations"]
        for what, name in ((5, "bert"), (6, "d_en"), (7, "t_en"), (1, "f0"), (2, "en"), (3, "dec")):
            st = c["stages"][name]; a = tts.tap(what)
            assert a.size == int(np.prod(st["shape"])), (name, a.size, st["shape"])
            e = float(np.abs(a[st["positions"]] - np.array(st["values"])).max() / st["max_abs"])
            worst[name] = max(worst.get(name, 0.0), e)
            assert e < 1e-4, (c["text"][:20], name, e)
        assert y.size == c["wave"]["n"] == 600 * sum(c["durations"])
        assert abs(float(np.sqrt(np.mean(y.astype(np.float64) ** 2))) / c["wave"]["rms"] - 1.0) < 0.05
    print("libskw_tts.so vs the torch fixture (micro, %d utterances): worst relative error per stage %s" % (len(fx["cases"]), {k: "%.1e" % v for k, v in worst.items()}))
    tts.close()


@pytest.mark.parametrize("size", ["micro", "small"])
def test_generator_matches_the_torch_restatement_given_the_same_curves(size):
    """The stages after the curves, live: tests/kokoro_torch_ref.py (torch, on this box's CPU) is handed the F0 / N curves libskw_tts.so predicted — the harmonic source integrates F0
    into a phase, so implementations whose curves differ by 1e-6 differ by 1e-2 after it — and must reproduce the GPU's decoder output, the harmonic source's STFT (magnitudes;
    phases where a bin has energy: the published SineGen phase law, which rounds 3-4 had not followed) and, handed the GPU's source spectrum as well (a phase at +-pi comes out on
    either side on an ulp), the post-convolution spectrum and the waveform."""
    torch = pytest.importorskip("torch"); pytest.importorskip("transformers")
    import kokoro_torch_ref as ktr
    d = kokoro_lib.synth_kokoro_dir(size)
    m = ktr.build_from_tensors(kokoro_lib.load_model_tensors(d)); voices = kokoro_lib.load_voices(d)
    tts = kokoro_lib.Tts(d); tts.taps(True)
    for text, sid, speed in ((TEXTS[0], 50, 1.0), (TEXTS[1], 7, 1.25)):
        ids = tts.tokenize(text)
        y, _ = tts.generate(None, sid, speed, ids=ids)
        f0, en, dec, post, har = tts.tap(1), tts.tap(2), tts.tap(3), tts.tap(4).reshape(-1, 22), tts.tap(8).reshape(-1, 22)
        ref_s = torch.from_numpy(voices[sid, kokoro_lib.style_row(ids.size)].copy()).unsqueeze(0)
        tid = torch.from_numpy(ids.astype(np.int64)).unsqueeze(0)
        _, dur, t1 = m.forward_with_tokens(tid, ref_s, speed, None, kokoro_lib.source_noise, curves=(f0, en))
        audio, _, t2 = m.forward_with_tokens(tid, ref_s, speed, None, kokoro_lib.source_noise, curves=(f0, en), har=har)
        assert dur.tolist() == tts.tap(0).astype(np.int64).tolist()
        rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / np.abs(b).max())
        e_dec = rel(t1["dec"].numpy().reshape(-1), dec)
        h = t1["har"].numpy()
        e_mag = rel(h[:, :11], har[:, :11])
        ph = np.abs(h[:, 11:] - har[:, 11:]); ph = np.minimum(ph, 2 * np.pi - ph); live = har[:, :11] > 1e-3 * har[:, :11].max()
        e_post, e_wave = rel(t2["post"].numpy(), post), rel(audio.numpy(), y)
        print("kokoro %s, %d tokens: torch given the GPU's curves: decoder %.1e, source magnitudes %.1e, phases %.1e rad; given its source spectrum too: spectrum %.1e, waveform %.1e"
              % (size, ids.size, e_dec, e_mag, float(ph[live].max()), e_post, e_wave))
        assert e_dec < 1e-4 and e_mag < 1e-4 and ph[live].max() < 2e-3 and e_post < 1e-4 and e_wave < 1e-3
    tts.close()


def test_damaged_model_directories_are_refused_with_a_message(tmp_path):
    """kokoro_node.rs:705-745 checks that the files exist and hands their paths on; what is IN them is the library's to survive: a truncated or corrupted model.onnx / voices.bin /
    tokens.txt / lexicon ends in skw_tts_create's error string (the node: "Failed to create TTS engine"), or in an engine that synthesises, never in a crash of the host."""
    import shutil
    src = kokoro_lib.synth_kokoro_dir("micro")
    rng = np.random.default_rng(11)
    refused = ran = 0
    files = ["model.onnx", "voices.bin", "tokens.txt", "lexicon-us-en.txt"]
    for case in range(40):
        d = str(tmp_path / ("k%d" % case)); shutil.copytree(src, d)
        name = files[case % 4]; p = os.path.join(d, name); blob = bytearray(open(p, "rb").read())
        kind = (case // 4) % 3
        if kind == 0:
            blob = blob[:int(rng.integers(0, max(1, len(blob))))]                                     # truncated
        elif kind == 1:
            for _ in range(int(rng.integers(1, 5))):
                blob[int(rng.integers(0, min(len(blob), 4096)))] = int(rng.integers(0, 256))          # corrupted near the start (structure)
        else:
            blob = bytearray(rng.integers(0, 256, int(rng.integers(1, 5000)), dtype=np.uint8).tobytes())  # not this kind of file at all
        open(p, "wb").write(bytes(blob))
        try:
            t = kokoro_lib.Tts(d)
        except RuntimeError as e:
            assert len(str(e)) > 8, (name, kind)
            refused += 1
            continue
        try:
            t.generate("hello there", 0, 1.0)
        except RuntimeError as e:
            assert len(str(e)) > 8
        t.close(); ran += 1
    assert refused >= 12 and refused + ran == 40, (refused, ran)
