"""GPU tests of the Kokoro TTS path (BASELINE.json configs[4]): the synthesiser behind include/skw_tts.h against
oracle/skw_kokoro_oracle.c, the node libkokoro.so through the plugin C ABI, and the voice-agent chain
30 s clip -> libwhisper.so (Whisper-small) -> Transcription -> Text -> libkokoro.so -> 24 kHz frames.
PARITY UNPINNED for the synthesiser's arithmetic (include/skw_tts.h): the checker is this repository's own restatement, not the reference's
(Kokoro-82M inside onnxruntime, absent offline); the node's text front end IS pinned, by the reference's own vectors (tests/test_cpu_kokoro.py)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import kokoro_lib
import minihost
from streamkit_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KOKORO = os.path.join(ROOT, "streamkit_amd", "libkokoro.so")

# tolerances of the float path, relative to the tensor's RMS: every contraction is f64-accumulated on both sides, so what differs is sinf / sin / cos
# between libm and the device's (a few ulps) and the order of f64 partial sums (1e-16)
TOL_STAGE = 1e-4
TOL_WAVE = 1e-3


def _rel(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(np.mean((a - b) ** 2)) / (np.sqrt(np.mean(b * b)) + 1e-30))


TEXTS = ["Hello world. This is a test of the synthesiser, 1 2 3!", "Short one.", "你好。 Mixed: café, naïve?",
         "A considerably longer sentence, with commas; semicolons: and colons - so that the token count moves the style row and the frame count grows well past a few hundred frames."]


@pytest.mark.parametrize("size", ["micro", "small"])
def test_synthesiser_matches_oracle_stage_by_stage(size):
    d = kokoro_lib.synth_kokoro_dir(size)
    tts = kokoro_lib.Tts(d); orc = kokoro_lib.OracleTts(d)
    assert tts.L.skw_tts_sample_rate(tts.h) == 24000 and tts.L.skw_tts_num_speakers(tts.h) == 103
    for k, text in enumerate(TEXTS):
        sid, speed = (50, 1.0) if k % 2 == 0 else (7, 1.25)
        assert tts.tokenize(text).tolist() == kokoro_lib.tokenize(text, d)                       # the product's tokeniser == the Python restatement
        y, rate = tts.generate(text, sid, speed)
        r = orc.synth(text, sid, speed)
        assert rate == 24000
        assert np.array_equal(tts.tap(0).astype(np.int32), r["dur"]), (size, k)                  # durations: integers, exact
        F = int(r["dur"].sum())
        assert y.size == 600 * F - 5 == r["y"].size                                              # 600 samples per frame, centre-trimmed
        e = {"f0": _rel(tts.tap(1), r["f0"]), "energy": _rel(tts.tap(2), r["en"]), "decoder": _rel(tts.tap(3), r["z"].ravel()), "spec": _rel(tts.tap(4), r["o"].ravel()), "wave": _rel(y, r["y"])}
        print("kokoro %s text %d: %d tokens, %d frames, %.2f s of audio in %.2f ms on the GPU; rel rms err %s" % (size, k, r["ids"].size, F, y.size / 24000.0, tts.last_ms(), {a: "%.2g" % b for a, b in e.items()}))
        assert all(v < TOL_STAGE for n, v in e.items() if n != "wave") and e["wave"] < TOL_WAVE, e
        assert np.isfinite(y).all() and 1e-3 < float(np.sqrt((y ** 2).mean())) < 0.5
    with pytest.raises(RuntimeError, match="speaker id"):
        tts.generate("Hello.", 1000, 1.0)
    with pytest.raises(RuntimeError, match="speed must be positive"):
        tts.generate("Hello.", 0, 0.0)
    with pytest.raises(RuntimeError, match="no symbol of the text"):
        tts.generate("☃☃", 0, 1.0)
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
        assert y.size == r["y"].size and _rel(y, r["y"]) < TOL_WAVE, (i, sent)
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
    assert y.size == r["y"].size and _rel(y, r["y"]) < TOL_WAVE
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
            assert y.size == r["y"].size and _rel(y, r["y"]) < TOL_WAVE, sent[:40]
    done = [t[1] for t in tts.telemetry() if t[0] == "tts.done"]
    assert len(done) == len(frames) and sum(x["audio_samples"] for x in done) == total
    print("configs[4]: 30 s clip -> Whisper-small (f16_mfma) -> %d sentence(s), %d chars -> Kokoro-shaped synthesiser -> %.1f s of 24 kHz audio; chain wall time %.2f s (packet feeding included), TTS latency %s ms"
          % (len(want), sum(len(s) for s in want), total / 24000.0, wall, [x["latency_ms"] for x in done]))
    stt.destroy(); tts.destroy()
