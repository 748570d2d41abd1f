"""GPU tests of the drop-in boundary: libwhisper.so driven through the StreamKit native-plugin C ABI by the C++ mini-host,
replaying the reference host's call sequence (wrapper.rs) and the config-1 node chain around it."""
import json
import os
import threading

import numpy as np
import pytest

from streamkit_amd import minihost
import oracle_lib
from oracle_lib import OracleModel
from streamkit_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def plugin():
    return minihost.Plugin()


def _expected_transcription(om, pcm_segment, start_ms, language="en"):
    """What lib.rs:648-695 builds from whisper.cpp's segments, computed from the oracle's full()."""
    po = om.default_params(); po.suppress_nst = 1
    r = om.full(pcm_segment, po)
    segs = []
    for s in r["segments"]:
        text = s["text"].decode().strip()
        if text:
            segs.append({"text": text, "start_time_ms": start_ms + s["t0"] * 10, "end_time_ms": start_ms + s["t1"] * 10, "confidence": None})
    return {"text": " ".join(s["text"] for s in segs), "segments": segs, "language": language, "metadata": None} if segs else None


def _feed(node, pcm, packet=960):
    for i in range(0, pcm.size, packet):
        assert node.process_audio(pcm[i:i + packet]) == 0, node.last_error()


def test_oneshot_forced_cut_matches_oracle(plugin, tiny_model_path):
    om = OracleModel(tiny_model_path)
    node = plugin.create_node({"model_path": tiny_model_path, "vad_mode": "always", "emit_vad_events": True})
    pcm = np.concatenate([synth.clip(0), synth.clip(1, 16000 * 3)])          # 33 s: one forced cut at 939 frames, tail dropped
    _feed(node, pcm)
    assert node.flush() == 0
    outs = node.outputs()
    assert len(outs) == 1 and outs[0][0] == "out" and outs[0][1] == 3       # one Transcription packet on pin "out"
    got = json.loads(outs[0][2].decode())
    assert list(got.keys()) == ["text", "segments", "language", "metadata"]  # serde field order (crates/core/src/types.rs:166-175)
    assert got == _expected_transcription(om, pcm[:939 * 512], 0)
    tel = node.telemetry()
    assert [t[0] for t in tel] == ["vad.speech_start", "vad.speech_end", "vad.speech_start"]
    assert tel[0][1] == {"segment_id": "seg-0-1", "speech_probability": 1.0, "start_time_ms": 0, "threshold": 0.5}
    assert tel[1][1] == {"duration_ms": 30048, "end_time_ms": 30048, "reason": "max_duration", "segment_id": "seg-0-1", "silence_duration_ms": None, "start_time_ms": 0}
    assert tel[2][1]["segment_id"] == "seg-30048-2"
    node.destroy()


def test_energy_vad_silence_cut_and_absolute_timestamps(plugin, tiny_model_path):
    om = OracleModel(tiny_model_path)
    node = plugin.create_node({"model_path": tiny_model_path, "vad_mode": "energy", "min_silence_duration_ms": 320})
    speech = synth.clip(2, 512 * 150)                                       # 150 frames of "speech" (rms >> 0.01)
    pcm = np.concatenate([np.zeros(512 * 20, np.float32), speech, np.zeros(512 * 30, np.float32)])
    _feed(node, pcm, packet=1000)                                           # packet size not aligned with the 512 framing
    outs = node.outputs()
    assert len(outs) == 1
    got = json.loads(outs[0][2].decode())
    # which frames pass the energy gate is decided by the product's VAD; the oracle then sees exactly those samples
    frames = speech.reshape(150, 512)
    rms = np.sqrt((frames.astype(np.float32) ** 2).sum(axis=1) / 512.0)
    keep = (rms / (rms + 0.01)) >= 0.5
    assert keep.all()
    assert got == _expected_transcription(om, speech, 20 * 32)
    assert node.outputs()[0][2] == outs[0][2]
    node.destroy()


def test_error_paths_match_reference_messages(plugin, tiny_model_path):
    node = plugin.create_node({"model_path": tiny_model_path, "vad_mode": "always"})
    assert node.process_text("hello") == -1 and node.last_error() == "Whisper plugin only accepts audio packets"          # lib.rs:492
    assert node.process_null() == -1 and node.last_error() == "(null message)"                                            # sdk lib.rs:731-733
    n2 = plugin.create_node({"model_path": tiny_model_path, "vad_mode": "always"})
    assert n2.process_audio(np.zeros(960, np.float32), 48000, 1) == -1
    assert n2.last_error() == "Whisper requires 16kHz audio, got 48000Hz. Please add an audio_resample node upstream."    # lib.rs:185-187
    n3 = plugin.create_node({"model_path": tiny_model_path, "vad_mode": "always"})
    assert n3.process_audio(np.zeros(960, np.float32), 16000, 2) == -1
    assert n3.last_error() == "Whisper requires mono audio, got 2 channels. Please add an audio_resample node upstream."  # lib.rs:191-193
    with pytest.raises(RuntimeError, match="Failed to load Whisper model"):
        plugin.create_node({"model_path": "/nonexistent/ggml-small.bin"})
    assert node.update_params({"model_path": tiny_model_path, "vad_mode": "always", "min_silence_duration_ms": 100}) == 0
    assert node.update_params({"model_path": "/nonexistent.bin"}) == -1 and "Failed to reload Whisper model" in node.last_error()
    for n in (node, n2, n3):
        n.destroy()


def test_flush_drops_tail_by_default_and_flush_tail_is_additive(plugin, tiny_model_path):
    om = OracleModel(tiny_model_path)
    pcm = synth.clip(4, 16000 * 6)
    a = plugin.create_node({"model_path": tiny_model_path, "vad_mode": "always"})
    _feed(a, pcm); assert a.flush() == 0 and a.outputs() == []               # reference behaviour: no flush override
    b = plugin.create_node({"model_path": tiny_model_path, "vad_mode": "always", "flush_tail": True})
    _feed(b, pcm); assert b.flush() == 0
    outs = b.outputs()
    assert len(outs) == 1 and json.loads(outs[0][2].decode()) == _expected_transcription(om, pcm, 0)   # incl. the < 512 samples of the unfinished VAD frame
    a.destroy(); b.destroy()


def test_concurrent_instances_share_the_model_and_batch(plugin, tiny_model_path):
    """many instances on different OS threads (wrapper.rs:398) -> the per-GPU scheduler batches them; results must equal the serial ones"""
    clips = [np.concatenate([synth.clip(c), np.zeros(1024, np.float32)]) for c in range(6)]
    def run(params):
        nodes = [plugin.create_node(params) for _ in clips]
        ths = [threading.Thread(target=_feed, args=(n, c, 1920)) for n, c in zip(nodes, clips)]
        [t.start() for t in ths]; [t.join() for t in ths]
        outs = [[o[2] for o in n.outputs()] for n in nodes]
        [n.destroy() for n in nodes]
        return outs
    par = run({"model_path": tiny_model_path, "vad_mode": "always", "batch_window_ms": 50})
    ser = run({"model_path": tiny_model_path, "vad_mode": "always", "batch_window_ms": 0, "max_batch": 1})
    assert par == ser and all(len(o) == 1 for o in par)


def test_config1_node_chain(plugin, tiny_model_path):
    """BASELINE configs[0]: 16 kHz WAV -> 1920-sample demux frames -> audio::resampler (pass-through, 960-sample packets)
    -> whisper -> core::json_serialize NDJSON (externally tagged packet)."""
    om = OracleModel(tiny_model_path)
    pcm = np.concatenate([synth.clip(8), synth.clip(9, 16000 * 2)])
    rs = minihost.Resampler(16000, 960, 960)
    node = plugin.create_node({"model_path": tiny_model_path, "vad_mode": "always", "language": "en"})
    for i in range(0, pcm.size, 1920):
        rs.push(pcm[i:i + 1920], 16000, 1)
        for pk in rs.packets():
            assert pk["samples"].size == 960
            assert node.process_audio(pk["samples"]) == 0
    rs.finish()
    for pk in rs.packets():
        assert node.process_audio(pk["samples"]) == 0
    node.flush()
    ndjson = node.json_serialize(pretty=False, newline_delimited=True)          # the C++ restatement of core::json_serialize
    assert ndjson == b"".join(b'{"Transcription":' + o[2] + b"}\n" for o in node.outputs())   # the plugin's JSON is already serde's byte for byte
    lines = ndjson.splitlines()
    assert len(lines) == 1
    want = _expected_transcription(om, pcm[:939 * 512], 0)
    assert json.loads(lines[0])["Transcription"] == want
    pretty = node.json_serialize(pretty=True, newline_delimited=False)
    assert pretty.startswith(b'{\n  "Transcription": {\n    "text": ') and json.loads(pretty)["Transcription"] == want
    node.destroy()


def test_concurrent_instances_form_batches_and_keep_their_own_results(plugin, tiny_model_path):
    """Config 2's shape at test size: several instances driven at once (one feeding thread each), whose segments the per-(model, GPU)
    scheduler batches together; every instance must get exactly the transcript it would get alone."""
    om = OracleModel(tiny_model_path)
    clips = [(c, 16000 * s) for c, s in [(1, 30), (2, 12), (3, 30), (4, 5), (5, 21), (6, 30)]]
    pcms = [synth.clip(c, n) for c, n in clips]
    nodes = [plugin.create_node({"model_path": tiny_model_path, "vad_mode": "always", "flush_tail": True, "batch_window_ms": 30, "max_batch": 8}) for _ in clips]
    minihost.run_oneshot(nodes, pcms)
    for node, pcm in zip(nodes, pcms):
        outs = node.outputs()
        assert len(outs) == 1 and json.loads(outs[0][2].decode()) == _expected_transcription(om, pcm, 0)
        node.destroy()


def test_silero_vad_gates_what_whisper_sees(plugin, tiny_model_path):
    """W2: with a Silero model file at vad_model_path (vad_mode auto) the 512-sample frames are gated by the restated v5 network
    (vad.rs:67-120 contract); the oracle's own Silero probabilities + the segmentation oracle say which samples reach Whisper."""
    from silero_lib import OracleSilero, speechlike, synth_silero_path
    om = OracleModel(tiny_model_path)
    vad_path = synth_silero_path()
    node = plugin.create_node({"model_path": tiny_model_path, "vad_model_path": vad_path, "min_silence_duration_ms": 320, "emit_vad_events": True})
    assert not any("energy" in l for l in node.logs())                       # auto -> silero because the file exists
    pattern = ((20, 0.0), (150, 0.25), (40, 0.0), (100, 0.2), (45, 0.0))
    pcm = speechlike(355, seed=2, pattern=pattern)
    _feed(node, pcm, packet=960)
    ov = OracleSilero(vad_path)
    prob = np.array([ov.process_chunk(pcm[i * 512:(i + 1) * 512]) for i in range(355)], np.float32)
    cuts = oracle_lib.segment_sim(prob, 0.5, 320, 30.0)
    assert len(cuts) == 2 and all(c[3] == 1 for c in cuts)                   # two silence cuts
    speech_frames = np.flatnonzero(prob >= 0.5)
    outs = node.outputs()
    assert len(outs) == 2
    pos = 0
    for (start_ms, end_ms, n_samples, _, _, _), out in zip(cuts, outs):
        idx = speech_frames[pos:pos + n_samples // 512]; pos += n_samples // 512
        seg = np.concatenate([pcm[i * 512:(i + 1) * 512] for i in idx])
        assert idx[0] * 32 == start_ms
        assert json.loads(out[2].decode()) == _expected_transcription(om, seg, start_ms)
    tel = [t for t in node.telemetry() if t[0] == "vad.speech_start"]
    assert len(tel) == 2 and abs(tel[0][1]["speech_probability"] - float(prob[speech_frames[0]])) < 1e-6
    node.destroy()


def test_vad_modes_and_failure_strings(plugin, tiny_model_path, tmp_path):
    from silero_lib import synth_silero_path
    with pytest.raises(RuntimeError) as e:                                    # lib.rs:382-383 + vad.rs:46
        plugin.create_node({"model_path": tiny_model_path, "vad_mode": "silero", "vad_model_path": str(tmp_path / "nope.onnx")})
    assert "Failed to initialize VAD: Failed to load VAD model from '%s'" % (tmp_path / "nope.onnx") in str(e.value)
    node = plugin.create_node({"model_path": tiny_model_path, "vad_model_path": str(tmp_path / "nope.onnx")})   # auto + no file: energy gate, logged
    assert any("using vad_mode=energy" in l for l in node.logs())
    assert node.update_params({"model_path": tiny_model_path, "vad_mode": "silero", "vad_model_path": str(tmp_path / "nope.onnx")}) == -1
    assert node.last_error().startswith("Failed to reload VAD: Failed to load VAD model from")                    # lib.rs:557-560
    assert node.update_params({"model_path": tiny_model_path, "vad_model_path": synth_silero_path()}) == 0
    node.destroy()


def test_default_configuration_with_a_q5_1_model_file(plugin):
    """The reference's default model is a q5_1 file (lib.rs:114-116): with default params (precision exact) the drop-in multiplies it the way
    ggml does (q8 activation blocks, integer block dots) and the Transcription packet equals what the oracle's restatement of that arithmetic gives."""
    from conftest import quantized_model
    path = quantized_model("tiny", "q5_1")
    om = OracleModel(path)
    assert om.quant == 7
    node = plugin.create_node({"model_path": path, "vad_mode": "always", "flush_tail": True})
    pcm = synth.clip(6, 16000 * 14)
    _feed(node, pcm); assert node.flush() == 0
    outs = node.outputs()
    assert len(outs) == 1 and json.loads(outs[0][2].decode()) == _expected_transcription(om, pcm, 0)
    node.destroy()


def test_precision_param_f16_mfma(plugin, tiny_model_path):
    """(additive) precision: "f16_mfma" runs the f16 matrix-core kernels behind the same ABI; the transcript equals the exact
    instance's unless the exact argmax was a near-tie (tests/test_gpu_f16.py holds the detailed bar)."""
    pcm = synth.clip(4, 16000 * 12)
    outs = {}
    for prec in ("exact", "f16_mfma"):
        node = plugin.create_node({"model_path": tiny_model_path, "vad_mode": "always", "flush_tail": True, "precision": prec})
        _feed(node, pcm); assert node.flush() == 0
        o = node.outputs(); assert len(o) == 1
        outs[prec] = json.loads(o[0][2].decode()); node.destroy()
    a, b = outs["exact"], outs["f16_mfma"]
    # the whole transcript: identical, or — the only excuse — the engine-level run of the same clip shows a near-tie of the exact argmax at the first differing token
    if a != b:
        from streamkit_amd import engine
        from test_gpu_f16 import same_or_near_tie
        m = engine.Model(tiny_model_path); ctx = engine.Context(m, max_batch=1, max_samples=16000 * 13)
        pp = ctx.default_params(); pp.suppress_nst = 1                                  # the node's default (lib.rs:634)
        re_ = ctx.full_batch([pcm], pp)[0]; ctx.set_precision("f16_mfma"); rf = ctx.full_batch([pcm], pp)[0]
        assert not same_or_near_tie(rf, re_, "plugin precision param, clip 4")      # asserts the margin bound itself
        ctx.close(); m.close()
    assert b["text"] and b["segments"] and [s["start_time_ms"] for s in a["segments"]][:1] == [s["start_time_ms"] for s in b["segments"]][:1]
    with pytest.raises(RuntimeError):
        plugin.create_node({"model_path": tiny_model_path, "precision": "bf8"})


def test_gpu_device_auto_deals_instances_over_gpus(plugin, tiny_model_path):
    """(additive) gpu_device: "auto": instance k -> GPU k mod n.  On a one-GPU box every instance lands on device 0 and shares one engine."""
    import torch
    n_dev = torch.cuda.device_count()
    nodes = [plugin.create_node({"model_path": tiny_model_path, "vad_mode": "always", "flush_tail": True, "gpu_device": "auto"}) for _ in range(max(2, n_dev))]
    pcm = synth.clip(6, 16000 * 6)
    want = None
    for nd in nodes:
        _feed(nd, pcm); assert nd.flush() == 0
        o = nd.outputs(); assert len(o) == 1
        want = want or o[0][2]
        assert o[0][2] == want                                                # every GPU gives the same transcript (exact mode)
        nd.destroy()
    with pytest.raises(RuntimeError):
        plugin.create_node({"model_path": tiny_model_path, "gpu_device": "first"})


def test_input_sample_rate_48k_equals_resampler_node_then_whisper(plugin, tiny_model_path):
    """(additive) input_sample_rate: 48 kHz packets straight into libwhisper.so give the Transcription the two-node chain gives —
    libresampler.so (the audio::resampler node's arithmetic, R1-R3) -> libwhisper.so — and what the oracle's resampler + full() give.
    (Opus decode always yields 48 kHz, crates/nodes/src/audio/codecs/opus.rs:70,103; SURVEY.md section 8b "Resampler boundary".)"""
    import os
    rs = minihost.Plugin(os.path.join(minihost.ROOT, "streamkit_amd", "libresampler.so"))
    om = OracleModel(tiny_model_path)
    n48 = 48000 * 9 + 1234                                                   # ragged: a remainder chunk at the end (R3)
    t = np.arange(n48) / 48000.0
    x48 = (0.3 * np.sin(2 * np.pi * 310.0 * t) * (0.6 + 0.4 * np.sin(2 * np.pi * 3.1 * t)) + 0.2 * np.sin(2 * np.pi * 1270.0 * t)).astype(np.float32)
    # (a) the chain of two plugins, 20 ms packets as the Opus decoder emits them
    r = rs.create_node({"target_sample_rate": 16000, "chunk_frames": 960, "output_frame_size": 960})
    w = plugin.create_node({"model_path": tiny_model_path, "vad_mode": "always", "flush_tail": True})
    for i in range(0, n48, 960):
        assert r.process_audio(x48[i:i + 960], 48000, 1) == 0, r.last_error()
    assert r.flush() == 0
    chain16 = np.concatenate([np.frombuffer(o[2], dtype=np.float32) for o in r.outputs()])
    for o in r.outputs():
        assert w.process_audio(np.frombuffer(o[2], dtype=np.float32)) == 0, w.last_error()
    assert w.flush() == 0
    chain = [json.loads(o[2].decode()) for o in w.outputs()]
    # (b) the one node
    f = plugin.create_node({"model_path": tiny_model_path, "vad_mode": "always", "flush_tail": True, "input_sample_rate": 48000})
    for i in range(0, n48, 960):
        assert f.process_audio(x48[i:i + 960], 48000, 1) == 0, f.last_error()
    assert f.flush() == 0
    fused = [json.loads(o[2].decode()) for o in f.outputs()]
    assert fused == chain and len(fused) == 1
    # (c) the oracle: rubato restatement, then full() on the whole 512-frames the segmenter hands over plus the flushed tail
    orc = oracle_lib.OracleResampler(16000 / 48000, 960, 1)
    parts = [orc.process(x48[i:i + 960])[0] for i in range(0, n48 - n48 % 960, 960)]
    rem = n48 % 960
    parts.append(oracle_lib.OracleResampler(16000 / 48000, rem, 1).process(x48[n48 - rem:])[0])      # the remainder through a fresh resampler (resampler.rs:564-570)
    want16 = np.concatenate(parts)
    assert np.array_equal(want16.view(np.uint32), chain16.view(np.uint32))
    assert fused[0] == _expected_transcription(om, chain16, 0)
    # a packet at the wrong rate names the configured one; 16 kHz instances keep the reference's message
    assert f.process_audio(x48[:960], 16000, 1) != 0 and "configured for 48000Hz input" in f.last_error()
    assert w.process_audio(x48[:960], 48000, 1) != 0 and "Whisper requires 16kHz audio, got 48000Hz" in w.last_error()
    for nd in (r, w, f):
        nd.destroy()


def test_whisper_plugin_contains_cpp_exceptions(plugin, tiny_model_path):
    """A packet whose sample_count cannot be buffered throws std::length_error inside the segmenter's buffer; process_packet returns it as a
    CResult error (nothing unwinds into the host) and the next instance is unaffected."""
    node = plugin.create_node({"model_path": tiny_model_path, "vad_mode": "always"})
    small = np.zeros(16, np.float32)
    rc = minihost.lib().mh_process_audio(node.h, small.ctypes.data, (1 << 61), 16000, 1)
    assert rc != 0 and node.last_error().startswith("Whisper plugin: "), node.last_error()
    node.destroy()
    node = plugin.create_node({"model_path": tiny_model_path, "vad_mode": "always", "flush_tail": True})
    _feed(node, synth.clip(2, 16000 * 4)); assert node.flush() == 0 and len(node.outputs()) == 1
    node.destroy()


def test_engine_workspace_follows_max_segment_duration(plugin, tiny_model_path):
    """The shared engine's workspace is sized from max_segment_duration_secs (31 s of samples by default, not the schema's 121 s) and grows when an
    instance that allows longer segments hands one over."""
    pcm = synth.clip(21, 16000 * 40)
    node = plugin.create_node({"model_path": tiny_model_path, "vad_mode": "always", "flush_tail": True, "max_segment_duration_secs": 45.0, "precision": "f16_mfma", "max_batch": 3})
    _feed(node, pcm); assert node.flush() == 0, node.last_error()
    outs = [json.loads(o[2].decode()) for o in node.outputs()]
    assert len(outs) == 1 and outs[0]["segments"][-1]["end_time_ms"] > 30000       # one 40 s segment, two Whisper windows
    node.destroy()


def _cache_stats():
    import ctypes as C
    import os
    L = C.CDLL(os.path.join(minihost.ROOT, "streamkit_amd", "libwhisper.so"))      # the handle the mini-host already holds
    loads, hits = C.c_int(), C.c_int()
    L.skw_whisper_plugin_cache_stats(C.byref(loads), C.byref(hits))
    return loads.value, hits.value


def _workspace_stats():
    import ctypes as C
    import os
    L = C.CDLL(os.path.join(minihost.ROOT, "streamkit_amd", "libwhisper.so"))
    e, w, n = C.c_int(), C.c_int(), C.c_int()
    L.skw_whisper_plugin_workspace_stats(C.byref(e), C.byref(w), C.byref(n))
    return e.value, w.value, n.value


def test_workspace_is_returned_with_the_last_instance_and_the_model_stays(plugin, micro_model_path):
    """the reference caches the CONTEXT (weights) for the life of the process and drops each instance's WhisperState with the instance (lib.rs:160-180, 376-380).
    The batch workspace is this build's state: the last instance of an engine to go returns it (6.6 GB at 64 rows of Whisper-small), the model stays cached, and the next instance
    gets a workspace back at creation — no model load, same transcript."""
    params = {"model_path": micro_model_path, "vad_mode": "always", "flush_tail": True, "gpu_device": 0, "precision": "exact", "max_batch": 3, "max_segment_duration_secs": 17.0}
    pcm = synth.clip(6, 16000 * 5)
    a = plugin.create_node(params); b = plugin.create_node(params)
    e1, w1, n1 = _workspace_stats()
    loads1 = _cache_stats()[0]
    _feed(a, pcm); assert a.flush() == 0; out_a = [o[2] for o in a.outputs()]
    a.destroy()
    e2, w2, n2 = _workspace_stats()
    assert (e2, w2, n2) == (e1, w1, n1 - 1)                 # one instance still holds the engine: its workspace stays
    b.destroy()
    e3, w3, n3 = _workspace_stats()
    assert (e3, w3, n3) == (e1, w1 - 1, n1 - 2)             # the last one gone: workspace returned, engine (model) still cached
    c = plugin.create_node(params)
    e4, w4, n4 = _workspace_stats()
    assert (e4, w4, n4) == (e1, w1, n1 - 1) and _cache_stats()[0] == loads1      # back at creation, without a model load
    _feed(c, pcm); assert c.flush() == 0
    assert [o[2] for o in c.outputs()] == out_a and len(out_a) == 1
    c.destroy()


def test_model_cache_outlives_its_instances(plugin, micro_model_path):
    """W5 (lib.rs:170-180, 330-374): the context cache holds strong references for the life of the process, which is what prewarm relies on
    (apps/skit/src/plugins.rs:265-306: the prewarm node is dropped at once, the loaded model stays).  create -> destroy -> create: the second
    create is a CACHE HIT with no second model load, and it transcribes like the first."""
    params = {"model_path": micro_model_path, "vad_mode": "always", "flush_tail": True, "gpu_device": 0, "precision": "exact", "max_batch": 2}
    pcm = synth.clip(4, 16000 * 6)
    loads0, hits0 = _cache_stats()
    first = plugin.create_node(params)                       # the prewarm shape: created ...
    log1 = "\n".join(first.logs())
    loads1, hits1 = _cache_stats()
    first.destroy()                                          # ... and dropped immediately
    if loads1 == loads0 + 1:
        assert "CACHE MISS" in log1 and "Whisper model loaded and cached (model_load_ms=" in log1
    else:                                                    # an earlier test of this process already cached this key: equally a hit
        assert loads1 == loads0 and "CACHE HIT" in log1
    second = plugin.create_node(params)
    loads2, hits2 = _cache_stats()
    assert loads2 == loads1, "the model was loaded again after its only instance went away"
    assert hits2 == hits1 + 1 and any("CACHE HIT" in l for l in second.logs())
    _feed(second, pcm)
    assert second.flush() == 0
    out2 = second.outputs()
    second.destroy()
    third = plugin.create_node(dict(params, max_batch=5, batch_window_ms=0))      # scheduler params follow the latest instance; the cached model serves it
    assert _cache_stats()[0] == loads1
    _feed(third, pcm)
    assert third.flush() == 0
    assert [o[2] for o in third.outputs()] == [o[2] for o in out2] and len(out2) == 1
    third.destroy()


@pytest.mark.parametrize("vad_mode, extra", [("always", {}), ("energy", {"min_silence_duration_ms": 320})])
def test_packetisation_does_not_change_what_comes_out(plugin, tiny_model_path, vad_mode, extra):
    """The host hands the node whatever packet sizes its upstream produces (960-sample resampler packets in the sample pipelines, 1920 from the WAV demuxer, anything from a codec):
    the node re-frames to 512 samples (lib.rs:404-416), so the Transcription packets and the telemetry must not depend on how the same samples were cut into packets.  Six random
    cuttings (1 .. 7 000 samples per packet, empty packets included) of one 70 s stream with pauses against the 960-sample cutting: byte-identical outputs, identical events."""
    rng = np.random.default_rng(21)
    parts = [synth.clip(3, 16000 * 31), np.zeros(16000 * 2, np.float32), synth.clip(4, 16000 * 9), np.zeros(16000, np.float32), synth.clip(5, 16000 * 26), np.zeros(16000, np.float32)]
    pcm = np.concatenate(parts)
    cfg = dict({"model_path": tiny_model_path, "vad_mode": vad_mode, "emit_vad_events": True}, **extra)

    def run(cuts):
        node = plugin.create_node(cfg); pos = 0
        for n in cuts:
            assert node.process_audio(pcm[pos:pos + n]) == 0, node.last_error()
            pos += n
        assert pos >= pcm.size and node.flush() == 0
        out = ([(o[0], o[1], bytes(o[2])) for o in node.outputs()], node.telemetry())
        node.destroy()
        return out

    ref = run([960] * ((pcm.size + 959) // 960))
    assert len(ref[0]) >= 2
    for trial in range(6):
        cuts = []; left = pcm.size
        while left > 0:
            n = int(rng.choice([0, 1, 17, 511, 512, 513, 960, 1920, 4096, int(rng.integers(1, 7000))])); n = min(n, left); cuts.append(n); left -= n
        got = run(cuts)
        assert got[0] == ref[0] and got[1] == ref[1], (trial, len(got[0]), len(ref[0]))


def test_many_instances_with_different_parameters_on_threads_keep_their_own_results(plugin, tiny_model_path):
    """A stress of the scheduler's grouping and routing: 14 instances on 14 OS threads (wrapper.rs:398 calls every instance from blocking-pool threads), with DIFFERENT decode
    parameters and precisions (a batch shares one parameter set, so the scheduler must split them), different clip lengths, packet sizes and arrival times, some destroyed and
    re-created while the others are mid-stream.  Every exact-precision instance must emit exactly the oracle's transcript for its own audio and parameters; every f16_mfma one the packets the
    same configuration emits when run alone (same number and shape; the text may part at a near-tie, see below)."""
    import time
    om = OracleModel(tiny_model_path)
    rng = np.random.default_rng(33)
    jobs = []
    for i in range(14):
        pcm = synth.clip(40 + i, int(16000 * rng.choice([4, 9, 17, 30])))
        cfg = {"model_path": tiny_model_path, "vad_mode": "always", "flush_tail": True, "batch_window_ms": int(rng.choice([0, 2, 20])), "max_batch": int(rng.choice([1, 4, 64])),
               "suppress_blank": bool(rng.integers(0, 2)), "suppress_non_speech_tokens": bool(rng.integers(0, 2)), "language": str(rng.choice(["en", "de", "auto"])),
               "precision": "f16_mfma" if i % 5 == 4 else "exact"}
        jobs.append((cfg, pcm, int(rng.choice([480, 960, 1920, 4000])), float(rng.uniform(0, 0.05))))

    def alone(cfg, pcm):
        node = plugin.create_node(cfg); _feed(node, pcm, 960); assert node.flush() == 0
        out = [bytes(o[2]) for o in node.outputs()]; node.destroy(); return out

    results = [None] * len(jobs); errors = []

    def worker(k):
        try:
            cfg, pcm, packet, delay = jobs[k]
            time.sleep(delay)
            if k % 7 == 3:                      # an instance that is dropped and created again mid-way (a pipeline that restarts)
                tmp = plugin.create_node(cfg); _feed(tmp, pcm[:16000], packet); tmp.destroy()
            node = plugin.create_node(cfg); _feed(node, pcm, packet); assert node.flush() == 0
            results[k] = [bytes(o[2]) for o in node.outputs()]; node.destroy()
        except Exception as e:                  # noqa: BLE001
            errors.append((k, repr(e)))

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(len(jobs))]
    [t.start() for t in ths]; [t.join() for t in ths]
    assert not errors, errors
    for k, (cfg, pcm, _, _) in enumerate(jobs):
        if cfg["precision"] == "exact":
            po = om.default_params(); po.suppress_blank = int(cfg["suppress_blank"]); po.suppress_nst = int(cfg["suppress_non_speech_tokens"])
            po.lang_id = {"en": 0, "de": 2, "auto": -1}[cfg["language"]]
            r = om.full(pcm, po)
            segs = [{"text": s["text"].decode().strip(), "start_time_ms": s["t0"] * 10, "end_time_ms": s["t1"] * 10, "confidence": None} for s in r["segments"] if s["text"].decode().strip()]
            got = [json.loads(b.decode()) for b in results[k]]
            assert len(got) == (1 if segs else 0), k
            if segs:
                assert got[0]["segments"] == segs and got[0]["text"] == " ".join(s["text"] for s in segs), k
                assert got[0]["language"] == cfg["language"], k          # the configured string, "auto" included (lib.rs:687: self.config.language.clone())
        else:
            # f16_mfma is a tolerance precision: which GEMM form a prompt pass takes depends on how many rows share it (>= 256 rows: the big-tile kernel, whose K chain is not the
            # small kernels' four quarters), so a near-tie can fall differently in a batch than alone — seen once in eight runs of this test.  What must hold: the same packets
            # in number and shape, and the same transcript unless the two runs part at a decision (then both are valid f16_mfma transcripts; streamkit_amd/parity.py bounds them).
            ref = alone(cfg, pcm)
            assert len(results[k]) == len(ref), k
            for a, b in zip(results[k], ref):
                ja, jb = json.loads(a.decode()), json.loads(b.decode())
                assert list(ja.keys()) == list(jb.keys()) and ja["language"] == jb["language"] and len(ja["segments"]) >= 1 and ja["segments"][0]["start_time_ms"] == jb["segments"][0]["start_time_ms"], k


def test_an_instance_continues_its_own_ladder_generator_across_segments(plugin):
    """The reference node owns one whisper_state per instance (lib.rs:377-379) and whisper.cpp's sampled passes draw from that state's std::mt19937, which is never re-seeded: the
    second segment of a stream that needs the temperature ladder draws from where the first stopped.  A model whose greedy pass fails the default log-prob threshold (every window
    needs one sampled pass) through two instances at once, three forced-cut segments each: every Transcription equals the oracle's run of THAT instance's segments in order on
    one continuing generator — whichever batch a segment landed in — and the second and third differ from what a freshly seeded generator would give."""
    from conftest import _ensure_built
    import subprocess
    path = "/tmp/skw_test_micro_gamma8.bin"
    if not os.path.exists(path):
        subprocess.check_call([_ensure_built(), path + ".tmp", "--size", "micro", "--gamma_text", "8"]); os.replace(path + ".tmp", path)
    om = OracleModel(path); po = om.default_params(); po.suppress_nst = 1
    from streamkit_amd import engine
    streams = [synth.clip(61, 16000 * 26), synth.clip(62, 16000 * 25)]
    cfg = {"model_path": path, "vad_mode": "always", "max_segment_duration_secs": 8.0, "batch_window_ms": 40}
    nodes = [plugin.create_node(cfg) for _ in streams]
    ths = [threading.Thread(target=_feed, args=(n, s, 960)) for n, s in zip(nodes, streams)]
    [t.start() for t in ths]; [t.join() for t in ths]
    differs_from_fresh = 0
    for node, pcm in zip(nodes, streams):
        outs = [json.loads(o[2].decode()) for o in node.outputs()]
        cuts = oracle_lib.segment_sim(np.ones(pcm.size // 512, np.float32), 0.5, 700, 8.0)
        assert len(cuts) == 3 and len(outs) == 3
        state = engine.rng_state_new(); pos = 0
        for k, (cut, got) in enumerate(zip(cuts, outs)):
            seg = pcm[pos:pos + cut[2]]; pos += cut[2]
            r = om.full(seg, po, rng_state=state)
            assert r["fallback_requested"] >= 1
            want = [{"text": s["text"].decode().strip(), "start_time_ms": cut[0] + s["t0"] * 10, "end_time_ms": cut[0] + s["t1"] * 10, "confidence": None} for s in r["segments"] if s["text"].decode().strip()]
            assert got["segments"] == want, k
            fresh = om.full(seg, po)
            if k > 0 and [s["text"] for s in fresh["segments"]] != [s["text"] for s in r["segments"]]:
                differs_from_fresh += 1
        node.destroy()
    assert differs_from_fresh >= 2


def test_instance_churn_does_not_grow_device_or_host_memory(plugin, micro_model_path):
    """A server creates and drops instances for every Oneshot request and every session: 60 cycles of (4 instances, 3 s of audio each, flush, destroy) after a warm-up cycle must
    leave the device's free memory and the process's resident set where they were (the model stays cached by design; the workspace, the pinned staging buffers, the per-instance
    VAD and generator state and the results must all come back)."""
    import torch
    import psutil
    cfg = {"model_path": micro_model_path, "vad_mode": "always", "flush_tail": True, "batch_window_ms": 1}
    pcm = synth.clip(70, 16000 * 3)

    def cycle():
        nodes = [plugin.create_node(cfg) for _ in range(4)]
        for n in nodes:
            _feed(n, pcm, 960); assert n.flush() == 0; assert len(n.outputs()) == 1
        for n in nodes:
            n.destroy()

    keep = plugin.create_node(cfg)                      # one instance stays: the engine's workspace is not torn down and rebuilt every cycle
    for _ in range(3):
        cycle()
    torch.cuda.synchronize(); free0 = torch.cuda.mem_get_info()[0]; rss0 = psutil.Process().memory_info().rss
    for _ in range(60):
        cycle()
    torch.cuda.synchronize(); free1 = torch.cuda.mem_get_info()[0]; rss1 = psutil.Process().memory_info().rss
    keep.destroy()
    assert free0 - free1 < 64 << 20, "device memory shrank by %.1f MB over 240 instances" % ((free0 - free1) / 2 ** 20)
    assert rss1 - rss0 < 96 << 20, "resident set grew by %.1f MB over 240 instances" % ((rss1 - rss0) / 2 ** 20)
