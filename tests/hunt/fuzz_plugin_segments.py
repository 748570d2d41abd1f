"""Differential hunt for the Whisper NODE's framing, gate and cut rules (W1, W3, W4, W7): random streams of tone bursts and pauses, random vad_threshold, min_silence_duration_ms,
max_segment_duration_secs and packet sizes through libwhisper.so (vad_mode energy: the probability of a frame is rms / (rms + 0.01), a closed form both sides can evaluate) against a
Python restatement of the reference's loop (plugins/native/whisper/src/lib.rs:404-494, 582-702: 512-sample frames; speech frames only are buffered; a cut on max duration checked
BEFORE the time advances, or after silence_threshold_frames non-speech frames; end time = abs - (silence_frames - 1) * 32; no flush) with the oracle transcribing each cut:
Transcription packets (text, segments with absolute times, language) and the vad.speech_start / vad.speech_end events must be identical.
Usage (GPU box): python tests/hunt/fuzz_plugin_segments.py [cases] [seed]"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import conftest  # noqa: E402,F401
from oracle_lib import OracleModel  # noqa: E402
from streamkit_amd import minihost, synth  # noqa: E402


def reference_loop(pcm, thr, min_silence_ms, max_secs, gate=None):
    """lib.rs:404-494 on an energy gate -> [(start_ms, end_ms, reason, silence_ms or None, samples, seg_id, speech_probability at start)]"""
    cuts = []; buf = []; abs_ms = 0; seg_start = 0; counter = 0; seg_id = None; silence = 0; p_start = 0.0
    sil_thr = min_silence_ms // 32
    max_ms = int(np.float32(max_secs) * np.float32(1000.0))
    for f in range(pcm.size // 512):
        fr = pcm[f * 512:(f + 1) * 512]
        if gate is not None:
            prob = np.float32(gate.process_chunk(fr))                     # the Silero gate as the oracle evaluates it (vad.rs:67-120: context and LSTM state carried)
        else:
            rms = np.sqrt(np.float32((fr.astype(np.float32) ** 2).sum(dtype=np.float32)) / np.float32(512.0)); prob = np.float32(rms / (rms + np.float32(0.01)))
        if prob >= np.float32(thr):
            silence = 0
            if not buf:
                seg_start = abs_ms; counter += 1; seg_id = "seg-%d-%d" % (seg_start, counter); p_start = float(prob)
            buf.append(fr)
            if abs_ms - seg_start >= max_ms:
                cuts.append((seg_start, abs_ms + 32, "max_duration", None, np.concatenate(buf), seg_id, p_start)); buf = []; silence = 0
        else:
            silence += 1
            if buf and silence >= sil_thr:
                cuts.append((seg_start, abs_ms - (silence - 1) * 32, "silence", silence * 32, np.concatenate(buf), seg_id, p_start)); buf = []; silence = 0
        abs_ms += 32
    return cuts, (seg_id if buf else None, seg_start, p_start)


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    path = conftest.synth_model("tiny"); om = OracleModel(path); po = om.default_params(); po.suppress_nst = 1
    plugin = minihost.Plugin(); bad = 0; t0 = time.time(); n_cuts = 0
    silero = len(sys.argv) > 3 and sys.argv[3] == "silero"      # third argument: the Silero gate (a seeded Silero-shaped model) instead of the energy gate
    if silero:
        from silero_lib import OracleSilero, speechlike, synth_silero_path
        vad_path = synth_silero_path()
    for case in range(cases):
        parts = []
        for _ in range(int(rng.integers(1, 7))):
            parts.append(synth.clip(int(rng.integers(0, 1000)), int(rng.integers(300, 16000 * 14))) * np.float32(rng.choice([1.0, 1.0, 0.3, 0.05])))
            parts.append(np.zeros(int(rng.integers(0, 16000 * 2)), np.float32) + np.float32(rng.choice([0.0, 0.0, 1e-4])))
        pcm = np.concatenate(parts).astype(np.float32)
        if silero:                                                        # the seeded gate opens on speech-like tones and shuts on near-silence (tests/silero_lib.py speechlike)
            pat = tuple((int(rng.integers(3, 120)), float(rng.choice([0.0, 0.0, 0.2, 0.05, 0.5]))) for _ in range(int(rng.integers(2, 8))))
            pcm = speechlike(int(rng.integers(60, 600)), seed=int(rng.integers(0, 10000)), pattern=pat)
            pcm = (pcm * np.float32(rng.choice([1.0, 1.0, 0.5]))).astype(np.float32)
        thr = float(rng.choice([0.5, 0.5, 0.3, 0.8])); ms = int(rng.choice([700, 700, 320, 96, 1500])); mx = float(rng.choice([30.0, 30.0, 5.0, 9.5]))
        cfg = {"model_path": path, "vad_mode": "energy", "vad_threshold": thr, "min_silence_duration_ms": ms, "max_segment_duration_secs": mx, "emit_vad_events": True}
        if silero:
            cfg.update(vad_mode="silero", vad_model_path=vad_path)
        node = plugin.create_node(cfg); pos = 0
        while pos < pcm.size:
            k = int(rng.choice([960, 960, 512, 1920, int(rng.integers(1, 6000))])); assert node.process_audio(pcm[pos:pos + k]) == 0, node.last_error(); pos += k
        assert node.flush() == 0
        got = [json.loads(o[2].decode()) for o in node.outputs()]; tel = node.telemetry(); node.destroy()
        cuts, open_seg = reference_loop(pcm, thr, ms, mx, OracleSilero(vad_path) if silero else None); n_cuts += len(cuts)
        want = []; want_tel = []
        for (s0, e0, reason, sil, samples, sid, p0) in cuts:
            want_tel.append(("vad.speech_start", sid, s0)); want_tel.append(("vad.speech_end", sid, e0, reason, sil))
            r = om.full(samples, po)
            segs = [{"text": s["text"].decode().strip(), "start_time_ms": s0 + s["t0"] * 10, "end_time_ms": s0 + s["t1"] * 10, "confidence": None} for s in r["segments"] if s["text"].decode().strip()]
            if segs:
                want.append({"text": " ".join(s["text"] for s in segs), "segments": segs, "language": "en", "metadata": None})
        if open_seg[0]:
            want_tel.append(("vad.speech_start", open_seg[0], open_seg[1]))
        got_tel = [((t[0], t[1]["segment_id"], t[1]["start_time_ms"]) if t[0] == "vad.speech_start" else (t[0], t[1]["segment_id"], t[1]["end_time_ms"], t[1]["reason"], t[1]["silence_duration_ms"])) for t in tel]
        if got != want or got_tel != want_tel:
            bad += 1
            print("MISMATCH case %d: cfg %s, %d samples: %d / %d transcriptions, %d / %d events" % (case, {k: v for k, v in cfg.items() if k != "model_path"}, pcm.size, len(got), len(want), len(got_tel), len(want_tel)), flush=True)
            for a, b in zip(got_tel, want_tel):
                if a != b:
                    print("   first differing event: %s / %s" % (a, b), flush=True); break
        if case % 10 == 9:
            print("case %d: %d cuts so far, %d mismatches, %.0f s" % (case, n_cuts, bad, time.time() - t0), flush=True)
    print("DONE: %d cases, %d cuts, %d mismatches" % (cases, n_cuts, bad))
    sys.exit(1 if bad else 0)
