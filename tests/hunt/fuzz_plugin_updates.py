"""Differential hunt for `update_params` on a live Whisper node (W6; plugins/native/whisper/src/lib.rs:496-577): while a stream is being fed, the host replaces the node's
configuration at random packet boundaries — vad_threshold, min_silence_duration_ms, max_segment_duration_secs, emit_vad_events, suppression flags — and the node must behave as the
reference's does: the new configuration is a FRESH deserialisation (missing keys fall back to their defaults, not to the previous values); the speech buffer, the frame remainder,
the counters and the clock survive; silence_threshold_frames follows min_silence_duration_ms; the maximum duration and the threshold are read per frame.  Energy gate (stateless, so
the VAD rebuild on a threshold change is invisible), a Python restatement of the loop that takes the same updates at the same packet boundaries, the oracle on every cut.
Usage (GPU box): python tests/hunt/fuzz_plugin_updates.py [cases] [seed]"""
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

DEFAULTS = dict(vad_threshold=0.5, min_silence_duration_ms=700, max_segment_duration_secs=30.0, emit_vad_events=False, suppress_blank=True, suppress_non_speech_tokens=True)


class Loop:
    """the reference node's state across packets (lib.rs:381-402 fields, :404-494 loop)"""

    def __init__(self, cfg):
        self.cfg = dict(DEFAULTS, **cfg); self.frame_buffer = np.zeros(0, np.float32); self.buf = []; self.abs_ms = 0; self.start = 0; self.counter = 0; self.seg_id = None
        self.silence = 0; self.sil_thr = self.cfg["min_silence_duration_ms"] // 32; self.cuts = []; self.events = []

    def update(self, cfg):
        new = dict(DEFAULTS, **cfg)
        if new["min_silence_duration_ms"] != self.cfg["min_silence_duration_ms"]:
            self.sil_thr = new["min_silence_duration_ms"] // 32
        self.cfg = new

    def cut(self, end_ms, reason, sil):
        if self.cfg["emit_vad_events"] and self.seg_id is not None:
            self.events.append(("vad.speech_end", self.seg_id, end_ms, reason, sil))
        self.seg_id = None
        self.cuts.append((self.start, np.concatenate(self.buf), dict(self.cfg))); self.buf = []; self.silence = 0

    def push(self, samples):
        self.frame_buffer = np.concatenate([self.frame_buffer, samples])
        while self.frame_buffer.size >= 512:
            fr = self.frame_buffer[:512]; self.frame_buffer = self.frame_buffer[512:]
            rms = np.sqrt(np.float32((fr ** 2).sum(dtype=np.float32)) / np.float32(512.0)); prob = np.float32(rms / (rms + np.float32(0.01)))
            if prob >= np.float32(self.cfg["vad_threshold"]):
                self.silence = 0
                if not self.buf:
                    self.start = self.abs_ms; self.counter += 1; self.seg_id = "seg-%d-%d" % (self.start, self.counter)
                    if self.cfg["emit_vad_events"]:
                        self.events.append(("vad.speech_start", self.seg_id, self.start))
                self.buf.append(fr)
                if self.abs_ms - self.start >= int(np.float32(self.cfg["max_segment_duration_secs"]) * np.float32(1000.0)):
                    self.cut(self.abs_ms + 32, "max_duration", None)
            else:
                self.silence += 1
                if self.buf and self.silence >= self.sil_thr:
                    self.cut(self.abs_ms - (self.silence - 1) * 32, "silence", self.silence * 32)
            self.abs_ms += 32


if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    path = conftest.synth_model("tiny"); om = OracleModel(path)
    plugin = minihost.Plugin(); bad = 0; t0 = time.time(); n_cuts = n_upd = 0

    def random_cfg():
        c = {}
        for k, choices in [("vad_threshold", [0.5, 0.3, 0.8]), ("min_silence_duration_ms", [700, 320, 96, 1500]), ("max_segment_duration_secs", [30.0, 5.0, 9.5]),
                           ("emit_vad_events", [True, True, False]), ("suppress_blank", [True, False]), ("suppress_non_speech_tokens", [True, False])]:
            if rng.random() < 0.7:                  # a key left out falls back to its DEFAULT on an update
                c[k] = choices[int(rng.integers(0, len(choices)))]
        return c

    for case in range(cases):
        parts = []
        for _ in range(int(rng.integers(2, 7))):
            parts.append(synth.clip(int(rng.integers(0, 1000)), int(rng.integers(2000, 16000 * 12))) * np.float32(rng.choice([1.0, 1.0, 0.3, 0.05])))
            parts.append(np.zeros(int(rng.integers(0, 16000 * 2)), np.float32))
        pcm = np.concatenate(parts).astype(np.float32)
        c0 = random_cfg(); fixed = {"model_path": path, "vad_mode": "energy"}
        node = plugin.create_node(dict(fixed, **c0)); ref = Loop(c0); pos = 0
        while pos < pcm.size:
            k = int(rng.choice([960, 960, 1920, int(rng.integers(1, 8000))]))
            assert node.process_audio(pcm[pos:pos + k]) == 0, node.last_error(); ref.push(pcm[pos:pos + k]); pos += k
            if rng.random() < 0.01:
                c = random_cfg(); assert node.update_params(dict(fixed, **c)) == 0; ref.update(c); n_upd += 1
        assert node.flush() == 0
        got = [json.loads(o[2].decode()) for o in node.outputs()]; tel = node.telemetry(); node.destroy()
        want = []
        for (s0, samples, cfg) in ref.cuts:
            po = om.default_params(); po.suppress_blank = int(cfg["suppress_blank"]); po.suppress_nst = int(cfg["suppress_non_speech_tokens"])
            r = om.full(samples, po)
            segs = [{"text": s["text"].decode().strip(), "start_time_ms": s0 + s["t0"] * 10, "end_time_ms": s0 + s["t1"] * 10, "confidence": None} for s in r["segments"] if s["text"].decode().strip()]
            if segs:
                want.append({"text": " ".join(s["text"] for s in segs), "segments": segs, "language": "en", "metadata": None})
        n_cuts += len(ref.cuts)
        got_tel = [((t[0], t[1]["segment_id"], t[1]["start_time_ms"]) if t[0] == "vad.speech_start" else (t[0], t[1]["segment_id"], t[1]["end_time_ms"], t[1]["reason"], t[1]["silence_duration_ms"])) for t in tel]
        if got != want or got_tel != ref.events:
            bad += 1
            print("MISMATCH case %d: %d samples, first config %s: %d / %d transcriptions, %d / %d events" % (case, pcm.size, c0, len(got), len(want), len(got_tel), len(ref.events)), flush=True)
            for a, b in zip(got_tel, ref.events):
                if a != b:
                    print("   first differing event: %s / %s" % (a, b), flush=True); break
        if case % 10 == 9:
            print("case %d: %d cuts, %d updates so far, %d mismatches, %.0f s" % (case, n_cuts, n_upd, bad, time.time() - t0), flush=True)
    print("DONE: %d cases, %d cuts, %d updates, %d mismatches" % (cases, n_cuts, n_upd, bad))
    sys.exit(1 if bad else 0)
