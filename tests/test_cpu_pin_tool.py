"""tests/pin_against_whisper_cpp.py — the turnkey comparison with the real reference's output (whisper-cli -ojf) for whoever has a trained model file: its parsing and its
divergence report, exercised on a JSON of whisper-cli's shape written from the oracle's own result (there is no whisper.cpp here to produce one)."""
import json
import os
import subprocess
import sys
import wave

import numpy as np

from streamkit_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tests", "pin_against_whisper_cpp.py")


def _ms(cs):
    ms = cs * 10
    return "%02d:%02d:%02d,%03d" % (ms // 3600000, ms // 60000 % 60, ms // 1000 % 60, ms % 1000)


def test_pin_tool_reads_whisper_cli_json_and_reports_the_first_divergence(oracle_tiny, tiny_model_path, tmp_path):
    om = oracle_tiny
    pcm16 = np.clip(np.round(synth.clip(3, 16000 * 14) * 32768.0), -32768, 32767).astype("<i2")
    wav = str(tmp_path / "clip.wav")
    with wave.open(wav, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm16.tobytes())
    po = om.default_params(); po.temperature_inc = 0.0
    r = om.full((pcm16.astype(np.float32) / 32768.0).astype(np.float32), po)
    assert len(r["segments"]) >= 2
    doc = {"systeminfo": "synthetic", "model": {"type": "tiny"}, "params": {"language": "en"}, "result": {"language": "en"}, "transcription": []}
    for s in r["segments"]:
        doc["transcription"].append({"timestamps": {"from": _ms(s["t0"]), "to": _ms(s["t1"])}, "offsets": {"from": s["t0"] * 10, "to": s["t1"] * 10}, "text": s["text"].decode(),
                                     "tokens": [{"text": om.token_bytes(t).decode("utf-8", "replace"), "id": int(t), "p": 0.5, "t_dtw": -1} for t in s["tokens"]]})
    good = str(tmp_path / "good.json"); json.dump(doc, open(good, "w"))
    run = lambda j: subprocess.run([sys.executable, TOOL, "--model", tiny_model_path, "--wav", wav, "--json", j], capture_output=True, text=True, timeout=300)
    ok = run(good)
    assert ok.returncode == 0 and "%d of %d segments identical" % (len(r["segments"]), len(r["segments"])) in ok.stdout, ok.stdout + ok.stderr
    doc["transcription"][1]["tokens"][2]["id"] += 1                                           # one token of the second segment differs
    bad = str(tmp_path / "bad.json"); json.dump(doc, open(bad, "w"))
    ko = run(bad)
    assert ko.returncode == 1 and "FIRST DIVERGENCE: segment 1, token 2" in ko.stdout and "margin" in ko.stdout, ko.stdout + ko.stderr
    doc["transcription"][1]["tokens"][2]["id"] -= 1; doc["transcription"][0]["offsets"]["to"] += 20      # a segment boundary differs by 20 ms
    json.dump(doc, open(bad, "w"))
    ko = run(bad)
    assert ko.returncode == 1 and "segment 0: times differ" in ko.stdout, ko.stdout
