#!/usr/bin/env python3
"""Writes tests/golden/kokoro_torch_micro.json: per-stage known answers of the INDEPENDENT torch restatement of the published Kokoro network (tests/kokoro_torch_ref.py: ALBERT is
transformers' AlbertModel, every other module restated from hexgrad/kokoro's modules.py / istftnet.py / model.py) on the seeded "micro" model tools/make_synth_kokoro.py writes,
loaded with load_state_dict(strict=True).  Run in the build container (torch CPU): python tests/golden/make_kokoro_torch_goldens.py

Per case: the token ids, speaker, speed; the predicted durations (integers: exact); and for each stage through the decoder's output — bert, d_en, t_en, the F0 and N curves, the
decoder output — the shape, the largest magnitude and 96 seeded sample positions with their values.  The stages after the curves (harmonic source, spectrum, waveform) integrate F0
into a phase, so two implementations whose curves differ by 1e-6 differ by 1e-2 there: those are compared live, with the curves handed over (tests/test_gpu_kokoro.py
test_generator_matches_the_torch_restatement_given_the_same_curves), and their free-running values are recorded here only as a coarse envelope (RMS, length)."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import kokoro_lib  # noqa: E402
import kokoro_torch_ref as ktr  # noqa: E402

CASES = [("Hello world. This is a test of the synthesiser, 1 2 3!", 50, 1.0), ("Short one.", 7, 1.25), ("A considerably longer sentence, with commas; semicolons: and colons - so that the style row moves.", 3, 0.8)]
STAGES = ("bert", "d_en", "t_en", "f0", "en", "dec")


def main():
    torch.set_num_threads(4)
    d = kokoro_lib.synth_kokoro_dir("micro")
    m = ktr.build_from_tensors(kokoro_lib.load_model_tensors(d))
    voices = kokoro_lib.load_voices(d)
    out = {"model": "tools/make_synth_kokoro.py --size micro --seed 1234", "generator": "tests/golden/make_kokoro_torch_goldens.py", "torch": torch.__version__, "cases": []}
    for k, (text, sid, speed) in enumerate(CASES):
        ids = np.asarray(kokoro_lib.tokenize(text, d), np.int64)
        ref_s = torch.from_numpy(voices[sid, kokoro_lib.style_row(ids.size)].copy()).unsqueeze(0)
        audio, dur, taps = m.forward_with_tokens(torch.from_numpy(ids).unsqueeze(0), ref_s, speed, None, kokoro_lib.source_noise)
        case = {"text": text, "ids": ids.tolist(), "sid": sid, "speed": speed, "style_row": kokoro_lib.style_row(ids.size), "durations": dur.tolist(), "stages": {},
                "wave": {"n": int(audio.numel()), "rms": float(audio.pow(2).mean().sqrt())}}
        rng = np.random.default_rng(100 + k)
        for name in STAGES:
            a = taps[name].numpy().astype(np.float32).reshape(-1)
            pos = np.sort(rng.choice(a.size, size=min(96, a.size), replace=False))
            case["stages"][name] = {"shape": list(taps[name].shape), "max_abs": float(np.abs(a).max()), "positions": pos.tolist(), "values": [float(v) for v in a[pos]]}
        out["cases"].append(case)
    path = os.path.join(HERE, "kokoro_torch_micro.json")
    json.dump(out, open(path, "w"), indent=0)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
