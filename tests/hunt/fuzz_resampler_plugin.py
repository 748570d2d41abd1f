"""Differential hunt for the resampler NODE (R1-R4 together): libresampler.so through the native ABI against the C++ restatement of crates/nodes audio::resampler in the mini-host —
random input rates, targets, channel counts, chunk_frames, output_frame_size (0 = as produced, or a multiple the node accepts), random packet sizes and a tail: the packets that
come out must be the same in number, size and every sample (bit for bit), incl. the remainder path with its fresh resampler and the final short frame.
Usage (GPU box): python tests/hunt/fuzz_resampler_plugin.py [cases] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from streamkit_amd import minihost  # noqa: E402

RATES = [8000, 11025, 16000, 22050, 24000, 32000, 44100, 48000, 96000]

if __name__ == "__main__":
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    p = minihost.Plugin(os.path.join(ROOT, "streamkit_amd", "libresampler.so"))
    bad = refused = 0; t0 = time.time()
    for case in range(cases):
        fin, fout = int(rng.choice(RATES)), int(rng.choice(RATES)); ch = int(rng.integers(1, 3))
        chunk = int(rng.choice([160, 480, 960, 1024, 2048])); ofs = int(rng.choice([0, 0, 160, 320, 480, 960, 1920]))
        n_frames = int(rng.integers(1, fin * 2)); x = (rng.standard_normal(n_frames * ch) * 0.3).astype(np.float32)
        cfg = {"target_sample_rate": fout, "chunk_frames": chunk, "output_frame_size": ofs}
        try:
            node = p.create_node(cfg)
        except RuntimeError:
            try:
                minihost.Resampler(fout, chunk, ofs); bad += 1; print("plugin refused a configuration the host node takes: %s" % cfg, flush=True)
            except Exception:
                refused += 1
            continue
        ref = minihost.Resampler(fout, chunk, ofs)
        pos = 0
        while pos < n_frames:
            k = min(n_frames - pos, int(rng.choice([1, 160, 441, 960, 1920, int(rng.integers(1, 5000))])))
            assert node.process_audio(x[pos * ch:(pos + k) * ch], fin, ch) == 0, node.last_error()
            ref.push(x[pos * ch:(pos + k) * ch], fin, ch); pos += k
        assert node.flush() == 0; ref.finish()
        got = [np.frombuffer(o[2], dtype=np.float32) for o in node.outputs()]; exp = [pk["samples"] for pk in ref.packets()]
        ok = [g.size for g in got] == [e.size for e in exp] and all(np.array_equal(g.view(np.uint32), e.view(np.uint32)) for g, e in zip(got, exp))
        if not ok:
            bad += 1; print("MISMATCH case %d: %d -> %d Hz, %d ch, %s, %d frames: %d / %d packets" % (case, fin, fout, ch, cfg, n_frames, len(got), len(exp)), flush=True)
        node.destroy()
        if case % 50 == 49:
            print("case %d: %d mismatches, %d configurations refused by both, %.0f s" % (case, bad, refused, time.time() - t0), flush=True)
    print("DONE: %d cases, %d mismatches, %d refused by both" % (cases, bad, refused))
    sys.exit(1 if bad else 0)
