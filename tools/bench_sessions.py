#!/usr/bin/env python3
"""BASELINE config 4 (Dynamic sessions) through the plugin boundary: N live streams on one GPU, 960-sample packets paced at real time
(60 ms), utterances separated by silence so the (energy) VAD closes a segment every few seconds; reports the segment-end -> transcript
latency distribution.  Streams are replicas (no collectives); on a multi-GPU host each instance would set its own `gpu_device`.
usage: python tools/bench_sessions.py [--streams 8] [--seconds 24] [--size small]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--streams", type=int, default=8); ap.add_argument("--seconds", type=float, default=24.0); ap.add_argument("--size", default="small")
    ap.add_argument("--utterance-s", type=float, default=4.0); ap.add_argument("--gap-s", type=float, default=1.0); ap.add_argument("--batch-window-ms", type=float, default=2.0)
    ap.add_argument("--stagger-ms", type=float, default=0.0, help="offset between the streams' utterance boundaries (0 = all streams end segments together)")
    ap.add_argument("--precision", default="f16_mfma", choices=["exact", "f16_mfma"])
    a = ap.parse_args()
    import torch  # noqa: F401
    from streamkit_amd import minihost
    from conftest import synth_model
    from streamkit_amd import synth
    path = synth_model(a.size)
    plug = minihost.Plugin(); L = minihost.lib()
    L.mh_run_paced.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_long, C.POINTER(C.c_double), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.mh_run_paced.restype = C.c_int
    n_total = int(a.seconds * 16000); ut = int(a.utterance_s * 16000); gap = int(a.gap_s * 16000)
    pcms = []
    for i in range(a.streams):
        x = np.zeros(n_total, np.float32); pos = int(i * a.stagger_ms * 16); k = 0
        while pos + ut + gap <= n_total:
            x[pos:pos + ut] = synth.clip(100 * i + k, ut); pos += ut + gap; k += 1
        pcms.append(x)
    params = {"model_path": path, "vad_mode": "energy", "min_silence_duration_ms": 500, "batch_window_ms": a.batch_window_ms, "max_batch": max(8, a.streams), "precision": a.precision}
    warm = plug.create_node(params); warm.process_audio(pcms[0][:ut + gap]); warm.destroy()          # model load + first-use costs outside the measurement
    nodes = [plug.create_node(params) for _ in range(a.streams)]
    n = a.streams; max_lat = 64
    hs = (C.c_void_p * n)(*[x.h for x in nodes]); ptrs = (C.c_void_p * n)(*[p.ctypes.data for p in pcms]); ns = (C.c_size_t * n)(*[p.size for p in pcms])
    lat = (C.c_double * (n * max_lat))(); nl = (C.c_int * n)(); wall = C.c_double()
    rc = L.mh_run_paced(hs, n, ptrs, ns, 960, 60000, lat, max_lat, nl, C.byref(wall))
    assert rc == 0, [x.last_error() for x in nodes]
    ls = np.array([lat[i * max_lat + j] for i in range(n) for j in range(nl[i])])
    for x in nodes: x.destroy()
    print(json.dumps({"what": "Dynamic sessions: %d paced streams on one GPU, %.0f s each, %.1f s utterances" % (n, a.seconds, a.utterance_s), "segments": int(ls.size),
                      "latency_ms": {"p50": round(float(np.percentile(ls, 50)), 1), "p95": round(float(np.percentile(ls, 95)), 1), "max": round(float(ls.max()), 1)},
                      "aggregate_rtf": round(n * a.seconds / (wall.value * 1e-3), 2), "wall_s": round(wall.value * 1e-3, 2), "batch_window_ms": a.batch_window_ms, "stagger_ms": a.stagger_ms, "model": a.size}))


if __name__ == "__main__":
    main()
