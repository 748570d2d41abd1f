"""Runs seeded single-instruction problems on the GPU and on the model (oracle/), and saves every element where they differ (operands in slot order, accumulator, both results)
to gpurun_out/mfma_mismatch.npz — the input of the next refinement of include/skw_mfma_model.h.  Usage (GPU box): python tests/hunt/mfma_mismatch_dump.py [problems per kind [seed offset [kind whose tiles to keep]]]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as ol  # noqa: E402
import test_gpu_mfma_model as T  # noqa: E402
from streamkit_amd import engine  # noqa: E402
import ctypes as C  # noqa: E402

if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    SEED0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    tool = os.path.join(ROOT, "tools", "make_synth_model")
    path = "/tmp/skw_probe_tiny.bin"
    os.system("%s %s --size tiny > /dev/null" % (tool, path))
    m = engine.Model(path); c = engine.Context(m, max_batch=1, max_samples=16000)
    L = engine.lib()
    L.skw_debug_mfma16x32.restype = C.c_int
    L.skw_debug_mfma16x32.argtypes = [C.c_void_p, C.c_long, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    if len(sys.argv) > 3:        # also keep whole tiles of one kind with the GPU's outputs (raw material for tests/golden/make_mfma_hw_vectors.py's element fixture)
        a, b, cc = T._operands(np.random.default_rng(SEED0 + 5000), 2048, sys.argv[3])
        np.savez_compressed(os.path.join(ROOT, "gpurun_out", "mfma_tiles_%s.npz" % sys.argv[3]), A=a, B=b, C=cc, D=T._hw_tiles((c, L), a, b, cc))
    rows = {"a": [], "b": [], "c": [], "hw": [], "sw": [], "kind": []}
    for ki, kind in enumerate(["act", "wide", "tie", "tiny", "cross"]):
        for seed in range(4):
            a, b, cc = T._operands(np.random.default_rng(1000 * ki + seed + SEED0), P, kind)
            hw = T._hw_tiles((c, L), a, b, cc); sw = ol.mfma_f16_tiles(a, b, cc)
            ok = np.isfinite(hw) & ((np.abs(hw) >= 2.0 ** -126) | (hw == 0))
            bad = (hw.view(np.uint32) != sw.view(np.uint32)) & ok
            idx = np.argwhere(bad)
            print(kind, seed, "mismatches", len(idx), "of", bad.size, flush=True)
            for p, i, j in idx[:2000]:
                rows["a"].append(a[p, i, :]); rows["b"].append(b[p, :, j]); rows["c"].append(cc[p, i, j]); rows["hw"].append(hw[p, i, j]); rows["sw"].append(sw[p, i, j]); rows["kind"].append(ki)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    np.savez_compressed(os.path.join(ROOT, "gpurun_out", "mfma_mismatch.npz"), a=np.array(rows["a"], np.uint16).reshape(-1, 32), b=np.array(rows["b"], np.uint16).reshape(-1, 32),
                        c=np.array(rows["c"], np.float32), hw=np.array(rows["hw"], np.float32), sw=np.array(rows["sw"], np.float32), kind=np.array(rows["kind"], np.int32))
    print("saved", len(rows["c"]), "cases")
