"""k_gemm16w's K loop under the counters: the Q/K projection shape (96 000 x 768 x 768, per-head f16 epilogue) as the product launches it and with parts switched off
(tools/gemm16w_probe.py's bits), a few launches each, for `rocprofv3 --kernel-trace --pmc ... -- python3 tools/probe/gemm16w_skeleton_pmc.py` (one counter group per run).
The launches of one configuration share a grid size, so the per-kernel CSV rows can be told apart by their order: the configurations run in the order printed."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import synth_model  # noqa: E402
from streamkit_amd import engine  # noqa: E402

m = engine.Model(synth_model("small")); ctx = engine.Context(m, max_batch=1)
L = engine.lib()
L.skw_debug_gemm16.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float)]
NOEPI, NOMFMA, NOW, NOA = 1024, 4096, 8192, 16384
for name, bits in [("full (product kernel)", 0), ("probe build, nothing off", 1 << 20), ("no epilogue", NOEPI), ("no MFMA, no epilogue", NOMFMA | NOEPI), ("skeleton", NOMFMA | NOW | NOA | NOEPI)]:
    ms = C.c_float()
    assert L.skw_debug_gemm16(ctx.h, 96000, 768, 768, 4, 512 | bits, 4, C.byref(ms)) == 0
    print("%-28s %.1f us per launch" % (name, ms.value * 1e3), flush=True)
