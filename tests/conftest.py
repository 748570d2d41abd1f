import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


HAVE_GPU = _have_gpu()

# The oracle's OpenMP regions are many and short (a decode step is dozens).  With spinning waits, anything that takes a core away — a second pytest,
# numpy's busy-waiting BLAS workers, torch's own team, a hypervisor descheduling a vCPU — turns every barrier into a scheduler quantum: a 40 s suite was
# seen to stall for > 15 min, intermittently, in the 8-vCPU container the CPU tier runs in.  There: passive waits and single-threaded BLAS (70 s, also
# under six busy loops).  On the GPU box (dedicated cores) the parity tests lean on the oracle and passive waits triple their time (16 min instead of
# 5.5), so it keeps a bounded spin.  Set here, before oracle/libskw_oracle.so brings in the system libgomp (torch, imported above, carries its own copy).
if HAVE_GPU:
    os.environ.setdefault("GOMP_SPINCOUNT", "100000")     # ~50 us of spinning, then sleep
else:
    os.environ.setdefault("GOMP_SPINCOUNT", "2000")
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    os.environ.setdefault("OPENBLAS_NUM_THREADS", "1")
    os.environ.setdefault("MKL_NUM_THREADS", "1")


def pytest_collection_modifyitems(config, items):
    if HAVE_GPU:
        return
    skip = pytest.mark.skip(reason="no GPU in this environment")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def _ensure_built():
    tool = os.path.join(ROOT, "tools", "make_synth_model")
    if not os.path.exists(tool):
        subprocess.check_call(["gcc", "-O2", "-o", tool, tool + ".c", "-lm"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "libskw_oracle.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    for lib in ("libskw_engine.so", "libwhisper.so", "libresampler.so", "libskw_minihost.so", "libskw_vad.so", "libskw_tts.so", "libkokoro.so", "libskw_dist.so"):
        if not os.path.exists(os.path.join(ROOT, "streamkit_amd", lib)):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "streamkit_amd", "csrc")])
            break
    return tool


def synth_model(size, seed=1234, vocab=51865, mels=80, layers=None):
    """vocab 51864: the English-only files' vocabulary (*.en, the reference's default model); 51866 + mels 128: large-v3's; layers=(encoder, decoder): turbo's unequal counts"""
    tool = _ensure_built()
    tag = size + ("" if (vocab, mels) == (51865, 80) else "_v%d_m%d" % (vocab, mels)) + ("_l%d_%d" % layers if layers else "")
    path = "/tmp/skw_test_%s_%d.bin" % (tag, seed)
    if not os.path.exists(path):
        extra = ["--audio-layers", str(layers[0]), "--text-layers", str(layers[1])] if layers else []
        subprocess.check_call([tool, path + ".tmp", "--size", size, "--seed", str(seed), "--vocab", str(vocab), "--mels", str(mels)] + extra)
        os.replace(path + ".tmp", path)
    return path


def quantized_model(size, kind, seed=1234, vocab=51865, mels=80):
    """The synthetic model re-encoded with block-quantised 2-D weights (tools/quantize_ggml.py), as whisper.cpp's quantize tool lays them out."""
    src = synth_model(size, seed, vocab, mels)
    path = src[:-4] + "_%s.bin" % kind
    if not os.path.exists(path):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "quantize_ggml.py"), src, path + ".tmp", kind])
        os.replace(path + ".tmp", path)
    return path


@pytest.fixture(scope="session")
def eng():
    from streamkit_amd import engine
    return engine


@pytest.fixture(scope="session", autouse=True)
def _alloc_poison():
    """SKW_TEST_ALLOC_POISON=1 python -m pytest tests -m gpu: every floating-point workspace buffer the engine does not zero starts as NaNs (skw_debug_alloc_poison, skw_engine.hip), so a
    kernel that reads what no kernel wrote changes a result instead of passing on whatever hipMalloc returned.  profiles/r05t holds the suite's log in that mode."""
    on = os.environ.get("SKW_TEST_ALLOC_POISON") == "1" and HAVE_GPU
    if on:
        from streamkit_amd import engine
        engine.lib().skw_debug_alloc_poison(1)
        import ctypes as _C
        _C.CDLL(os.path.join(ROOT, "streamkit_amd", "libskw_tts.so"), mode=_C.RTLD_GLOBAL).skw_tts_debug_alloc_poison(1)
    yield
    if on:
        engine.lib().skw_debug_alloc_poisoned.restype = ctypes.c_long
        n = engine.lib().skw_debug_alloc_poisoned()
        print("\n[alloc poison] %d workspace buffers started as NaNs" % n)
        assert n > 0


@pytest.fixture(scope="session")
def built():
    _ensure_built()
    return ROOT


@pytest.fixture(scope="session")
def tiny_model_path():
    return synth_model("tiny")


@pytest.fixture(scope="session")
def micro_model_path():
    return synth_model("micro")


@pytest.fixture(scope="session")
def small_model_path():
    return synth_model("small")


@pytest.fixture(scope="session")
def oracle_tiny(tiny_model_path):
    from oracle_lib import OracleModel
    return OracleModel(tiny_model_path)


@pytest.fixture(scope="session")
def oracle_micro(micro_model_path):
    from oracle_lib import OracleModel
    return OracleModel(micro_model_path)
