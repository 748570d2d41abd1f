"""CPU tests of the drop-in boundary: the C-ABI libraries load and export every symbol the headers declare, the plugin's
static metadata matches the reference node's, and nothing computes without a GPU (loud failure, no fallback)."""
import ctypes as C
import os
import re

import pytest

from streamkit_amd import minihost
from conftest import HAVE_GPU, ROOT


def _declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = set()
    for m in re.finditer(r"\b(skw_[a-z0-9_]+|streamkit_native_plugin_api)\s*\(", text):
        names.add(m.group(1))
    return sorted(names)


def test_engine_exports_every_declared_symbol(built):
    L = C.CDLL(os.path.join(ROOT, "streamkit_amd", "libskw_engine.so"))
    names = _declared_functions("skw_engine.h")
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), "libskw_engine.so lacks %s" % n


def test_vad_library_exports_every_declared_symbol(built):
    L = C.CDLL(os.path.join(ROOT, "streamkit_amd", "libskw_vad.so"))
    names = _declared_functions("skw_vad.h")
    assert names == ["skw_vad_create", "skw_vad_free", "skw_vad_process_chunk", "skw_vad_reset", "skw_vad_state"]
    for n in names:
        assert hasattr(L, n), "libskw_vad.so lacks %s" % n


def test_dist_library_exports_every_declared_symbol_and_fails_loudly_without_a_gpu(built):
    """include/skw_dist.h (the transcript gather over RCCL as a C ABI): every declared entry point is exported; with no GPU a group cannot be made and says why"""
    L = C.CDLL(os.path.join(ROOT, "streamkit_amd", "libskw_dist.so"))
    names = _declared_functions("skw_dist.h")
    assert names == ["skw_dist_all_gather_tokens", "skw_dist_create_local", "skw_dist_create_rank", "skw_dist_free", "skw_dist_last_error", "skw_dist_n_local", "skw_dist_unique_id",
                     "skw_dist_world"]
    for n in names:
        assert hasattr(L, n), "libskw_dist.so lacks %s" % n
    import torch
    if not torch.cuda.is_available():
        from streamkit_amd import dist as skd
        import pytest
        with pytest.raises(RuntimeError, match="no HIP device"):
            skd.CGather.local([0])


def test_plugin_exports_the_one_symbol(built):
    assert hasattr(C.CDLL(os.path.join(ROOT, "streamkit_amd", "libresampler.so")), "streamkit_native_plugin_api")
    L = C.CDLL(os.path.join(ROOT, "streamkit_amd", "libwhisper.so"))
    assert _declared_functions("streamkit_native_abi.h") == ["streamkit_native_plugin_api"]
    assert hasattr(L, "streamkit_native_plugin_api")


def test_abi_struct_sizes_match_rust_repr_c(built, tmp_path):
    # sizes quoted in SURVEY.md §8b from types.rs (x86-64): CResult 16, CAudioFrame 24, CPacket 24, CAudioFormat 12, CPacketMetadata 48
    src = tmp_path / "sz.c"
    src.write_text('#include "streamkit_native_abi.h"\n#include <stdio.h>\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",sizeof(CResult),sizeof(CAudioFrame),sizeof(CPacket),sizeof(CAudioFormat),sizeof(CPacketMetadata),sizeof(CNativePluginAPI));return 0;}')
    import subprocess
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-o", str(exe), str(src)])
    out = subprocess.check_output([str(exe)]).decode().split()
    assert out == ["16", "24", "24", "12", "48", "56"]


def test_plugin_metadata_matches_reference_node(built):
    # plugins/native/whisper/src/lib.rs:224-320
    p = minihost.Plugin()
    md = p.metadata
    assert md["kind"] == "whisper" and md["registered_as"] == "plugin::native::whisper"
    # the reference's accepted type first; (additive, for input_sample_rate) mono f32 at any rate second — 0 is the host's wildcard (packet_meta.rs:97-109)
    assert md["inputs"] == [{"name": "in", "accepts": [{"type": 0, "sample_rate": 16000, "channels": 1, "sample_format": 0},
                                                       {"type": 0, "sample_rate": 0, "channels": 1, "sample_format": 0}]}]
    assert md["outputs"] == [{"name": "out", "type": 3}]
    assert md["categories"] == ["ml", "speech", "transcription"]
    props = md["param_schema"]["properties"]
    ref_defaults = {"model_path": "models/ggml-base.en-q5_1.bin", "language": "en", "vad_model_path": "models/silero_vad.onnx", "vad_threshold": 0.5,
                    "min_silence_duration_ms": 700, "max_segment_duration_secs": 30.0, "n_threads": 0, "use_gpu": False, "gpu_device": 0,
                    "suppress_blank": True, "suppress_non_speech_tokens": True, "emit_vad_events": False}
    for k, v in ref_defaults.items():
        assert props[k]["default"] == v, k
    assert props["precision"]["default"] == "exact" and props["vad_mode"]["default"] == "auto"      # additive params keep reference behaviour by default
    assert props["input_sample_rate"]["default"] == 16000 and props["input_resample_mode"]["default"] == "linear"
    assert props["gpu_device"]["maximum"] == 7 and props["min_silence_duration_ms"]["minimum"] == 100


def test_gain_plugin_from_reference_speaks_the_same_abi(built, tmp_path):
    # the mini-host must also load the reference's own C exemplar (proves the harness speaks the real ABI)
    ref = "/root/reference/examples/plugins/gain-native-c"
    if not os.path.exists(ref):
        pytest.skip("reference tree not present (GPU box)")
    import subprocess
    so = tmp_path / "libgain.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-I", ref, "-o", str(so), os.path.join(ref, "gain_plugin.c"), "-lm"])
    p = minihost.Plugin(str(so))
    assert p.metadata["kind"] == "gain_c" or "gain" in p.metadata["kind"]
    n = p.create_node({"gain": 2.0})
    import numpy as np
    assert n.process_audio(np.full(16, 0.25, np.float32), 48000, 1) == 0
    outs = n.outputs()
    assert len(outs) == 1 and outs[0][1] == 0
    assert np.allclose(np.frombuffer(outs[0][2], dtype=np.float32), 0.5)
    n.destroy()


@pytest.mark.skipif(HAVE_GPU, reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly(built, micro_model_path):
    from streamkit_amd import engine
    with pytest.raises(RuntimeError, match="no HIP device|no CPU fallback"):
        engine.Model(micro_model_path)
    p = minihost.Plugin()
    with pytest.raises(RuntimeError, match="Plugin failed to create instance"):
        p.create_node({"model_path": micro_model_path})


def test_plugin_rejects_bad_config_before_touching_the_gpu(built):
    p = minihost.Plugin()
    with pytest.raises(RuntimeError, match="Invalid config"):
        p.create_node({"vad_threshold": "high"})
    with pytest.raises(RuntimeError, match="Invalid config: precision"):
        p.create_node({"precision": "fp8"})
    with pytest.raises(RuntimeError, match="Invalid config: gpu_device"):
        p.create_node({"gpu_device": "first"})
    with pytest.raises(RuntimeError, match="Invalid config: input_sample_rate"):
        p.create_node({"input_sample_rate": 12.5})
    with pytest.raises(RuntimeError, match="Invalid config: input_resample_mode"):
        p.create_node({"input_resample_mode": "cubic"})


def test_no_cpp_exception_crosses_the_abi(built):
    """SURVEY.md section 8b "Errors": panics / C++ exceptions must not cross.  A packet whose sample_count cannot be buffered makes the node's
    std::vector throw std::length_error inside process_packet; the entry point must hand the host a CResult error, not unwind into it.
    (The resampler's pass-through branch needs no GPU; the Whisper node's same guard is exercised on the GPU box, tests/test_gpu_plugin.py.)"""
    import numpy as np
    p = minihost.Plugin(os.path.join(ROOT, "streamkit_amd", "libresampler.so"))
    n = p.create_node({"target_sample_rate": 16000, "output_frame_size": 960})
    small = np.zeros(16, np.float32)
    rc = minihost.lib().mh_process_audio(n.h, small.ctypes.data, (1 << 61), 16000, 1)      # pass-through branch: output_buffer.insert(huge range) throws before it reads
    assert rc != 0 and "resampler plugin:" in n.last_error()
    assert n.process_audio(np.zeros(960, np.float32), 16000, 1) == -2                       # the host's rule: a node whose process failed is Failed (wrapper.rs:468-483)
    n.destroy()                                                                              # ... and is destroyed cleanly
    n2 = p.create_node({"target_sample_rate": 16000, "output_frame_size": 960})              # the library is intact: the next instance works
    assert n2.process_audio(np.zeros(960, np.float32), 16000, 1) == 0 and len(n2.outputs()) == 1
    n2.destroy()
