"""Silero VAD gate: product (libskw_vad.so) == oracle bit for bit, oracle == torch restatement to 1e-5, the ONNX
readers agree, the reference's call contract (vad.rs:67-120: 576-sample window, [2,1,128] state, 64-sample context) and messages."""
import os

import numpy as np
import pytest

import onnx_mini
from silero_lib import OracleSilero, ProductVad, speechlike, synth_silero_path


@pytest.fixture(scope="module")
def built_vad(built):
    return built


def test_product_equals_oracle_bitwise(built_vad):
    path = synth_silero_path()
    pv, ov = ProductVad(path), OracleSilero(path)
    audio = speechlike(200)
    pp, po = [], []
    for i in range(200):
        fr = audio[i * 512:(i + 1) * 512]
        pp.append(pv.process_chunk(fr)); po.append(ov.process_chunk(fr))
        if i % 37 == 0:
            assert np.array_equal(pv.state().view(np.uint32), ov.state.view(np.uint32)), i
    pp, po = np.array(pp, np.float32), np.array(po, np.float32)
    assert np.array_equal(pp.view(np.uint32), po.view(np.uint32))
    # the engineered gate: shut on near-silence, open on the tones (SURVEY §8d "VAD caveat": synthetic audio must still open the gate)
    assert pp[5:18].max() < 0.1 and pp[25:78].min() > 0.9 and pp[90:108].max() < 0.1
    pv.reset(); ov.reset()
    assert not pv.state().any()                                               # reset clears (h, c) and the context (vad.rs:139-142)
    assert pv.process_chunk(audio[:512]) == pp[0] and ov.process_chunk(audio[:512]) == po[0]
    pv.close()


def test_lstm_operator_form_is_the_same_network(built_vad):
    a, b = ProductVad(synth_silero_path()), ProductVad(synth_silero_path(lstm_op=True))
    audio = speechlike(60, seed=3)
    pa = [a.process_chunk(audio[i * 512:(i + 1) * 512]) for i in range(60)]
    pb = [b.process_chunk(audio[i * 512:(i + 1) * 512]) for i in range(60)]
    assert pa == pb
    oa = OracleSilero(synth_silero_path(lstm_op=True))
    assert [oa.process_chunk(audio[i * 512:(i + 1) * 512]) for i in range(60)] == pa
    a.close(); b.close()


def test_oracle_against_torch_restatement(built_vad):
    import torch
    import torch.nn.functional as F
    path = synth_silero_path()
    ov = OracleSilero(path)
    w = {k: torch.from_numpy(v) for k, v in ov.w.items()}
    cell = torch.nn.LSTMCell(128, 128)
    with torch.no_grad():
        cell.weight_ih.copy_(w["decoder.rnn.weight_ih"]); cell.weight_hh.copy_(w["decoder.rnn.weight_hh"])
        cell.bias_ih.copy_(w["decoder.rnn.bias_ih"]); cell.bias_hh.copy_(w["decoder.rnn.bias_hh"])
    audio = speechlike(80, seed=5)
    h = torch.zeros(1, 128); c = torch.zeros(1, 128); ctx = torch.zeros(64)
    worst = 0.0
    with torch.no_grad():
        for i in range(80):
            fr = torch.from_numpy(audio[i * 512:(i + 1) * 512])
            x = torch.cat([ctx, fr])[None]                                        # [1, 576]  (vad.rs:72-80)
            x = F.pad(x[:, None], (0, 64), mode="reflect")                        # [1, 1, 640]
            st = F.conv1d(x, w["stft.forward_basis_buffer"], stride=128)          # [1, 258, 4]
            mag = torch.sqrt(st[:, :129] ** 2 + st[:, 129:] ** 2)
            y = mag
            for l, s in enumerate((1, 2, 2, 1)):
                y = F.relu(F.conv1d(y, w["encoder.%d.reparam_conv.weight" % l], w["encoder.%d.reparam_conv.bias" % l], stride=s, padding=1))
            h, c = cell(y[:, :, 0], (h, c))
            p = torch.sigmoid(F.conv1d(F.relu(h)[:, :, None], w["decoder.decoder.2.weight"], w["decoder.decoder.2.bias"]))[0, 0, 0]
            ctx = fr[-64:]
            po = ov.process_chunk(audio[i * 512:(i + 1) * 512])
            worst = max(worst, abs(float(p) - float(po)))
            assert np.abs(ov.state[0, 0] - h[0].numpy()).max() < 1e-4 and np.abs(ov.state[1, 0] - c[0].numpy()).max() < 1e-4
    assert worst < 1e-5, worst


def test_readers_agree_and_pick_the_16k_subgraph(built_vad):
    path = synth_silero_path()
    ts = onnx_mini.read_tensors(path)
    graphs = sorted({g for g, _, _ in ts})
    assert len(graphs) >= 2                                                       # an 8 kHz and a 16 kHz sub-graph with equally shaped tensors
    g16 = [g for g, _, a in ts if a.shape == (258, 1, 256)][0]
    g8 = [g for g, _, a in ts if a.shape == (130, 1, 128)][0]
    assert g16 != g8 and sum(1 for g, _, a in ts if a.shape == (512, 128)) == 4
    # the product bound the 16 kHz tensors: its output differs from a file whose 16 kHz LSTM is altered, not from one whose 8 kHz LSTM is
    import tools_path  # noqa: F401
    import make_synth_silero as ms
    base = ProductVad(path)
    audio = speechlike(30, seed=9)
    ref = [base.process_chunk(audio[i * 512:(i + 1) * 512]) for i in range(30)]
    for alter16 in (False, True):
        w16, w8 = ms.weights_16k(1234), ms.weights_8k(1234)
        (w16 if alter16 else w8)["decoder.rnn.bias_ih"] += 0.5
        g8b, g16b = ms.branch("If_0_else_branch__Inline_0__", w8), ms.branch("If_0_then_branch__Inline_0__", w16)
        top = [ms.node("If", ["is16k"], ["output"], [ms.attr_graph("else_branch", g8b), ms.attr_graph("then_branch", g16b)])]
        data = ms.vi(1, 8) + ms.ld(7, ms.graph("g", top, []))
        p2 = "/tmp/skw_silero_alt_%d.onnx" % alter16
        open(p2, "wb").write(data)
        v = ProductVad(p2)
        got = [v.process_chunk(audio[i * 512:(i + 1) * 512]) for i in range(30)]
        assert (got != ref) == alter16
        v.close()
    base.close()


def test_error_messages_follow_the_reference(built_vad, tmp_path):
    with pytest.raises(RuntimeError) as e:
        ProductVad(str(tmp_path / "missing.onnx"))
    assert str(e.value).startswith("Failed to load VAD model from '%s': " % (tmp_path / "missing.onnx"))      # vad.rs:46
    bad = tmp_path / "bad.onnx"; bad.write_bytes(b"\x08\x08\x3a\x02\x12\x00")                                  # a ModelProto with an empty graph
    with pytest.raises(RuntimeError) as e:
        ProductVad(str(bad))
    assert "Failed to load VAD model from" in str(e.value) and "STFT basis" in str(e.value)
    junk = tmp_path / "junk.onnx"; junk.write_bytes(os.urandom(4096))
    with pytest.raises(RuntimeError):
        ProductVad(str(junk))


def test_damaged_onnx_files_are_refused_or_run_but_never_crash(built_vad, tmp_path):
    """`vad_model_path` is a file the operator downloads (README.md:131-140): a truncated download or a corrupted byte must end in skw_vad_create's error string — which the node
    turns into the reference's "Failed to initialize VAD" (lib.rs:382-383) — or in a gate that runs, never in a crash of the host.  Truncations at 30 points and 300 files with
    1 - 3 flipped bytes, half of them next to the graph's structure (field tags and lengths around the tensor names), seeded."""
    import ctypes as C
    import re
    good = open(synth_silero_path(), "rb").read()
    L = C.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "streamkit_amd", "libskw_vad.so"))
    L.skw_vad_create.restype = C.c_void_p; L.skw_vad_create.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
    L.skw_vad_free.argtypes = [C.c_void_p]; L.skw_vad_process_chunk.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_float)]
    rng = np.random.default_rng(3)
    sites = [m.start() for m in re.finditer(rb"(stft|encoder|decoder)[.]", good)]
    assert len(sites) >= 10
    path = str(tmp_path / "damaged.onnx")
    refused = ran = 0

    def attempt(blob):
        nonlocal refused, ran
        with open(path, "wb") as f:
            f.write(blob)
        err = C.create_string_buffer(512)
        h = L.skw_vad_create(path.encode(), err, 512)
        if not h:
            assert len(err.value) > 5
            refused += 1
            return
        x = np.zeros(512, np.float32); pr = C.c_float()
        L.skw_vad_process_chunk(h, x.ctypes.data, C.byref(pr)); L.skw_vad_free(h)
        ran += 1

    for cut in [0, 1, 7, 100, 1000, len(good) // 3, len(good) // 2, len(good) - 1] + [int(x) for x in rng.integers(0, len(good), 22)]:
        attempt(good[:cut])
    assert refused >= 25          # (a cut inside the last tensor's padding may still load)
    for i in range(300):
        b = bytearray(good)
        for _ in range(int(rng.integers(1, 4))):
            site = int(rng.choice(sites)) + int(rng.integers(-12, 4)) if i % 2 else int(rng.integers(0, len(b)))
            b[max(0, min(len(b) - 1, site))] = int(rng.integers(0, 256))
        attempt(bytes(b))
    assert refused + ran == 330
