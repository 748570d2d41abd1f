"""CPU tests that pin the ORACLE: against independent restatements (numpy float64 mel, HF Whisper fp32), against the
committed golden vectors, and for internal consistency.  (No reference output exists for this path: parity unpinned,
see oracle/skw_oracle.h.)"""
import json
import os
import sys

import numpy as np
import pytest

import oracle_lib
from ggml_reader import read_ggml
from streamkit_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def _ulp_diff(a, b):
    a = np.asarray(a, np.float32); b = np.asarray(b, np.float32)
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))


def test_expf_logf_accuracy(oracle_micro):
    rng = np.random.default_rng(0)
    x = np.concatenate([-rng.random(200000) * 85.9, rng.random(50000) * 80.0, -np.exp(rng.normal(0, 3, 100000))]).astype(np.float32)
    x = x[x >= -86.0]
    got = oracle_micro.math(0, x)
    ref = np.exp(x.astype(np.float64)).astype(np.float32)
    assert _ulp_diff(got, ref).max() <= 2
    assert np.all(oracle_micro.math(0, np.array([-86.5, -1000.0, -np.inf], np.float32)) == 0.0)
    y = np.exp(rng.normal(0, 15, 300000)).astype(np.float32)
    y = y[(y > 1e-37) & np.isfinite(y)]
    assert _ulp_diff(oracle_micro.math(1, y), np.log(y.astype(np.float64)).astype(np.float32)).max() <= 2
    # monotone where it matters (softmax ordering)
    xs = np.sort(x); e = oracle_micro.math(0, xs)
    assert np.all(np.diff(e) >= 0)


def test_f16_rounding_is_ieee_rne(oracle_micro):
    rng = np.random.default_rng(1)
    x = np.concatenate([rng.normal(0, 1, 200000), rng.normal(0, 1e-5, 100000), rng.normal(0, 3e4, 50000),
                        np.float16(np.arange(0, 65504, 7.25)).astype(np.float64) + 2.0 ** -14]).astype(np.float32)
    h = np.arange(0x0000, 0x7bff, dtype=np.uint16).view(np.float16).astype(np.float32)
    ties = (h[:-1].astype(np.float64) + h[1:].astype(np.float64)) / 2     # exact midpoints are representable in f32
    x = np.concatenate([x, ties.astype(np.float32), -ties.astype(np.float32)])
    with np.errstate(over="ignore"):
        ref = x.astype(np.float16).astype(np.float32)
    got = oracle_micro.math(3, x)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def _numpy_mel(pcm, filters):
    """whisper.cpp's log_mel_spectrogram semantics with an exact float64 DFT (independent of the oracle's fp32 FFT)."""
    n = pcm.size
    padded = np.zeros(n + 480000 + 400, dtype=np.float64)
    padded[200:200 + n] = pcm
    padded[:200] = pcm[200:0:-1]
    n_len = (padded.size - 400) // 160
    n_calc = min((n + 200) // 160 + 1, n_len)
    hann = (0.5 * (1.0 - np.cos(2.0 * np.pi * np.arange(400) / 400))).astype(np.float32).astype(np.float64)
    idx = np.arange(n_calc)[:, None] * 160 + np.arange(400)[None, :]
    fr = padded[idx] * hann
    p = np.abs(np.fft.rfft(fr, axis=1)) ** 2
    mel = np.full((filters.shape[0], n_len), -10.0)
    mel[:, :n_calc] = np.log10(np.maximum(filters.astype(np.float64) @ p.T, 1e-10))
    mel = np.maximum(mel, mel.max() - 8.0)
    return ((mel + 4.0) / 4.0).astype(np.float32), 1 + (n + 200 - 400) // 160


@pytest.mark.parametrize("n_samples", [480000, 16000 * 3 + 77, 480768])
def test_mel_against_float64_restatement(oracle_micro, micro_model_path, n_samples):
    _, filters, _, _ = read_ggml(micro_model_path)
    pcm = synth.clip(5, n_samples)
    mel, n_org = oracle_micro.log_mel(pcm)
    ref, ref_org = _numpy_mel(pcm, filters)
    assert mel.shape == ref.shape and n_org == ref_org
    assert np.abs(mel - ref).max() < 1e-4          # north_star tolerance for mel frames


def test_mel_matches_transformers_feature_extractor(oracle_micro, micro_model_path):
    """K1 against an INDEPENDENT implementation: transformers' WhisperFeatureExtractor (numpy: periodic Hann(400), centred reflect padding, hop 160, power spectrum, Slaney mel
    filterbank, log10 floor 1e-10, max - 8 clamp, (x + 4) / 4).  (i) The 80 x 201 filterbank this repository computes for its model files (tools/make_synth_model.c) is HF's
    `mel_filter_bank(norm="slaney", mel_scale="slaney")`; (ii) the oracle's log-mel of a 30 s clip equals HF's on every frame both define the same way.  They differ by design at
    the clip's end only: whisper.cpp appends zeros after the audio (restated in oracle/skw_frontend.c), HF reflects the padded buffer's end and drops its last frame — frames >= 2998."""
    tr = pytest.importorskip("transformers")
    fe = tr.WhisperFeatureExtractor(feature_size=80, sampling_rate=16000, hop_length=160, chunk_length=30, n_fft=400)
    _, filters, _, _ = read_ggml(micro_model_path)
    hf_filters = np.asarray(fe.mel_filters, np.float64).T                      # [80][201]
    assert filters.shape == hf_filters.shape == (80, 201)
    assert np.abs(filters.astype(np.float64) - hf_filters).max() < 2e-7 * hf_filters.max()
    for seed, n in ((5, 480000), (9, 480000), (3, 16000 * 7 + 123)):
        pcm = synth.clip(seed, n)
        mel, _ = oracle_micro.log_mel(pcm)
        hf = fe(pcm, sampling_rate=16000, return_tensors="np").input_features[0]      # [80][3000], the clip zero-padded to 30 s
        last = min(2998, (n - 200) // 160)                                       # frames whose 400-sample window ends inside the audio see the same samples in both
        # the clamp is max - 8 over all frames: both maxima sit in the audio (the padding is silence), so the floor is shared
        err = float(np.abs(mel[:, :last] - hf[:, :last]).max())
        print("log-mel vs transformers, seed %d, %d samples: max abs difference %.2e over %d frames" % (seed, n, err, last))
        assert err < 2e-4, (seed, n, err)      # measured 5.5e-5 .. 9.0e-5: the oracle follows whisper.cpp in an fp32 FFT, HF computes in float64


def test_mel_frame_counts(oracle_micro):
    # 30 s -> 6000 frames / n_len_org 2999; the plugin's forced cut hands over 30.048 s -> 6004 / 3004
    for n, n_len, n_org in [(480000, 6000, 2999), (480768, 6004, 3004), (16000, 3100, 99)]:
        mel, org = oracle_micro.log_mel(np.zeros(n, np.float32) + 0.01)
        assert mel.shape[1] == n_len and org == n_org


def test_short_input_returns_no_segments(oracle_micro):
    r = oracle_micro.full(synth.clip(0, 1500))       # n_len_org 9 < delta_min = 10 frames (100 ms): "input is too short"
    assert r["segments"] == [] and r["n_windows"] == 0
    r = oracle_micro.full(synth.clip(0, 15000))      # 0.94 s: transcribed since whisper.cpp #2065 (the pre-1.6 rule dropped anything under 1 s)
    assert r["n_windows"] == 1 and len(r["tokens"]) > 0


def test_decoder_prompt_batching_is_exact(oracle_micro):
    pcm = synth.clip(1, 16000 * 5)
    mel, _ = oracle_micro.log_mel(pcm)
    _, ck, cv = oracle_micro.encode(mel)
    d1 = oracle_micro.decoder(ck, cv); a = d1.step([50258, 50259, 50359, 50364, 100], 0)
    d2 = oracle_micro.decoder(ck, cv); d2.step([50258], 0); d2.step([50259, 50359], 1); d2.step([50364], 3); b = d2.step([100], 4)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_golden_tokens(oracle_micro):
    gold = json.load(open(os.path.join(HERE, "golden", "oracle_micro_seed1234.json")))
    p = oracle_micro.default_params(); p.suppress_nst = 1
    for case in gold["cases"]:
        r = oracle_micro.full(synth.clip(case["clip"], case["n_samples"]), p)
        assert [t[0] for t in r["tokens"]] == case["tokens"], case
        assert [[s["t0"], s["t1"], s["text"].decode()] for s in r["segments"]] == case["segments"]
        assert r["n_windows"] == case["n_windows"] and r["fallback_requested"] == case["fallback_requested"]


def _hf_whisper_from_ggml(path):
    """HF transformers' Whisper (fp32) carrying the weights of the GGML file at `path`; returns (model, hparams)."""
    import torch
    import transformers as tr
    hp, _, _, T = read_ggml(path)
    cfg = tr.WhisperConfig(vocab_size=hp["n_vocab"], num_mel_bins=hp["n_mels"], d_model=hp["n_audio_state"], encoder_layers=hp["n_audio_layer"],
                           decoder_layers=hp["n_text_layer"], encoder_attention_heads=hp["n_audio_head"], decoder_attention_heads=hp["n_text_head"],
                           encoder_ffn_dim=4 * hp["n_audio_state"], decoder_ffn_dim=4 * hp["n_text_state"], max_source_positions=hp["n_audio_ctx"],
                           max_target_positions=hp["n_text_ctx"], activation_function="gelu", pad_token_id=50257, bos_token_id=50257,
                           eos_token_id=50257, decoder_start_token_id=50258)
    model = tr.WhisperModel(cfg).eval().float()
    sd = {}
    t = lambda n: torch.from_numpy(T[n].astype(np.float32))
    sd["encoder.conv1.weight"] = t("encoder.conv1.weight"); sd["encoder.conv1.bias"] = t("encoder.conv1.bias").reshape(-1)
    sd["encoder.conv2.weight"] = t("encoder.conv2.weight"); sd["encoder.conv2.bias"] = t("encoder.conv2.bias").reshape(-1)
    sd["encoder.embed_positions.weight"] = t("encoder.positional_embedding")
    sd["encoder.layer_norm.weight"] = t("encoder.ln_post.weight"); sd["encoder.layer_norm.bias"] = t("encoder.ln_post.bias")
    sd["decoder.embed_tokens.weight"] = t("decoder.token_embedding.weight"); sd["decoder.embed_positions.weight"] = t("decoder.positional_embedding")
    sd["decoder.layer_norm.weight"] = t("decoder.ln.weight"); sd["decoder.layer_norm.bias"] = t("decoder.ln.bias")
    def blk(src, dst, cross):
        m = {"attn_ln": "self_attn_layer_norm", "attn.query": "self_attn.q_proj", "attn.key": "self_attn.k_proj", "attn.value": "self_attn.v_proj", "attn.out": "self_attn.out_proj",
             "mlp_ln": "final_layer_norm", "mlp.0": "fc1", "mlp.2": "fc2"}
        if cross:
            m.update({"cross_attn_ln": "encoder_attn_layer_norm", "cross_attn.query": "encoder_attn.q_proj", "cross_attn.key": "encoder_attn.k_proj", "cross_attn.value": "encoder_attn.v_proj", "cross_attn.out": "encoder_attn.out_proj"})
        for a, b in m.items():
            sd[dst + b + ".weight"] = t(src + a + ".weight")
            if src + a + ".bias" in T:
                sd[dst + b + ".bias"] = t(src + a + ".bias")
    for l in range(hp["n_audio_layer"]):
        blk("encoder.blocks.%d." % l, "encoder.layers.%d." % l, False)
    for l in range(hp["n_text_layer"]):
        blk("decoder.blocks.%d." % l, "decoder.layers.%d." % l, True)
    missing, unexpected = model.load_state_dict(sd, strict=False)
    assert not unexpected and all(k.endswith("k_proj.bias") for k in missing), (missing, unexpected)   # whisper has no key bias
    for k in missing:
        model.state_dict()[k].zero_()
    return model, hp


def test_hf_whisper_crosscheck(tiny_model_path, oracle_tiny):
    """Independent implementation check of the ARCHITECTURE (conv stem and padding, positional offsets, attention scaling, LN eps, K without bias, tied logits): HF transformers'
    Whisper in fp32 with the same weights, compared STAGE BY STAGE — conv stem + positions, encoder layer 0, encoder output, decoder logits — each bound at twice what was
    measured (the old 3e-2 / 0.1 sigma would have let a subtly wrong pad or offset through).  HF uses exact-erf GELU and fp32 activations where ggml uses the
    f16 tanh-GELU table and f16-rounded matmul operands, so agreement is ~1e-3 of a stage's range, not bitwise.  Measured (round 5, this container): conv stem 5.5e-4,
    layer 0 6.7e-4, encoder output 9.2e-4 of the stage's largest magnitude; logits 0.0040 sigma over 29 teacher-forced positions, argmax identical at all of them."""
    torch = pytest.importorskip("torch")
    pytest.importorskip("transformers")
    model, hp = _hf_whisper_from_ggml(tiny_model_path)
    pcm = synth.clip(2, 480000)
    mel, _ = oracle_tiny.log_mel(pcm)
    oracle_lib.debug_enable(True)
    try:
        x0_o = oracle_tiny.conv_stem(mel)
        enc_o, ck, cv = oracle_tiny.encode(mel)
        l0_o = oracle_lib.debug_get("l0.x2").reshape(hp["n_audio_ctx"], -1).copy()
    finally:
        oracle_lib.debug_enable(False)
    p = oracle_tiny.default_params(); p.suppress_nst = 1
    ids = [50258, 50259, 50359] + [t[0] for t in oracle_tiny.full(pcm, p)["tokens"]][:28]       # the oracle's own greedy transcript as the teacher
    assert len(ids) >= 23
    with torch.no_grad():
        feats = torch.from_numpy(mel[:, :3000]).unsqueeze(0)
        e = model.encoder
        stem = torch.nn.functional.gelu(e.conv2(torch.nn.functional.gelu(e.conv1(feats)))).permute(0, 2, 1) + e.embed_positions.weight
        out = e(feats, output_hidden_states=True)
        assert float((out.hidden_states[0] - stem).abs().max()) == 0.0                           # (what HF feeds its first layer IS conv stem + positions)
        enc_h = out.last_hidden_state[0].numpy()
        dec = model.decoder(input_ids=torch.tensor([ids]), encoder_hidden_states=out.last_hidden_state).last_hidden_state[0]
        logits_h = (dec @ model.decoder.embed_tokens.weight.T).numpy()
    rel = lambda h, o: float(np.abs(h - o).max() / np.abs(h).max())
    r_stem, r_l0, r_enc = rel(stem[0].numpy(), x0_o), rel(out.hidden_states[1][0].numpy(), l0_o), rel(enc_h, enc_o)
    assert r_stem < 1.1e-3, r_stem
    assert r_l0 < 1.4e-3, r_l0
    assert r_enc < 1.9e-3, r_enc
    d = oracle_tiny.decoder(ck, cv)
    worst, agree = 0.0, 0
    for n in range(3, len(ids) + 1):                                                             # positions 2 .. : every one a decision the greedy decoder makes
        lo = d.step(ids[:3] if n == 3 else ids[n - 1:n], 0 if n == 3 else n - 1)
        lh = logits_h[n - 1]
        worst = max(worst, float(np.abs(lh - lo).max() / lh.std()))
        agree += int(lh.argmax()) == int(lo.argmax())
    assert worst < 0.008, worst
    assert agree == len(ids) - 2 >= 20, (agree, len(ids) - 2)
    print("HF cross-check: conv stem %.2e, layer 0 %.2e, encoder %.2e of range; logits %.4f sigma over %d positions, argmax agrees at all" % (r_stem, r_l0, r_enc, worst, agree))


def test_discrete_distribution_restatement_matches_libstdcxx(tmp_path):
    """The fallback passes sample with std::discrete_distribution over std::mt19937; the oracle restates both (skw_oracle.c).
    Pin the restatement against the real library: tests/cpp/discrete_ref.cpp compiled with g++ here."""
    import ctypes as C
    import subprocess
    exe = str(tmp_path / "discrete_ref")
    subprocess.check_call(["g++", "-O1", "-o", exe, os.path.join(HERE, "cpp", "discrete_ref.cpp")])
    rng = np.random.default_rng(7)
    lib = oracle_lib.lib()
    for seed, n, kind in [(0, 51865, "peaked"), (0, 1000, "flat"), (5, 37, "sparse"), (123, 4096, "tiny")]:
        if kind == "peaked":
            lg = rng.normal(0, 4, n).astype(np.float32); w = np.exp(lg - lg.max()).astype(np.float32); w[rng.random(n) < 0.3] = 0.0
        elif kind == "flat":
            w = np.ones(n, np.float32)
        elif kind == "sparse":
            w = np.zeros(n, np.float32); w[[3, 17, 36]] = [0.25, 0.5, 0.25]
        else:
            w = (rng.random(n) * 1e-30).astype(np.float32)
        n_draws = 700   # crosses the 624-word twist boundary (2 words per draw)
        inp = "%d %d %d\n%s\n" % (seed, n_draws, n, " ".join(repr(float(x)) for x in w))
        ref = [int(x) for x in subprocess.check_output([exe], input=inp.encode()).split()]
        out = (C.c_int32 * n_draws)()
        lib.skwo_discrete_draw(w.ctypes.data_as(C.c_void_p), n, C.c_uint32(seed), n_draws, out)
        assert list(out) == ref, kind


def test_temperature_ladder_in_oracle(micro_model_path):
    """Thresholds that no pass can meet walk the whole ladder (0, 0.2 .. 1.0) and keep the last pass; with the ladder off only the greedy pass runs."""
    om = oracle_lib.OracleModel(micro_model_path)
    pcm = synth.clip(3, 16000 * 12)
    base = om.full(pcm)
    assert base["fallback_requested"] == 0
    p = om.default_params(); p.logprob_thold = 1.0; p.no_speech_thold = 2.0      # avg_logprobs < 1 always; no_speech_prob < 2 always
    lad = om.full(pcm, p)
    assert lad["fallback_requested"] == 6 * lad["n_windows"]
    assert lad["n_decode_steps"] > base["n_decode_steps"]
    p.temperature_inc = 0.0
    off = om.full(pcm, p)
    assert off["fallback_requested"] == off["n_windows"] and [t[0] for t in off["tokens"]] == [t[0] for t in base["tokens"]]
    assert [t[0] for t in lad["tokens"]] != [t[0] for t in base["tokens"]] or lad["n_windows"] != base["n_windows"]   # sampled at t = 1.0


@pytest.mark.parametrize("kind", ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0"])
def test_quantised_ggml_is_read_as_its_dequantised_f16_twin(kind, tmp_path):
    """The reference's default model is ggml-base.en-q5_1.bin.  The loaders decode block-quantised tensors at load time
    (include/skw_ggml_quant.h); an independent numpy decode of the same file, written back as f16, must transcribe identically."""
    from conftest import quantized_model
    from ggml_reader import dequantize_file_to_f16
    qpath = quantized_model("micro", kind)
    twin = str(tmp_path / "twin.bin")
    assert dequantize_file_to_f16(qpath, twin) > 10
    pcm = synth.clip(5, 16000 * 11)
    a = oracle_lib.OracleModel(qpath, quant_mode=0).full(pcm)
    b = oracle_lib.OracleModel(twin).full(pcm)
    assert a["tokens"] == b["tokens"] and a["segments"] == b["segments"] and len(a["tokens"]) > 0


def _np_q8_mul_mat(kind, blocks, n_out, n_in, A):
    """ggml's quantised mul_mat restated step by step in numpy float32 scalars (include/skw_ggml_quant.h (a)), independent of the C code."""
    import struct
    f1 = np.float32
    bb = {"q4_0": 18, "q4_1": 20, "q5_0": 22, "q5_1": 24, "q8_0": 34}[kind]
    nb = n_in // 32
    raw = np.frombuffer(blocks, np.uint8).reshape(n_out * nb, bb)
    qw = np.zeros((n_out * nb, 32), np.int32); dw = np.zeros(n_out * nb, np.float32); mw = np.zeros(n_out * nb, np.float32)
    for i, b in enumerate(raw):
        o = 0
        dw[i] = np.frombuffer(b[o:o + 2].tobytes(), np.float16)[0]; o += 2
        if kind in ("q4_1", "q5_1"):
            mw[i] = np.frombuffer(b[o:o + 2].tobytes(), np.float16)[0]; o += 2
        qh = 0
        if kind in ("q5_0", "q5_1"):
            qh = struct.unpack("<I", b[o:o + 4].tobytes())[0]; o += 4
        qs = b[o:]
        if kind == "q8_0":
            qw[i] = qs.view(np.int8)
            continue
        for j in range(16):
            x0, x1 = int(qs[j]) & 15, int(qs[j]) >> 4
            if kind in ("q5_0", "q5_1"):
                x0 |= ((qh >> j) & 1) << 4; x1 |= ((qh >> (j + 16)) & 1) << 4
            off = 8 if kind == "q4_0" else 16 if kind == "q5_0" else 0
            qw[i, j], qw[i, j + 16] = x0 - off, x1 - off
    rows = A.shape[0]
    out = np.zeros((rows, n_out), np.float32)
    for r in range(rows):
        qa = np.zeros((nb, 32), np.int32); da = np.zeros(nb, np.float32); sa = np.zeros(nb, np.float32)
        for b in range(nb):
            x = A[r, b * 32:(b + 1) * 32]
            amax = f1(np.abs(x).max()); d = f1(amax / f1(127)); idv = f1(f1(1) / d) if d != 0 else f1(0)
            v = (x * idv).astype(np.float32).astype(np.float64)
            q = (np.sign(v) * np.floor(np.abs(v) + 0.5)).astype(np.int32)          # roundf: half away from zero (exact in f64)
            qa[b] = q; da[b] = f1(np.float16(d)); sa[b] = f1(np.float16(f1(f1(int(q.sum())) * d)))
        for n in range(n_out):
            sumf = f1(0)
            for b in range(nb):
                i = n * nb + b; sumi = int((qw[i] * qa[b]).sum())
                if kind == "q4_0":
                    t = f1(f1(f1(sumi) * dw[i]) * da[b])
                else:
                    t = f1(f1(dw[i] * da[b]) * f1(sumi))
                    if kind in ("q4_1", "q5_1"):
                        t = f1(t + f1(mw[i] * sa[b]))
                sumf = f1(sumf + t)
            out[r, n] = sumf
    return out


@pytest.mark.parametrize("kind", ["q4_0", "q4_1", "q5_0", "q5_1", "q8_0"])
def test_ggml_q8_mul_mat_matches_numpy_restatement(kind):
    """The oracle's quantised mul_mat (activation rows -> q8_0 / q8_1 blocks, integer block dots, f32 scales, block-ascending sum)
    against a numpy restatement written from the same description: bit-identical, including rows with zero blocks and ties in roundf."""
    import ctypes as C
    sys.path.insert(0, os.path.join(oracle_lib.ROOT, "tools"))
    from quantize_ggml import quantize_blocks, TYPES
    rng = np.random.default_rng(7)
    n_out, n_in, rows = 6, 96, 4
    W = (rng.standard_normal((n_out, n_in)) * 0.3).astype(np.float32)
    blocks = quantize_blocks(W.reshape(-1, 32), kind)
    A = (rng.standard_normal((rows, n_in)) * 2.0).astype(np.float32)
    A[1, 32:64] = 0.0                                              # a block of zeros: d = 0, id = 0
    A[2, :32] = np.linspace(-127, 127, 32, dtype=np.float32) * 0.5   # amax = 63.5 -> id = 2: values land on .5 (roundf ties, away from zero)
    out = np.zeros((rows, n_out), np.float32)
    buf = (C.c_uint8 * len(blocks)).from_buffer_copy(blocks)
    assert oracle_lib.lib().skwo_debug_linear_q8(TYPES[kind], buf, n_out, n_in, A.ctypes.data, rows, out.ctypes.data) == 0
    want = _np_q8_mul_mat(kind, blocks, n_out, n_in, A)
    assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
    ref = A @ W.T                                                  # and it is a sensible approximation of the real product
    assert np.abs(out - ref).max() < 0.12 * np.abs(ref).max() + 0.5


@pytest.mark.parametrize("kind", ["q5_1", "q8_0"])
def test_quantised_file_runs_ggml_q8_arithmetic_by_default(kind):
    """A uniformly quantised file is multiplied the way ggml does it (q8 activation blocks) unless the f16 twin is asked for; the two
    arithmetics are different computations (logits differ) of nearly the same function."""
    from conftest import quantized_model
    qpath = quantized_model("micro", kind)
    q8 = oracle_lib.OracleModel(qpath)
    tw = oracle_lib.OracleModel(qpath, quant_mode=0)
    assert q8.quant == {"q5_1": 7, "q8_0": 8}[kind] and tw.quant == 0
    pcm = synth.clip(5, 16000 * 11)
    a, b = q8.full(pcm), tw.full(pcm)
    assert len(a["tokens"]) > 0 and len(b["tokens"]) > 0
    la = q8.decode_logits(pcm, [q8.sot()]) if hasattr(q8, "decode_logits") else None
    if la is not None:
        lb = tw.decode_logits(pcm, [tw.sot()])
        assert not np.array_equal(la, lb) and np.abs(la - lb).max() < 0.25 * np.abs(lb).max()


def test_language_auto_detect_is_consistent_in_oracle(micro_model_path):
    """language = "auto" (lang_id < 0): whisper_lang_auto_detect_with_state picks the language token with the largest [sot]-step logit
    on the first window; decoding with that id explicitly must give the same transcript."""
    om = oracle_lib.OracleModel(micro_model_path)
    pcm = synth.clip(21, 16000 * 9)
    p = om.default_params(); p.lang_id = -1
    auto = om.full(pcm, p)
    assert 0 <= auto["lang_id"] < 99
    p.lang_id = auto["lang_id"]
    fixed = om.full(pcm, p)
    assert fixed["lang_id"] == auto["lang_id"] and fixed["tokens"] == auto["tokens"] and fixed["segments"] == auto["segments"]


def test_prompt_past_survives_a_no_speech_window(oracle_tiny):
    """whisper_full_with_state re-inserts the prompt portion it took for a window unconditionally and appends the window's own tokens
    only when it is speech: a 75 s clip whose middle window is classed no-speech (thresholds chosen so) yields segments
    from windows 0 and 2 only, and window 2 is still decoded with window 0's text as its prompt (one extra [prev] + carried tokens
    show up as decoder steps)."""
    pcm = synth.clip(11, 16000 * 75)
    p = oracle_tiny.default_params(); p.no_speech_thold = -1.0; p.logprob_thold = -0.3; p.temperature_inc = 0.0
    r = oracle_tiny.full(pcm, p)
    assert r["n_windows"] == 3
    win = sorted({min(2, max(0, s["t0"]) // 3000) for s in r["segments"][:-1]} | {2})
    assert not any(3000 <= s["t0"] < 5900 for s in r["segments"]) and any(0 <= s["t0"] < 3000 for s in r["segments"]) and win == [0, 2]
    q = oracle_tiny.default_params(); q.temperature_inc = 0.0
    rq = oracle_tiny.full(pcm, q)                                             # with the default thresholds every window is speech
    assert rq["n_windows"] == 3 and len(rq["tokens"]) > len(r["tokens"])
