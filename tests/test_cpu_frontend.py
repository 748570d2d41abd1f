"""CPU tests: behaviour that the reference SOURCE pins by itself (SURVEY.md §8c last row) — segmentation arithmetic,
resampler packetisation/lengths, JSON wire shapes — checked on the oracle AND on the product's host-side C++ (same inputs)."""
import json

import numpy as np
import pytest

from streamkit_amd import minihost
import oracle_lib

GOLD = json.load(open(__import__("os").path.join(__import__("os").path.dirname(__file__), "golden", "segmentation_goldens.json")))


@pytest.mark.parametrize("impl", ["oracle", "product"])
def test_segmentation_goldens(built, impl):
    sim = oracle_lib.segment_sim if impl == "oracle" else minihost.segment_sim
    for case in GOLD["cases"]:
        prob = np.zeros(case["n_frames"], dtype=np.float32)
        for a, b in case["speech_runs"]:
            prob[a:b] = 1.0
        cuts = sim(prob, case.get("threshold", 0.5), case.get("min_silence_ms", 700), case.get("max_secs", 30.0))
        assert [c[:5] for c in cuts] == case["cuts"], case["name"]


def test_segmentation_forced_cut_is_939_frames(built):
    # lib.rs:455-465 + :487: the cut fires when 32*n >= max_ms BEFORE the increment -> 939 frames = 480768 samples = 30.048 s
    cuts = oracle_lib.segment_sim(np.ones(2000, dtype=np.float32))
    assert cuts[0][:4] == [0, 30048, 939 * 512, 0] and cuts[0][5] == 938
    assert cuts[1][:4] == [30048, 60096, 939 * 512, 0]


def test_segmenter_product_equals_oracle_random(built):
    rng = np.random.default_rng(7)
    for trial in range(40):
        n = int(rng.integers(50, 4000))
        # piecewise-constant speech/silence with random run lengths
        prob = np.zeros(n, dtype=np.float32); i = 0; state = rng.random() < 0.5
        while i < n:
            run = int(rng.integers(1, 400)); prob[i:i + run] = rng.random() * 0.5 + (0.5 if state else 0.0); i += run; state = not state
        thr = float(rng.choice([0.3, 0.5, 0.7])); ms = int(rng.choice([100, 320, 700, 1000, 5000])); mx = float(rng.choice([5.0, 12.5, 30.0, 120.0]))
        assert oracle_lib.segment_sim(prob, thr, ms, mx) == minihost.segment_sim(prob, thr, ms, mx)


def test_segmenter_agrees_with_a_third_restatement_of_the_reference_loop(built):
    """The product's segmenter and the oracle's simulator against a third restatement, written from plugins/native/whisper/src/lib.rs:404-494 in Python for this test: a frame
    is speech when its probability reaches the threshold; only speech frames are buffered; the maximum duration is checked with the time of the frame that was just buffered
    (before the 32 ms advance) and ends the segment at that time + 32; silence_threshold_frames = min_silence_duration_ms / 32 (integer) non-speech frames close a segment at
    abs - (silence_frames - 1) * 32.  One cut = [start_ms, end_ms, samples handed to Whisper, reason (0 max duration, 1 silence), silence ms or -1, index of the closing frame]."""
    def loop(prob, thr, min_silence_ms, max_secs):
        cuts = []; n_buf = 0; abs_ms = 0; start = 0; silence = 0
        sil_thr = min_silence_ms // 32; max_ms = int(np.float32(max_secs) * np.float32(1000.0))
        for f, pr in enumerate(prob):
            if pr >= np.float32(thr):
                silence = 0
                if n_buf == 0:
                    start = abs_ms
                n_buf += 1
                if abs_ms - start >= max_ms:
                    cuts.append([start, abs_ms + 32, n_buf * 512, 0, -1, f]); n_buf = 0; silence = 0
            else:
                silence += 1
                if n_buf and silence >= sil_thr:
                    cuts.append([start, abs_ms - (silence - 1) * 32, n_buf * 512, 1, silence * 32, f]); n_buf = 0; silence = 0
            abs_ms += 32
        return cuts
    rng = np.random.default_rng(17)
    n_cuts = 0
    for trial in range(120):
        n = int(rng.integers(1, 3000))
        prob = np.zeros(n, dtype=np.float32); i = 0; state = rng.random() < 0.5
        while i < n:
            run = int(rng.integers(1, 300)); prob[i:i + run] = rng.random(min(run, n - i)) * 0.5 + (0.5 if state else 0.0) if trial % 2 else rng.random() * 0.5 + (0.5 if state else 0.0)
            i += run; state = not state
        thr = float(rng.choice([0.3, 0.5, 0.5, 0.7, 0.75])); ms = int(rng.choice([31, 32, 100, 320, 700, 1000])); mx = float(rng.choice([0.5, 5.0, 9.5, 12.5, 30.0]))
        want = loop(prob, thr, ms, mx); n_cuts += len(want)
        assert minihost.segment_sim(prob, thr, ms, mx, max_cuts=4096) == want, (trial, thr, ms, mx)
        assert oracle_lib.segment_sim(prob, thr, ms, mx, max_cuts=4096) == want, (trial, thr, ms, mx)
    assert n_cuts > 300


def test_resampler_reference_length_golden(built):
    # resampler.rs:816-837: 960 interleaved stereo samples (480 frames) 48k->24k, chunk_frames 960, output_frame_size 0
    # -> remainder path with a fresh FastFixedIn(480 frames): |len - 480| < 10, rate/channels preserved
    r = minihost.Resampler(24000, 960, 0)
    r.push(np.full(960, 0.5, dtype=np.float32), 48000, 2)
    r.finish()
    pk = r.packets()
    assert len(pk) == 1 and abs(pk[0]["samples"].size - 480) < 10
    assert np.allclose(pk[0]["samples"][8:], 0.5)   # linear interpolation of a constant (after the zero history) is that constant


def test_resampler_48k_to_16k_steady_state_and_packets(built):
    # R1 + R4: chunk 960 @48 kHz mono -> 318 frames from the first chunk, 320 afterwards; emitted as exact 960-sample packets,
    # duration_us = frames*1e6/rate (integer), running timestamp, sequence 0,1,2...
    x = np.sin(np.arange(48000 * 2) * 0.01).astype(np.float32)
    r = minihost.Resampler(16000, 960, 960)
    for i in range(0, x.size, 1920):
        r.push(x[i:i + 1920], 48000, 1, ts=1000 if i == 0 else None)
    r.finish()
    pk = r.packets()
    total = sum(p["samples"].size for p in pk)
    assert total == 318 + 320 * 99
    assert all(p["samples"].size == 960 for p in pk[:-1]) and pk[-1]["samples"].size == total - 960 * (len(pk) - 1)
    assert [p["sequence"] for p in pk] == list(range(len(pk)))
    assert pk[0]["duration_us"] == 60000 and pk[0]["timestamp_us"] == 1000 and pk[1]["timestamp_us"] == 61000
    # sample values: product C++ node == oracle restatement == plain numpy lerp on the delayed index grid
    orc = oracle_lib.OracleResampler(16000 / 48000, 960, 1)
    ref = np.concatenate([orc.process(x[i:i + 960])[0] for i in range(0, x.size, 960)])
    got = np.concatenate([p["samples"] for p in pk])
    assert np.array_equal(got, ref)
    idx = -4.0 + 3.0 * (1 + np.arange(ref.size)); k = np.floor(idx).astype(int); frac = (idx - k).astype(np.float32)
    xp = np.concatenate([np.zeros(16, np.float32), x])
    lerp = (1.0 - frac) * xp[k + 16] + frac * xp[k + 17]
    assert np.allclose(ref, lerp, atol=1e-6)


def test_resampler_passthrough_rechunks_wav_frames(built):
    # config 1: 16 kHz WAV demuxed in 1920-sample frames (wav.rs:34) -> R2 re-chunks to 960-sample packets, samples untouched
    x = np.arange(1920 * 5 + 100, dtype=np.float32)
    r = minihost.Resampler(16000, 960, 960)
    for i in range(0, x.size, 1920):
        r.push(x[i:i + 1920], 16000, 1)
    r.finish()
    pk = r.packets()
    assert np.array_equal(np.concatenate([p["samples"] for p in pk]), x)
    assert [p["samples"].size for p in pk] == [960] * 10 + [100]


def test_resampler_config_validation(built):
    with pytest.raises(ValueError):
        minihost.Resampler(0)
    with pytest.raises(ValueError):
        minihost.Resampler(16000, 960, 1000)
    r = minihost.Resampler(16000, 960, 960)
    r.push(np.zeros(960, np.float32), 48000, 1)
    with pytest.raises(RuntimeError, match="Audio format changed mid-stream"):
        r.push(np.zeros(960, np.float32), 44100, 1)


def test_json_helpers_match_serde_shapes(built):
    L = minihost.lib()
    assert L.mh_json_quote('a"b\\c\n\t\x01é'.encode()).decode() == '"a\\"b\\\\c\\n\\t\\u0001é"'
    assert L.mh_json_f32(0.5) == b"0.5" and L.mh_json_f32(1.0) == b"1.0"
    assert L.mh_utf8_trim(" \t hello world 　".encode()).decode() == "hello world"
    assert L.mh_utf8_valid(b"ok \xe2\x99\xaa", 6) == 1 and L.mh_utf8_valid(b"\xe2\x99", 2) == 0


def test_json_f32_is_the_widened_value_in_ryus_layout(built):
    """`json!({"threshold": self.config.vad_threshold})` holds the f32 widened to f64 (serde_json's Value), and serde_json prints f64 with ryu: shortest round-trip digits, plain
    decimals while the point lies within 16 digits right / 5 zeros left of the first digit, else d[.ddd]e[-]x.  Second implementation: Python's repr() for the digits (also shortest
    round-trip) laid out by those rules."""
    import ctypes as C
    L = minihost.lib(); L.mh_json_f32.restype = C.c_char_p; L.mh_json_f32.argtypes = [C.c_float]

    def ryu64(d):
        if d == 0:
            return "-0.0" if np.signbit(d) else "0.0"
        m, _, e = ("%r" % abs(float(d))).partition("e")
        if e:
            digits = m.replace(".", ""); e10 = int(e)
            if "." in m and m.endswith(".0"):
                digits = m[:-2]
        else:
            ip, _, fp = m.partition("."); fp = "" if fp == "0" else fp
            s = (ip + fp).lstrip("0"); e10 = len(ip.lstrip("0")) - 1 if ip.strip("0") else -(len(fp) - len(fp.lstrip("0")) + 1)
            digits = s
        digits = digits.rstrip("0") or "0"
        length = len(digits); k = e10 - (length - 1); kk = length + k; sign = "-" if d < 0 else ""
        if 0 <= k and kk <= 16:
            return sign + digits + "0" * k + ".0"
        if 0 < kk <= 16:
            return sign + digits[:kk] + "." + digits[kk:]
        if -5 < kk <= 0:
            return sign + "0." + "0" * (-kk) + digits
        return sign + (digits if length == 1 else digits[0] + "." + digits[1:]) + "e" + str(kk - 1)

    assert [L.mh_json_f32(v).decode() for v in (0.5, 1.0, 0.25, 100000.0, 0.0)] == ["0.5", "1.0", "0.25", "100000.0", "0.0"]
    assert L.mh_json_f32(np.float32(0.7)).decode() == "0.699999988079071" and L.mh_json_f32(np.float32(1e-5)).decode() == "9.999999747378752e-6" and L.mh_json_f32(np.float32(0.0001)).decode() == "0.00009999999747378752"
    assert L.mh_json_f32(np.float32(1e-7)).decode() == "1.0000000116860974e-7" and L.mh_json_f32(np.float32(2.0 ** 53)).decode() == "9007199254740992.0"
    assert L.mh_json_f32(np.float32(2.0 ** 60)).decode() == "1.152921504606847e18"
    rng = np.random.default_rng(29)
    vals = list(rng.random(2000).astype(np.float32)) + list(np.exp(rng.uniform(-60, 60, 2000)).astype(np.float32) * rng.choice([-1.0, 1.0], 2000).astype(np.float32))
    for v in vals:
        assert L.mh_json_f32(float(v)).decode() == ryu64(float(np.float32(v))), repr(float(np.float32(v)))
        assert float(L.mh_json_f32(float(v)).decode()) == float(np.float32(v))


def test_json_quote_and_trim_against_pythons_own_on_random_strings(built):
    """serde_json writes a string with the quote and the backslash escaped, backspace / form feed / newline / carriage return / tab in their short forms, the other control
    characters as six-character lower-case hex escapes, and everything else — DEL and all non-ASCII included — as it stands; json.dumps(ensure_ascii=False) follows the same rules,
    so it serves as the second implementation.  str::trim removes Unicode White_Space at both ends — U+0009..000D, 0020, 0085, 00A0, 1680, 2000..200A, 2028, 2029, 202F, 205F,
    3000; NOT U+001C..001F, which Python's own strip() also removes — so the comparison strips that set explicitly.  500 random strings over an alphabet of awkward characters."""
    import json
    L = minihost.lib()
    rng = np.random.default_rng(23)
    alphabet = [chr(c) for c in (0x61, 0x62, 0x20, 0x5a, 0x30, 0x39, 0x22, 0x5c, 0x2f, 0x08, 0x0c, 0x0a, 0x0d, 0x09, 0x01, 0x1f, 0x7f, 0x27, 0x3c, 0x3e, 0x26, 0xe9, 0xdf, 0x4f60, 0x597d,
                                         0x266a, 0xa0, 0x2028, 0x3000, 0xfeff, 0x1f600, 0x20)]
    WS = "".join(chr(c) for c in list(range(0x09, 0x0e)) + [0x20, 0x85, 0xa0, 0x1680] + list(range(0x2000, 0x200b)) + [0x2028, 0x2029, 0x202f, 0x205f, 0x3000])
    alphabet += [chr(0x85), chr(0x2003), chr(0x202f), chr(0x1c)]
    for _ in range(500):
        s = "".join(rng.choice(alphabet) for _ in range(int(rng.integers(0, 40))))
        assert L.mh_json_quote(s.encode()).decode() == json.dumps(s, ensure_ascii=False), repr(s)
        assert L.mh_utf8_trim(s.encode()).decode() == s.strip(WS), repr(s)


def test_binade_stepping_equals_the_sequential_index_walk():
    """The arithmetic k_resample_starts uses to propose every chunk's start index (whole binades of the f64 index per step, round-to-even
    ties included) against rubato's sequential `idx += t_ratio` walk, restated here in Python floats (IEEE f64, same roundings): identical
    output counts and carried indices for every chunk.  On the GPU k_resample_walk repeats this check for each launch."""
    import math
    import struct

    def seq(s, t, chunk):
        end = float(chunk - 9) - math.ceil(t); x = s; n = 0
        while x < end:
            x += t; n += 1
        return n, x - chunk

    def cdiv(a, b):
        q = int(float(a) / float(b))
        while q * b < a:
            q += 1
        while (q - 1) * b >= a:
            q -= 1
        return q

    def prop(s, t, chunk):
        end = float(chunk - 9) - math.ceil(t); x = s; n = 0
        while x < end and x < 4.0:
            x += t; n += 1
        while x < end:
            bits = struct.unpack("<q", struct.pack("<d", x))[0]
            e = ((bits >> 52) & 0x7FF) - 1023
            xi = (bits & 0xFFFFFFFFFFFFF) | (1 << 52)
            ts = math.ldexp(t, 52 - e); a = math.floor(ts); fr = ts - a; ti = int(a)
            if fr == 0.5:
                if xi & 1:
                    x = x + t; n += 1; continue
                ti += ti & 1
            elif fr > 0.5:
                ti += 1
            endi = int(math.ldexp(end, 52 - e))
            k = min(cdiv((1 << 53) - xi, ti) - 1, cdiv(endi - xi, ti) if xi < endi else 0)
            to_end = cdiv(endi - xi, ti) if xi < endi else 0
            xi += k * ti; n += k; x = math.ldexp(float(xi), e - 52)
            if k == to_end:
                break
            x = x + t; n += 1
        return n, x - chunk

    for rate in (44100, 22050, 11025, 48000, 8000, 32000, 96000, 24000, 37800):
        t = 1.0 / (16000 / rate); s = -4.0
        for c in range(800):
            a = seq(s, t, 960)
            assert prop(s, t, 960) == a, (rate, c)
            s = a[1]
