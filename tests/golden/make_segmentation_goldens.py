"""Goldens for the segmentation arithmetic, derived BY HAND from the reference source alone
(/root/reference/plugins/native/whisper/src/lib.rs:386, 404-494) — no implementation of ours is run to produce them.
Each cut is [start_ms, end_ms, n_samples, reason(0 max_duration / 1 silence), silence_ms or -1]."""
import json, os

cases = []
# 1. default params, continuous speech: the forced cut fires on the frame where 32*i - start >= 30000, i.e. i = 938
#    (0-based), after that frame was appended: 939 frames, end = 938*32 + 32 = 30048.
cases.append(dict(name="forced cut at 30 s holds 939 frames", n_frames=1000, speech_runs=[[0, 1000]],
                  cuts=[[0, 30048, 939 * 512, 0, -1]]))
# 2. 100 speech frames then silence, min_silence 700 ms -> threshold 21 frames; fires on the 21st silent frame
#    (frame index 120, abs = 3840): end = 3840 - 20*32 = 3200 = exactly the end of speech; silence_ms = 21*32 = 672.
cases.append(dict(name="silence cut after 21 frames", n_frames=200, speech_runs=[[0, 100]],
                  cuts=[[0, 3200, 100 * 512, 1, 672]]))
# 3. speech starting at frame 10 (start_ms 320), 50 frames, then silence.
cases.append(dict(name="segment start time is absolute", n_frames=200, speech_runs=[[10, 60]],
                  cuts=[[320, 1920, 50 * 512, 1, 672]]))
# 4. a short pause (10 frames < 21) does not cut and is NOT buffered: only speech frames are appended (lib.rs:453)
cases.append(dict(name="short pause is skipped, not buffered", n_frames=300, speech_runs=[[0, 40], [50, 90]],
                  cuts=[[0, 90 * 32, 80 * 512, 1, 672]]))
# 5. min_silence_duration_ms = 100 -> 3 frames; two segments, counter increments
cases.append(dict(name="min_silence 100 ms -> 3 frames", n_frames=100, min_silence_ms=100, speech_runs=[[0, 10], [20, 30]],
                  cuts=[[0, 320, 10 * 512, 1, 96], [640, 960, 10 * 512, 1, 96]]))
# 6. trailing speech not followed by enough silence is never cut (no flush in the reference)
cases.append(dict(name="tail is dropped", n_frames=120, speech_runs=[[0, 110]], cuts=[]))
# 7. max_segment_duration_secs = 5.0 -> 5000 ms: fires when 32*i >= 5000 -> i = 157 -> 158 frames, end = 157*32+32 = 5056
cases.append(dict(name="max 5 s", n_frames=400, max_secs=5.0, speech_runs=[[0, 400]],
                  cuts=[[0, 5056, 158 * 512, 0, -1], [5056, 10112, 158 * 512, 0, -1]]))
# 8. threshold semantics: probability >= threshold is speech (lib.rs:421): covered by the runs being exactly 1.0 / 0.0
json.dump(dict(source="/root/reference/plugins/native/whisper/src/lib.rs:386,404-494 (derived by hand)", cases=cases),
          open(os.path.join(os.path.dirname(__file__), "segmentation_goldens.json"), "w"), indent=1)
