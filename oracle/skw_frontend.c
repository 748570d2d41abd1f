/*
 * oracle/skw_frontend.c — TEST INFRASTRUCTURE (see skw_oracle.h: PARITY UNPINNED for arithmetic; the integer
 * behaviour restated here IS pinned by the reference source itself, SURVEY.md §8c last row).
 *
 * Plain-C restatement of the Whisper node's segmentation state machine
 * (/root/reference/plugins/native/whisper/src/lib.rs:404-494 and :582-612), driven by a precomputed
 * per-frame speech probability so that it does not depend on any VAD model.
 */
#include "skw_oracle.h"
#include <stdlib.h>
#include <string.h>

/* one cut: [start_ms, end_ms], number of samples handed to Whisper, reason 0 = max_duration, 1 = silence,
 * silence_ms (-1 when absent), the frame index at which it fired, and the segment counter */
int skwo_segment_sim(const float* prob, int n_frames, float threshold, uint64_t min_silence_ms, float max_secs,
                     int64_t* cuts /* [max_cuts][6] */, int max_cuts) {
    uint64_t absolute_time_ms = 0, segment_start_time_ms = 0, segment_counter = 0;
    size_t silence_frame_count = 0; const size_t silence_threshold_frames = (size_t)(min_silence_ms / 32);   /* lib.rs:386 */
    size_t speech_samples = 0; int n_cuts = 0;
    for (int f = 0; f < n_frames; ++f) {
        const int is_speech = prob[f] >= threshold;                                                          /* lib.rs:421 */
        if (is_speech) {
            silence_frame_count = 0;
            if (speech_samples == 0) { segment_start_time_ms = absolute_time_ms; segment_counter++; }        /* lib.rs:428-434 */
            speech_samples += 512;                                                                           /* lib.rs:453 */
            const uint64_t segment_duration_ms = absolute_time_ms - segment_start_time_ms;
            const uint64_t max_duration_ms = (uint64_t)(max_secs * 1000.0f);                                 /* lib.rs:460-461 */
            if (segment_duration_ms >= max_duration_ms) {
                if (n_cuts < max_cuts) { int64_t* c = cuts + 6 * n_cuts; c[0] = (int64_t)segment_start_time_ms;
                c[1] = (int64_t)(absolute_time_ms + 32); c[2] = (int64_t)speech_samples; c[3] = 0; c[4] = -1; c[5] = f; }
                n_cuts++; speech_samples = 0; silence_frame_count = 0;                                       /* lib.rs:463-464, 612, 699 */
            }
        } else {
            silence_frame_count += 1;
            if (speech_samples > 0 && silence_frame_count >= silence_threshold_frames) {                     /* lib.rs:471-472 */
                const uint64_t silence_frames = silence_frame_count > 0 ? silence_frame_count - 1 : 0;
                const uint64_t back = silence_frames * 32;
                const uint64_t end_time_ms = absolute_time_ms >= back ? absolute_time_ms - back : 0;         /* lib.rs:474-476 */
                if (n_cuts < max_cuts) { int64_t* c = cuts + 6 * n_cuts; c[0] = (int64_t)segment_start_time_ms;
                c[1] = (int64_t)end_time_ms; c[2] = (int64_t)speech_samples; c[3] = 1; c[4] = (int64_t)(silence_frame_count * 32); c[5] = f; }
                n_cuts++; speech_samples = 0; silence_frame_count = 0;
            }
        }
        absolute_time_ms += 32;                                                                              /* lib.rs:487 */
    }
    return n_cuts;
}
