"""Step-by-step comparison of the two precisions of the engine under teacher forcing (include/skw_engine.h, skw_full_batch_traced).

The exact precision is the one a CPU restates bit for bit (tests/test_gpu_parity.py holds it to oracle/).  The f16_mfma precision sums its
contractions in the matrix cores' order, so a free-running transcript can leave the exact one's at a near-tie of the greedy argmax and is then
compared with nothing.  Here the f16_mfma decoder is FED the exact run's tokens, so every one of its decisions is made on the history the exact run
had, and each is checked: same argmax, or the exact run's own top1 - top2 margin at that step is below MARGIN_BOUND; and the logits that decide
(the fed token's, the winner's) agree within LOGIT_ERR_BOUND.  Used by tests/test_gpu_f16.py on every clip of BASELINE.json configs[1] and by
bench.py, which prints the counts beside the headline number.  No oracle and no CPU path is involved: both runs are the HIP engine.
"""
import numpy as np

# Logit units, set from what is measured (bounds no looser than the evidence needs).
# Benchmark model (Whisper-small synthetic, logits span ~407): under teacher forcing on the 64 x 30 s batch, 6 473 decisions, the deciding logits of the two
# precisions differ by at most 0.217 - 0.264 across rounds 3's builds (6.5e-4 of the range: eleven f16 roundings per layer through 12 layers), and the 21 - 23
# decisions that differ all sit where the exact mode's own top1 - top2 margin is <= 0.12 and pick the exact mode's runner-up.  A flip needs
# margin <= err(top1) + err(top2); the margin bound sits just above the largest margin ever observed at a flip, well below twice the logit bound.
LOGIT_ERR_BOUND = 0.30
MARGIN_BOUND = 0.15
# The tiny / micro synthetic models' logits span ~630 (1.5 x the benchmark model's): 0.25 - 0.30 measured on ragged multi-window batches with temperature passes,
# <= 0.33 where the prompt pass uses the multi-query cross attention.  Same relative error; the absolute bound scales with the range.
# Round 5's hunt over random clips, levels, batch compositions and decode parameters (tests/hunt/fuzz_parity.py f16, profiles/r05v): benchmark geometry 17 692 decisions, 48 differ
# (every one inside MARGIN_BOUND), largest logit difference 0.238 — the 0.30 stands; tiny 8 168 decisions, 15 differ, 0.363 (clipped and near-silent clips reach what the
# benchmark-like ones did not) -> 0.40; the two-layer d = 128 "micro" models 19 703 decisions, 27 differ at margins <= 0.02, 0.61 - 0.63 -> 0.70 (their logits span more, and 128
# channels average less of the f16 rounding away).
SMALL_MODEL_LOGIT_ERR_BOUND = 0.40
MICRO_MODEL_LOGIT_ERR_BOUND = 0.70


def bounds_for(hp):
    """(logit_err_bound, margin_bound) for a model of these hyper-parameters: the benchmark geometry (d >= 768), the small test models, the d < 256 micro ones"""
    d = hp.n_text_state
    return (LOGIT_ERR_BOUND if d >= 768 else SMALL_MODEL_LOGIT_ERR_BOUND if d >= 256 else MICRO_MODEL_LOGIT_ERR_BOUND), MARGIN_BOUND


def teacher_forced_compare(ctx, clips=None, params=None, device_ptrs=None, n_samples=None, logit_err_bound=None, margin_bound=None):
    """Runs the batch in the exact precision (free, traced), then in f16_mfma fed with the exact run's decisions.  Returns a dict of counts and
    the two result lists; leaves the context in the precision it was in.  The bounds are in logit units and belong to a model's logit scale (the
    defaults: bounds_for(model)); a model with another scale passes its own (tests/test_gpu_f16.py: 7e-4 of the range)."""
    eb, mb = bounds_for(ctx.model.hp)
    logit_err_bound = eb if logit_err_bound is None else logit_err_bound
    margin_bound = mb if margin_bound is None else margin_bound
    was = ctx.get_precision()
    kw = dict(device_ptrs=device_ptrs, n_samples=n_samples)
    ctx.set_precision("exact")
    res_e, tr_e = ctx.full_batch(clips, params, trace=True, **kw)
    ctx.set_precision("f16_mfma")
    res_f, tr_f = ctx.full_batch(clips, params, forced=[t["chosen_id"] for t in tr_e], **kw)
    ctx.set_precision(was)
    steps = disagree = 0
    worst_margin = 0.0          # largest exact-mode margin at a step where the f16_mfma argmax differs
    worst_err = 0.0             # largest |logit difference| seen (fed token; winner when both modes agree on it)
    runner_up = 0               # disagreements where f16_mfma chose the exact mode's runner-up
    n_sampled = draws_differ = 0   # decisions inside temperature passes; of those, draws that came out differently (reported, not an argmax disagreement)
    per_clip = []
    for c, (a, b) in enumerate(zip(tr_e, tr_f)):
        if len(a) != len(b) or not np.array_equal(a["chosen_id"], b["forced_id"]) or not np.array_equal(a["chosen_id"], a["forced_id"]):
            raise AssertionError("clip %d: the forced run did not follow the exact run's decisions (%d vs %d steps)" % (c, len(a), len(b)))
        steps += len(a)
        # decisions of a temperature pass (t > 0) are draws from the distribution, not argmaxes, and their logits are the row's divided by t: what is
        # compared there is the argmax of the admissible logits (top1_id) and the logits scaled back by t
        sampled = a["temperature"] > 0
        n_sampled += int(sampled.sum())
        scale = np.where(sampled, a["temperature"], 1.0).astype(np.float64)
        err = np.abs(a["forced_logit"].astype(np.float64) - b["forced_logit"]) * scale
        same_top = a["top1_id"] == b["top1_id"]
        err_top = (np.abs(a["top1"].astype(np.float64) - b["top1"]) * scale)[same_top]
        e = float(max(err.max() if len(err) else 0.0, err_top.max() if len(err_top) else 0.0))
        worst_err = max(worst_err, e)
        pick_a = np.where(sampled, a["top1_id"], a["chosen_id"]); pick_b = np.where(sampled, b["top1_id"], b["chosen_id"])
        diff = np.nonzero(pick_a != pick_b)[0]
        m = ((a["top1"][diff] - a["top2"][diff]) * scale[diff]).astype(np.float64)
        disagree += len(diff)
        draws_differ += int(np.sum(sampled & (a["chosen_id"] != b["chosen_id"])))
        runner_up += int(np.sum(pick_b[diff] == a["top2_id"][diff]))
        if len(diff):
            worst_margin = max(worst_margin, float(m.max()))
        per_clip.append(dict(steps=len(a), disagreements=len(diff), max_logit_err=e, margins=[round(float(x), 5) for x in m]))
    return dict(steps_checked=steps, sampled_steps=n_sampled, sampled_draws_that_differ=draws_differ, argmax_disagreements=disagree, disagreements_on_exact_runner_up=runner_up,
                max_margin_at_disagreement=worst_margin if disagree else None, max_logit_err=worst_err,
                logit_err_bound=logit_err_bound, margin_bound=margin_bound,
                ok=bool(worst_err <= logit_err_bound and (not disagree or worst_margin < margin_bound)),
                per_clip=per_clip, results_exact=res_e, results_forced=res_f, traces_exact=tr_e, traces_forced=tr_f)
