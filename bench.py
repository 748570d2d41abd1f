#!/usr/bin/env python3
"""bench.py — real-time factor of the Whisper-small Oneshot batch path on MI355X (BASELINE.json metric).

One step = one pass of the hot path (log-mel -> encoder -> cross K/V -> batched greedy decode -> segments) over one batch
of 64 synthetic 30 s clips per GPU, PCM already resident in HBM.  N > 1: one process per GPU, clips sharded
c -> rank c mod N (independent clips: no data-path collective), then one RCCL all_gather of the fixed-size token buffers.
`python bench.py --gpus N` with no RANK in the environment starts the N ranks itself (fresh child processes; the parent never
touches the GPU); under torchrun it is one of the ranks.  Rank 0 prints ONE JSON line.

The headline runs the f16-MFMA precision (f16 operands on the f16 matrix cores, f32 accumulate).  Outside the timed region every one of
its greedy decisions on this batch is checked against the exact precision under teacher forcing (streamkit_amd/parity.py; the line carries
steps_checked / argmax_disagreements / max_margin_at_disagreement, and the process exits non-zero when a decision differs away from a
near-tie); the exact mode (f32-chain contractions, bit-identical to the CPU oracle) is timed beside it and reported under "modes", and a
sample of clips is compared with the oracle in the cpu_baseline leg.  `value` is with the PCM resident in HBM (the bench contract);
`value_pcie_inclusive` starts from pinned host memory.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

# /opt/skills/guides/MI355X_MICROARCH.md, chip-level parameters (dense peaks)
F32_MFMA_PEAK_TFLOPS = 157.3
F16_MFMA_PEAK_TFLOPS = 2500.0
HBM_PEAK_GBS = 8000.0


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clips", type=int, default=64, help="clips per GPU per step")
    ap.add_argument("--size", default="small")
    ap.add_argument("--precision", default="f16_mfma", choices=["f16_mfma", "exact"], help="precision of the headline number")
    ap.add_argument("--no-other-mode", action="store_true", help="skip timing the other precision beside the headline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-tts", action="store_true", help="skip the config-5 row: libskw_tts.so at Kokoro-82M's geometry (seeded weights), ~30 s of speech per call")
    ap.add_argument("--no-plugin-path", action="store_true", help="skip the SURVEY 8(d) config-2 leg through libwhisper.so (N plugin instances fed 960-sample packets)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse the N > 1 path on a one-GPU box)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses GPU 0")
    ap.add_argument("--gather", default="torch", choices=["torch", "c_abi"], help="N > 1: the transcript gather through torch.distributed (default) or through libskw_dist.so "
                    "(include/skw_dist.h: the same RCCL all_gather behind the C ABI a non-Python host binds; the 128-byte id travels by a torch.distributed broadcast)")
    return ap.parse_args()


def launch_ranks(args):
    """--gpus N > 1 without a launcher: start N fresh ranks.  This process must not initialise the GPU (a later exec / fork of a
    GPU-initialised process is what the pool forbids), so it only counts devices and relays rank 0's line."""
    import torch
    from streamkit_amd import dist as skd
    if args.gpus > skd.self_started_rank_limit():
        sys.stderr.write("bench.py: --gpus %d without a launcher would start %d rank processes itself; self-started jobs are limited to %d (streamkit_amd/dist.py, the GPU pool's process guard).\n"
                         "Launch the ranks the way the driver does: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d ...\n"
                         % (args.gpus, args.gpus, skd.self_started_rank_limit(), args.gpus, args.gpus))
        return 2
    n_dev = torch.cuda.device_count()          # does not create a HIP context
    if n_dev < args.gpus and not args.share_gpu:
        sys.stderr.write("bench.py: --gpus %d but only %d device(s) visible\n" % (args.gpus, n_dev))
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [p.wait() for p in procs]
    if any(rcs):
        sys.stderr.write("bench.py: rank exit codes %s\n" % rcs)
        sys.stdout.write(out0)
        return 1
    line = [l for l in out0.splitlines() if l.startswith("{")]
    if not line or json.loads(line[-1]).get("n_gpus") != args.gpus:
        sys.stderr.write("bench.py: rank 0 did not report n_gpus == %d\n" % args.gpus)
        return 1
    print(line[-1], flush=True)
    return 0


def plugin_path_leg(model_path, host_pcm, B, precision, key, reps=3):
    """SURVEY.md 8(d) config 2 as written: B concurrent plugin instances (libwhisper.so through the StreamKit native ABI v2, driven by the C++ mini-host that
    replays wrapper.rs's call sequence), each fed its 30 s clip from host memory in 960-sample RawAudio packets and flushed; wall clock from the first packet
    to the last Transcription packet.  Includes packet feeding, the 512-sample VAD framing (vad_mode: always), batch formation across instances, H2D copies,
    JSON building.  Model load is excluded the way the reference excludes it: the process-global context cache (a throw-away first round warms it).
    precision None = the `precision` parameter is NOT sent: what a host that only swaps libwhisper.so gets (the plugin's default, exact)."""
    import ctypes as C
    from streamkit_amd import minihost
    plug = minihost.Plugin()
    pcms = [host_pcm[i] for i in range(B)]
    params = {"model_path": model_path, "vad_mode": "always", "flush_tail": True, "max_batch": B, "batch_window_ms": 40, "suppress_non_speech_tokens": True}
    if precision is not None:
        params["precision"] = precision
    best, n_seg, log0 = None, 0, ""
    for rep in range(reps + 1):
        nodes = [plug.create_node(params) for _ in range(B)]
        if rep == 0:
            log0 = " | ".join(l for l in nodes[0].logs() if "CACHE" in l or "loaded and cached" in l)
        ms = minihost.run_oneshot(nodes, pcms, 960)
        outs = [n.outputs() for n in nodes]
        okp = all(len(o) == 1 and o[0][1] == 3 for o in outs)
        for n in nodes:
            n.destroy()
        if not okp:
            return {"value_" + key: None, key: {"error": "an instance did not emit exactly one Transcription packet: %s" % [len(o) for o in outs]}}
        n_seg = sum(len(json.loads(o[0][2].decode())["segments"]) for o in outs)
        if rep > 0:
            best = ms if best is None else min(best, ms)
    L = C.CDLL(os.path.join(ROOT, "streamkit_amd", "libwhisper.so"))
    loads, hits = C.c_int(), C.c_int()
    L.skw_whisper_plugin_cache_stats(C.byref(loads), C.byref(hits))
    audio_s = sum(p.size for p in pcms) / 16000.0
    return {"value_" + key: round(audio_s / (best * 1e-3), 1),
            key: {"what": "%d libwhisper.so instances (native plugin ABI v2, mini-host), each fed one 30 s clip as 960-sample packets from host memory, then flushed: "
                          "first packet in -> last Transcription JSON out; best of %d rounds" % (B, reps),
                  "wall_ms": round(best, 2), "instances": B, "packet_samples": 960, "batch_window_ms": 40, "vad_mode": "always",
                  "precision_param": precision if precision is not None else "(not sent: the plugin's default, exact)", "model_file": os.path.basename(model_path),
                  "segments": n_seg, "model_loads_in_process": loads.value, "context_cache_hits": hits.value, "first_create_log": log0}}


def quant_leg(f16_path, host_pcm, dev_ptrs, ns, B, device, kind="q5_1"):
    """The reference's DEFAULT model is a q5_1 file (plugins/native/whisper/src/lib.rs:114-116).  The benchmark model re-encoded as whisper.cpp's quantize tool lays it out
    (tools/quantize_ggml.py), run in the exact precision = ggml's own arithmetic for such files (q8 activation blocks, integer block dots, f32 scales): engine level
    (`value_quant_<kind>`, PCM resident in HBM like `value`) and through the plugin with no `precision` parameter (`value_plugin_path_<kind>_default`)."""
    from streamkit_amd import engine
    qpath = f16_path.replace(".bin", "_%s.bin" % kind)
    if not os.path.exists(qpath):
        tmp = "%s.tmp%d" % (qpath, os.getpid())
        subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "quantize_ggml.py"), f16_path, tmp, kind])
        os.replace(tmp, qpath)
    m = engine.Model(qpath, device=device)
    ctx = engine.Context(m, max_batch=B, max_samples=480000)
    ctx.set_precision("exact")
    p = ctx.default_params(); p.suppress_nst = 1
    ctx.full_batch(None, p, device_ptrs=dev_ptrs, n_samples=ns)
    best = None
    for _ in range(2):
        t0 = time.perf_counter(); ctx.full_batch(None, p, device_ptrs=dev_ptrs, n_samples=ns); dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    t = ctx.timing()
    out = {"value_quant_" + kind: round(B * 30.0 / best, 1),
           "quant_" + kind: {"what": "the benchmark model as a %s GGML file (the reference's default file type), exact precision = ggml's arithmetic for block-quantised weights, "
                                     "same %d clips resident in HBM; best of 2" % (kind, B), "ms_per_step": round(best * 1e3, 2), "encode_ms": round(t["encode_ms"], 2),
                             "decode_ms": round(t["decode_ms"], 2), "ggml_type_running": m.quant}}
    ctx.close(); m.close()
    out.update(plugin_path_leg(qpath, host_pcm, B, None, "plugin_path_%s_default" % kind, reps=2))
    return out


def tts_leg(reps=5):
    """BASELINE.json configs[4]'s second half, reported beside the headline (never part of `value`): libskw_tts.so — the published Kokoro-82M architecture in fp32 HIP
    (seeded weights, PARITY UNPINNED) — speaking ~30 s in one call; GPU-event time of the call, best of `reps`."""
    import numpy as np
    from streamkit_amd import tts as skt
    tts = skt.Tts(skt.synth_kokoro_dir("kokoro82m"))
    try:
        ids = np.concatenate([[0], np.random.default_rng(5).integers(1, 60, 300), [0]]).astype(np.int32)
        y, _ = tts.generate(None, 50, 1.0, ids=ids)
        speed = float(np.clip((y.size // 600) / 1200.0, 0.3, 3.0))
        ms = []
        for _ in range(reps):
            y, rate = tts.generate(None, 50, speed, ids=ids); ms.append(tts.last_ms())
        secs = y.size / float(rate)
        return {"tts": {"what": "libskw_tts.so, Kokoro-82M geometry (81.1 M seeded parameters, fp32 on v_mfma_f32_16x16x4_f32), %d tokens -> %d frames in one call; GPU-event time, best of %d"
                                % (ids.size, y.size // 600, reps), "audio_s": round(secs, 2), "gpu_ms": round(min(ms), 2), "x_real_time": round(secs * 1000.0 / min(ms), 1), "parity": "unpinned"}}
    finally:
        tts.close()


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))

    import numpy as np
    import torch
    import torch.distributed as dist
    from streamkit_amd import engine, synth
    from streamkit_amd import dist as skd

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    file_rank = local_rank
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("rank %d: no device %d (%d visible)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    gdev = torch.device("cuda", local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=gdev)
        else:
            dist.init_process_group(args.backend)
        # communicator set-up is lazy: force it now, outside the timed region, whatever --warmup is
        _w = torch.zeros(1, device=gdev if args.backend == "nccl" else "cpu")
        dist.all_reduce(_w)
    cgather = None
    if world > 1 and args.gather == "c_abi":
        idt = torch.zeros(128, dtype=torch.uint8, device=gdev if args.backend == "nccl" else "cpu")
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(skd.CGather.unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        cgather = skd.CGather.rank(bytes(idt.cpu().numpy().tobytes()), rank, world, local_rank)

    # model (random-init weights of the Whisper-small architecture in whisper.cpp's GGML container; not timed)
    tool = os.path.join(ROOT, "tools", "make_synth_model")
    if not os.path.exists(tool):
        subprocess.check_call(["gcc", "-O2", "-o", tool, tool + ".c", "-lm"])
    # one model file per node, written by local rank 0 (eight concurrent 0.49 GB generations would make the first multi-GPU run host-bound before the
    # timed region); the generator is deterministic, the rename is atomic, the other ranks wait for the name to appear
    path = "/tmp/skw_bench_%s_seed1234_v3.bin" % args.size
    if file_rank == 0:
        tmp = "%s.tmp%d" % (path, os.getpid())
        subprocess.check_call([tool, tmp, "--size", args.size, "--seed", "1234"])
        os.replace(tmp, path)
        open(path + ".ready", "w").write(str(os.stat(path).st_mtime_ns))
    if world > 1:
        dist.barrier()          # every rank of this node sees rank 0's file from here on (one node: the driver's launch shape)
    if not os.path.exists(path):
        raise SystemExit("rank %d: model file %s was not written" % (rank, path))
    t_load = time.perf_counter()
    model = engine.Model(path, device=local_rank)          # GGML file -> HBM: weights, kperm / natural-k copies, fragment-order images (outside the timed region, as the
    torch.cuda.synchronize()                               # reference excludes it through its context cache / prewarm: lib.rs:330-374, plugins.rs:265-306)
    model_load_ms = 1000.0 * (time.perf_counter() - t_load)
    hp = model.hp
    B = args.clips
    n_samples = 480000
    t_ctx = time.perf_counter()
    ctx = engine.Context(model, max_batch=B, max_samples=n_samples)
    ctx_create_ms = 1000.0 * (time.perf_counter() - t_ctx)
    params = ctx.default_params()
    params.suppress_nst = 1   # the reference node's default (lib.rs:634, suppress_non_speech_tokens = true)

    # inputs: clip c -> rank c mod world; resident in HBM before the timed region
    clip_ids = skd.shard_clip_ids(B * world, rank, world)
    host = np.stack([synth.clip(c, n_samples) for c in clip_ids])
    dev = torch.from_numpy(host).cuda()
    pinned = torch.from_numpy(host).pin_memory()
    torch.cuda.synchronize()
    ptrs = [dev[i].data_ptr() for i in range(B)]
    ns = [n_samples] * B
    gather_s = [0.0]

    def step(from_host=False):
        if from_host:    # PCIe-inclusive variant: the batch's PCM starts in pinned host memory
            dev.copy_(pinned, non_blocking=True)
            torch.cuda.synchronize()
        res = ctx.full_batch(None, params, device_ptrs=ptrs, n_samples=ns)
        if world > 1:   # the one exchange step: fixed-size int32 token buffers to every rank over RCCL
            t = time.perf_counter()
            if cgather is not None:
                skd.table_from_gathered(cgather.gather([skd.pack_tokens(res)])[0], world)
            else:
                skd.gather_tokens(skd.pack_tokens(res), world, device=gdev if args.backend == "nccl" else None)
            gather_s[0] += time.perf_counter() - t
        return res

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n_warm, n_steps, from_host=False):
        for _ in range(n_warm):
            res = step(from_host)
        barrier()
        gather_s[0] = 0.0
        t0 = time.perf_counter()
        for _ in range(n_steps):
            res = step(from_host)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], dtype=torch.float64, device=gdev if args.backend == "nccl" else "cpu")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, res

    audio_per_step = B * world * (n_samples / 16000.0)
    ctx.set_precision(args.precision)
    dt, res = timed(args.warmup, args.steps)
    gather_ms = 1000.0 * gather_s[0] / args.steps
    timing = ctx.timing()
    value = args.steps * audio_per_step / dt
    modes = {args.precision: {"value": round(value, 2), "ms_per_step": round(1000.0 * dt / args.steps, 3),
                              "encode_ms": round(timing["encode_ms"], 3), "decode_ms": round(timing["decode_ms"], 3), "mel_ms": round(timing["mel_ms"], 3)}}
    dt_h, _ = timed(1, max(2, min(args.steps, 5)), from_host=True)
    pcie_value = max(2, min(args.steps, 5)) * audio_per_step / dt_h

    other = "exact" if args.precision == "f16_mfma" else "f16_mfma"
    identical = None
    if not args.no_other_mode:
        n_o = max(2, min(args.steps, 5))
        ctx.set_precision(other)
        dt_o, res_o = timed(1, n_o)
        t_o = ctx.timing()
        modes[other] = {"value": round(n_o * audio_per_step / dt_o, 2), "ms_per_step": round(1000.0 * dt_o / n_o, 3),
                        "encode_ms": round(t_o["encode_ms"], 3), "decode_ms": round(t_o["decode_ms"], 3), "mel_ms": round(t_o["mel_ms"], 3)}
        same = [[t[0] for t in a["tokens"]] == [t[0] for t in b["tokens"]] and
                [(s["t0"], s["t1"], s["text"]) for s in a["segments"]] == [(s["t0"], s["t1"], s["text"]) for s in b["segments"]] for a, b in zip(res, res_o)]
        # a clip that differs must differ from a near-tie on: the exact mode's top1 - top2 logit margin at the first differing token
        ex = res_o if other == "exact" else res
        fa = res if other == "exact" else res_o
        margins = []
        for a, b, ok in zip(fa, ex, same):
            if not ok:
                ia, ib = [t[0] for t in a["tokens"]], [t[0] for t in b["tokens"]]
                k = next((i for i, (x, y) in enumerate(zip(ia, ib)) if x != y), min(len(ia), len(ib)))
                margins.append(min(b["tokens"][k][4] if k < len(ib) else float("inf"), b["min_margin"]))
        identical = {"identical_clips": int(sum(same)), "of": len(same), "largest_exact_mode_margin_at_a_divergence": round(max(margins), 4) if margins else None,
                     "note": "transcripts may differ only from a near-tie of the greedy argmax on (DESIGN.md section 1; tests/test_gpu_f16.py bounds the margin)"}
        ctx.set_precision(args.precision)

    # every decision of the f16_mfma precision on THIS batch, checked against the exact precision under teacher forcing (streamkit_amd/parity.py)
    parity = None
    if not args.no_other_mode:
        from streamkit_amd.parity import teacher_forced_compare
        try:
            tf = teacher_forced_compare(ctx, None, params, device_ptrs=ptrs, n_samples=ns)
            parity = {k: tf[k] for k in ("steps_checked", "sampled_steps", "sampled_draws_that_differ", "argmax_disagreements", "disagreements_on_exact_runner_up", "max_margin_at_disagreement", "max_logit_err", "logit_err_bound", "margin_bound", "ok")}
            parity["clips_checked"] = len(tf["per_clip"])
            parity["what"] = ("f16_mfma fed the exact precision's tokens: each of its greedy decisions, on every clip and step of this batch, equals the exact one "
                              "or sits where the exact top1 - top2 logit margin is below margin_bound; deciding logits agree within logit_err_bound")
            if identical is not None:
                parity["free_running_identical_clips"] = identical["identical_clips"]; parity["of"] = identical["of"]
        except (AssertionError, RuntimeError) as e:
            # the two precisions' CONTROL FLOW diverged under forcing (a fallback or an EOT decision on a near-tie): the instrument could not run to the end.
            # The metric line is still printed, carrying the failure, and the process exits non-zero after it.
            parity = {"ok": False, "error": "%s: %s" % (type(e).__name__, e), "max_margin_at_disagreement": None, "max_logit_err": None}
        ctx.set_precision(args.precision)

    fast = args.precision == "f16_mfma"
    out = {
        "metric": "real-time factor (audio-sec/wall-sec) Whisper-small Oneshot batch",
        "value": round(value, 2), "unit": "x real-time", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16" if fast else "f32", "data": "synthetic",
        "config": {"workload": "Whisper-%s Oneshot batch, %dx30 s synthetic 16 kHz clips per GPU (configs[1]), greedy T=0 with the temperature ladder armed, "
                               "random-init weights in GGML f16 container" % (args.size, B),
                   "precision": args.precision + (": f16 operands on v_mfma_f32_16x16x32_f16, f32 accumulate" if fast else ": f16 operands widened to f32, k-ordered chains on v_mfma_f32_16x16x4_f32"),
                   "clips_per_gpu": B, "sharding": "clip c -> rank c mod N; one RCCL all_gather of int32 [%d x 226] token buffers" % B,
                   "vad": "none (engine-level full(); the plugin path, 64 libwhisper.so instances fed 960-sample packets, is value_plugin_path)",
                   "identical_clips_f16_vs_exact": ("%d/%d" % (identical["identical_clips"], identical["of"])) if identical else None,
                   "model_load_ms": round(model_load_ms, 1), "ctx_create_ms": round(ctx_create_ms, 1),
                   "last_step": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in timing.items()},
                   "fallback_requested": int(sum(r["fallback_requested"] for r in res))},
        "modes": modes,
        "transcripts_f16_vs_exact": identical,
        "parity_f16_vs_exact_teacher_forced": parity,
        "gather_ms": round(gather_ms, 3) if world > 1 else 0.0,
        "value_pcie_inclusive": round(pcie_value, 2),
        "model_load_ms": round(model_load_ms, 1), "ctx_create_ms": round(ctx_create_ms, 1),
    }

    if rank == 0 and not args.no_roofline:
        # (1) the dominant kernel's launches in the TIMED configuration — step graphs, the engine's row groups on their streams, eight steps enqueued ahead — stamped by the
        # kernel's own clock (include/skw_engine.h skw_ctx_kernel_clock: first wave in -> last wave out, live rows per launch); one graph-capturing step, then the measured one
        kclk = None
        if fast:
            try:
                ctx.kernel_clock(True)
                ctx.full_batch(None, params, device_ptrs=ptrs, n_samples=ns)
                ctx.full_batch(None, params, device_ptrs=ptrs, n_samples=ns)
                kclk = ctx.kernel_clock_get(); kclk["step"] = ctx.timing(); kclk["records"] = ctx.kernel_clock_records()
                ctx.kernel_clock(False)
            except RuntimeError as e:
                kclk = {"error": str(e), "launches": 0}
        # (2) per-kernel-class HIP-event timing on the engine's streams, one extra (untimed, eager) step: the class table and which kernel dominates
        ctx.profile(True)
        ctx.full_batch(None, params, device_ptrs=ptrs, n_samples=ns)
        prof = ctx.profile_get()
        ctx.profile(False)
        tprof = ctx.timing()
        mfma_peak = F16_MFMA_PEAK_TFLOPS if fast else F32_MFMA_PEAK_TFLOPS
        # the dominant KERNEL: k_gemm_smallm is a class of six differently shaped decode products (none of which comes near the
        # cross-attention on its own), so it is listed under "kernels" but does not compete here
        dom = max(((k, v) for k, v in prof.items() if k != "k_gemm_smallm"), key=lambda kv: kv[1]["ms"])
        name, p = dom
        traffic = None   # HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json)
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            pt = pt.get(args.precision, pt).get("per_launch_bytes", {})
            traffic = pt[name]["total"] if name in pt else None
        except Exception:
            traffic = None
        kern = {}
        for k, v in prof.items():
            if not v["count"]:
                continue
            e = {"launches": v["count"], "ms": round(v["ms"], 3)}
            if v["flops"]:
                e["tflops"] = round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2)
            if v["bytes"]:
                e["gbs"] = round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1)
            kern[k] = e
        if name in ("k_gemm", "k_attn_encoder"):
            ach = p["flops"] / (p["ms"] * 1e-3) / 1e12
            pk = mfma_peak
            roof = {"bound": "mfma", "kernel": name, "achieved": round(ach, 3), "peak": pk, "unit": "TFLOP/s", "frac": round(ach / pk, 4)}
        else:
            ach = p["bytes"] / (p["ms"] * 1e-3) / 1e9
            roof = {"bound": "hbm", "kernel": name, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4)}
        roof.update({"traffic": traffic, "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc passes of this command, committed; not re-measured by this run)",
                     "avg_launch_ms": round(p["ms"] / max(1, p["count"]), 4), "launches": p["count"], "kernels": kern})
        if name == "k_dec_cross_attn":
            # the profile class covers the step's cross attention in either precision; the symbol rocprofv3 shows for it:
            roof["symbol"] = ("k_dec_cross_attn16<3,3> (one streaming pass over fragment-order K / V^T, non-temporal loads)" if fast and engine.switches()["XATTN_FRAG"][1] != 0
                              else "k_dec_cross_attn<24,4,3,...> (two-phase kernel over the row layouts)")
            # `achieved` books the bytes of LIVE rows only (a finished sequence's workgroups return at once): sum over launches of 4 x live rows x n_ctx x d,
            # over the summed launch time.  Beside it: one full launch (all rows live) in isolation, back to back over 12 different K / V^T images.
            import ctypes as C
            Lb = engine.lib()
            groups, grows = timing["decode_groups"], timing["decode_group_rows"]      # what the timed run's decode step actually was (skw_timing)
            row_bytes = 4.0 * hp.n_audio_ctx * hp.n_text_state                          # K and V^T of one row, f16
            roof["row_groups"] = groups; roof["rows_per_launch"] = grows
            roof["booking"] = "algorithmic bytes = 4 B x live rows x n_audio_ctx x n_text_state per launch (K and V^T, f16); finished rows' workgroups return at once and are not booked"
            def isolated(rows):      # the same kernel, `rows` rows all live, alone on the GPU, back to back over 12 different K / V^T images: HIP events AND the in-kernel clock on the same launches
                ue, uc = C.c_float(), C.c_float()
                if Lb.skw_debug_xattn(ctx.h, rows, hp.n_text_layer, 0, 240, C.byref(ue), C.byref(uc)) != 0 or ue.value <= 0:
                    return None
                by = row_bytes * rows
                return {"rows": rows, "bytes": by, "us": round(ue.value, 2), "us_in_kernel_clock": round(uc.value, 2) if uc.value > 0 else None,
                        "achieved": round(by / ue.value / 1e3, 1), "unit": "GB/s", "frac": round(by / ue.value / 1e3 / HBM_PEAK_GBS, 4)}
            full = isolated(B)
            if full:
                roof["full_launch"] = full
            if kclk and kclk.get("launches", 0) > 0:
                # in-kernel spans of the timed configuration + the dispatch / completion edges an event-stamped (= rocprofv3) duration adds, measured on isolated launches
                # of the SAME shape with both instruments
                iso = isolated(grows) if grows != B else full
                edge = (iso["us"] - iso["us_in_kernel_clock"]) if iso and iso["us_in_kernel_clock"] else 0.0
                n_l = kclk["launches"]; span = kclk["sum_us"] / n_l; live = kclk["sum_live_rows"] / n_l
                dur = span + edge
                per_launch = row_bytes * live / dur / 1e3
                # launches of different row groups overlap in time and then share HBM: the time the kernel was in flight at all is the UNION of the launch intervals
                rec = kclk["records"][np.argsort(kclk["records"][:, 0])]
                union = 0.0; cur_b, cur_e = rec[0, 0], rec[0, 1]
                for b_, e_, _ in rec[1:]:
                    if b_ > cur_e:
                        union += cur_e - cur_b; cur_b, cur_e = b_, e_
                    else:
                        cur_e = max(cur_e, e_)
                union += cur_e - cur_b
                tot_bytes = row_bytes * float(rec[:, 2].sum())
                agg = tot_bytes / union / 1e3
                # serial launches (one row group): the line's figure is bytes per launch / the duration HIP events and rocprofv3 report (in-kernel span + dispatch edges), so that it
                # follows from the committed kernel-stats CSV; the in-kernel figure stands beside it.  Overlapping launches (row groups on streams): the union of the intervals.
                top = per_launch if groups == 1 else agg
                roof.update({"achieved": round(top, 2), "frac": round(top / HBM_PEAK_GBS, 4), "avg_launch_ms": round(dur * 1e-3, 5), "launches": n_l,
                             "in_flight": {"achieved": round(agg, 2), "frac": round(agg / HBM_PEAK_GBS, 4), "what": "bytes of all launches / time at least one is in flight, in-kernel clock (no dispatch edges)"},
                             "definition": ("achieved = algorithmic bytes per launch (4 B x live rows x n_audio_ctx x n_text_state) / average launch duration, the duration being the kernel's own "
                                            "first-wave-in -> last-wave-out span in the timed, graph-launched configuration plus the dispatch edges (`per_launch`): what HIP events and rocprofv3 report"
                                            if groups == 1 else
                                            "achieved = algorithmic bytes of ALL the kernel's launches of one step / the time at least one of them is in flight (union of the launch intervals, in-kernel "
                                            "clock): with %d row groups on %d streams launches overlap (%.0f %% of the summed launch time) and stretch each other; `per_launch` is the strict per-launch figure"
                                            % (groups, groups, 100.0 * (1.0 - union / max(1e-9, float((rec[:, 1] - rec[:, 0]).sum()))))),
                             "in_flight_us_per_step": round(union, 1), "summed_launch_us_per_step": round(float((rec[:, 1] - rec[:, 0]).sum()), 1), "bytes_per_step": tot_bytes,
                             "per_launch": {"achieved": round(per_launch, 2), "frac": round(per_launch / HBM_PEAK_GBS, 4), "avg_launch_ms": round(dur * 1e-3, 5),
                                            "avg_launch_in_kernel_us": round(span, 3), "dispatch_edges_us": round(edge, 3), "live_rows_per_launch": round(live, 3),
                                            "min_launch_in_kernel_us": round(kclk["min_us"], 2), "max_launch_in_kernel_us": round(kclk["max_us"], 2)},
                             "live_row_fraction": round(live / grows, 4),
                             "clocked_step_ms": {"decode": round(kclk["step"]["decode_ms"], 2), "total": round(kclk["step"]["total_ms"], 2),
                                                 "note": "the step the launches were clocked in: graph-launched like the timed steps (compare modes.%s.decode_ms)" % args.precision},
                             "isolated_same_shape": iso})
                roof["timing"] = ("every launch of the graph-launched decode step (%d row group(s), %d rows per launch) stamped by the kernel itself on the device's %d kHz constant clock: "
                                  "first wave in -> last wave out; per_launch.avg_launch_ms = that span + dispatch_edges_us, the difference between HIP-event (hipExtLaunchKernelGGL begin / end = "
                                  "rocprofv3's duration) and in-kernel timing of isolated launches of the same shape" % (groups, grows, kclk["clock_khz"]))
                roof["profiler_note"] = ("one row group (the default): launches are serial, and per_launch.avg_launch_ms, HIP events and rocprofv3 --kernel-trace --stats of this command "
                                         "(profiles/, <round>_rocprofv3_kernel_stats*.csv; its average includes the ~2 %% of launches whose rows had all finished) agree within 3 %%.  "
                                         "With SKW_DECODE_GROUPS=2 the groups' launches overlap and a profiler serialises the two streams (decode 129 -> 197 ms under rocprofv3, profiles/r05b): "
                                         "its averages then describe launches that never ran, and only the in-kernel clock sees the timed configuration" if groups == 1 else
                                         "rocprofv3 --kernel-trace serialises the row groups' streams (decode 129 -> 197 ms, profiles/r05b): its per-kernel average equals `isolated_same_shape.us`, "
                                         "not a launch of this configuration")
                roof["eager_profile_avg_launch_ms"] = round(p["ms"] / max(1, p["count"]), 4)
            else:
                roof["live_row_fraction"] = round(p["bytes"] / (p["count"] * row_bytes * max(1, grows)), 4) if grows else None
                roof["timing"] = "per launch, HIP events stamped at the kernel's own begin and end (hipExtLaunchKernelGGL) in an eager profiled step"
                if kclk and kclk.get("error"):
                    roof["kernel_clock_error"] = kclk["error"]
            if groups > 1:
                roof["row_groups_note"] = ("the decode step runs as %d row groups on %d streams (SKW_DECODE_GROUPS; the default is one): "
                                           "a launch covers %d rows and shares HBM with the other group's kernels, so `frac` is per launch under that sharing; "
                                           "`full_launch` is one %d-row launch alone" % (groups, groups, grows, B))
        # the two phases and the front end against their own rooflines, from the timed step's GPU-event phase times
        nwin, nsteps, nrow = timing["n_windows"], timing["n_decode_steps"], timing["n_row_steps"]
        d, dt_, nc, L = hp.n_audio_state, hp.n_text_state, hp.n_audio_ctx, hp.n_text_layer
        enc_fl = sum(prof[k]["flops"] for k in ("k_gemm", "k_attn_encoder"))                  # 386.7 GF per window at Whisper-small
        xkv = 2.0 * L * nc * dt_ * 2                                                           # cross K/V bytes per sequence per step (55.3 MB)
        dec_w = 2.0 * (L * (4 * dt_ * dt_ + 4 * dt_ * dt_ + 8 * dt_ * dt_) + hp.n_vocab * dt_)  # decoder weights streamed per step (306 MB)
        dec_bytes = nrow * xkv + nsteps * dec_w                                                  # cross K/V only for the steps a row was live in
        fe_bytes = nwin * (n_samples * 4 + 2 * nc * hp.n_mels * 4.0)
        t_enc, t_dec, t_fe = timing["encode_ms"] * 1e-3, timing["decode_ms"] * 1e-3, timing["mel_ms"] * 1e-3
        ph = {"encode": {"bound": "mfma", "achieved": round(enc_fl / t_enc / 1e12, 2), "peak": mfma_peak, "unit": "TFLOP/s", "ms": round(t_enc * 1e3, 3)},
              "decode": {"bound": "hbm", "achieved": round(dec_bytes / t_dec / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "ms": round(t_dec * 1e3, 3),
                         "bytes": "row-steps x %.1f MB cross K/V + steps x %.0f MB weights" % (xkv / 1e6, dec_w / 1e6), "steps": nsteps, "row_steps": nrow,
                         "live_row_fraction": round(nrow / max(1.0, float(nsteps) * B), 4)},
              "front_end": {"bound": "hbm", "achieved": round(fe_bytes / t_fe / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "ms": round(t_fe * 1e3, 3)}}
        for v in ph.values():
            v["frac"] = round(v["achieved"] / v["peak"], 4)
        t_roof = enc_fl / (mfma_peak * 1e12) + dec_bytes / (HBM_PEAK_GBS * 1e9) + fe_bytes / (HBM_PEAK_GBS * 1e9)
        ph["combined"] = {"roofline_ms": round(t_roof * 1e3, 3), "measured_ms": round(timing["total_ms"], 3), "frac": round(t_roof * 1e3 / timing["total_ms"], 4)}
        roof["phases"] = ph
        roof["profiled_step_ms"] = {"encode": round(tprof["encode_ms"], 2), "decode": round(tprof["decode_ms"], 2),
                                    "note": "the step the per-kernel event pairs were taken in (eager launches, an event pair per kernel): its wall time is not the benchmark's"}
        out["roofline"] = roof

    if rank == 0 and world == 1 and not args.no_plugin_path:
        # the drop-in boundary, driver-run: (i) the headline precision, (ii) what a host gets that only swaps the .so (no `precision` parameter -> exact, the precision
        # that is bit-identical to the CPU oracle), (iii) the same on the reference's default file type (q5_1)
        out.update(plugin_path_leg(path, host, B, args.precision, "plugin_path"))
        out.update(plugin_path_leg(path, host, B, None, "plugin_path_default", reps=2))
        try:
            out.update(quant_leg(path, host, ptrs, ns, B, local_rank))
        except Exception as e:      # a side row must not cost the headline line
            out["quant_q5_1"] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["config"]["drop_in_defaults"] = ("a host that only replaces libwhisper.so sends no `precision` parameter and gets the exact precision (the one bit-identical to oracle/): "
                                             "value_plugin_path_default on an f16 file, value_plugin_path_q5_1_default on the reference's default file type; "
                                             "value / value_plugin_path opt into precision=f16_mfma (teacher-forced tolerance mode)")

    if rank == 0 and world == 1 and not args.no_tts:
        try:
            out.update(tts_leg())
        except Exception as e:      # a side row must not cost the headline line
            out["tts"] = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the oracle (a port) on the host cores, bounded sample of the same workload
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_lib import OracleModel
        om = OracleModel(path)
        threads = int(os.environ["OMP_NUM_THREADS"])
        n_cpu = 2
        ok = True
        op = om.default_params()
        op.suppress_nst = 1
        t0 = time.perf_counter()
        for i in range(n_cpu):
            ro = om.full(host[i], op)
            ok = ok and [t[0] for t in ro["tokens"]] == [t[0] for t in res[i]["tokens"]]
        cdt = time.perf_counter() - t0
        op.n_threads = 4    # the reference node's default n_threads = min(4, cores) (lib.rs:92, 139)
        t0 = time.perf_counter()
        om.full(host[2], op)
        cdt4 = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(n_cpu * 30.0 / cdt, 3), "unit": "x real-time", "cores": threads, "kind": "port",
                               "sample": "%d of the %d clips of this workload through oracle/ (OpenMP, %d threads); tokens %s the GPU's (%s mode)"
                                         % (n_cpu, B, threads, "identical to" if ok else "DIFFER from", args.precision),
                               "value_n_threads_4": round(30.0 / cdt4, 3)}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
    if parity is not None and not parity["ok"]:
        sys.stderr.write("bench.py: the f16_mfma precision left the exact precision away from a near-tie (max margin %s, max logit error %s)\n"
                         % (parity["max_margin_at_disagreement"], parity["max_logit_err"]))
        sys.exit(3)


if __name__ == "__main__":
    main()
