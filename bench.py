#!/usr/bin/env python3
"""bench.py — real-time factor of the Whisper-small Oneshot batch path on MI355X (BASELINE.json metric).

One step = one pass of the hot path (log-mel -> encoder -> cross K/V -> batched greedy decode -> segments) over one batch
of 64 synthetic 30 s clips per GPU, PCM already resident in HBM.  N > 1: one process per GPU (torchrun), clips sharded
c -> rank c mod N (independent clips: no data-path collective), then one RCCL all_gather of the fixed-size token buffers.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, os.cpu_count() or 1)))

F32_MFMA_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: Peak FP32 (matrix), dense
HBM_PEAK_GBS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--clips", type=int, default=64, help="clips per GPU per step")
    ap.add_argument("--size", default="small")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for N > 1 (nccl = RCCL over xGMI; gloo only to rehearse the N > 1 path on a one-GPU box)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses GPU 0")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist
    from streamkit_amd import engine, synth

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    file_rank = local_rank
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE %d != --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU fallback")
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
        # communicator set-up is lazy: force it now, outside the timed region, whatever --warmup is
        _w = torch.zeros(1, device=torch.device("cuda", local_rank) if args.backend == "nccl" else "cpu")
        dist.all_reduce(_w)

    # model (random-init weights of the Whisper-small architecture in whisper.cpp's GGML container; not timed)
    tool = os.path.join(ROOT, "tools", "make_synth_model")
    if not os.path.exists(tool):
        subprocess.check_call(["gcc", "-O2", "-o", tool, tool + ".c", "-lm"])
    path = "/tmp/skw_bench_%s_r%d.bin" % (args.size, file_rank)
    subprocess.check_call([tool, path, "--size", args.size, "--seed", "1234"])
    model = engine.Model(path, device=local_rank)
    B = args.clips
    n_samples = 480000
    ctx = engine.Context(model, max_batch=B, max_samples=n_samples)
    params = ctx.default_params()
    params.suppress_nst = 1   # the reference node's default (lib.rs:634, suppress_non_speech_tokens = true)

    # inputs: clip c -> rank c mod world; resident in HBM before the timed region
    from streamkit_amd.dist import shard_clip_ids
    clip_ids = shard_clip_ids(B * world, rank, world)
    host = np.stack([synth.clip(c, n_samples) for c in clip_ids])
    dev = torch.from_numpy(host).cuda()
    torch.cuda.synchronize()
    ptrs = [dev[i].data_ptr() for i in range(B)]
    ns = [n_samples] * B
    from streamkit_amd import dist as skd

    def step():
        res = ctx.full_batch(None, params, device_ptrs=ptrs, n_samples=ns)
        if world > 1:   # the one exchange step: fixed-size int32 token buffers to every rank over RCCL
            skd.gather_tokens(skd.pack_tokens(res), world, device=torch.device("cuda", local_rank) if args.backend == "nccl" else None)
        return res

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        res = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    timing = ctx.timing()
    audio_s = args.steps * B * world * (n_samples / 16000.0)
    value = audio_s / dt

    out = {
        "metric": "real-time factor (audio-sec/wall-sec) Whisper-small Oneshot batch",
        "value": round(value, 2), "unit": "x real-time", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1000.0 * dt / args.steps, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "Whisper-%s Oneshot batch, %dx30 s synthetic 16 kHz clips per GPU (configs[1]), greedy T=0, "
                               "random-init weights in GGML f16 container" % (args.size, B),
                   "clips_per_gpu": B, "sharding": "clip c -> rank c mod N; one RCCL all_gather of int32 [64 x 226] token buffers",
                   "vad": "none (engine-level full(); plugin path uses AlwaysSpeech when no Silero model is present)",
                   "last_step": {k: (round(v, 3) if isinstance(v, float) else v) for k, v in timing.items()},
                   "fallback_requested": int(sum(r["fallback_requested"] for r in res))},
    }

    if rank == 0 and not args.no_roofline:
        # per-kernel-class HIP-event timing on the engine's stream, one extra (untimed) step
        ctx.profile(True)
        ctx.full_batch(None, params, device_ptrs=ptrs, n_samples=ns)
        prof = ctx.profile_get()
        ctx.profile(False)
        dom = max(prof.items(), key=lambda kv: kv[1]["ms"])
        name, p = dom
        traffic = None   # HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json)
        try:
            pt = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["per_launch_bytes"]
            traffic = pt[name]["total"] if name in pt else None
        except Exception:
            traffic = None
        kern = {k: {"launches": v["count"], "ms": round(v["ms"], 3)} for k, v in prof.items() if v["count"]}
        if name in ("k_gemm", "k_gemm_smallm", "k_attn_encoder"):
            ach = p["flops"] / (p["ms"] * 1e-3) / 1e12
            out["roofline"] = {"bound": "mfma", "kernel": name, "achieved": round(ach, 3), "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                               "frac": round(ach / F32_MFMA_PEAK_TFLOPS, 4), "traffic": traffic,
                               "avg_launch_ms": round(p["ms"] / max(1, p["count"]), 4), "launches": p["count"], "kernels": kern}
        else:
            ach = p["bytes"] / (p["ms"] * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": name, "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic,
                               "avg_launch_ms": round(p["ms"] / max(1, p["count"]), 4), "launches": p["count"], "kernels": kern}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # CPU baseline: the oracle (a port) on the host cores, bounded sample of the same workload
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_lib import OracleModel
        om = OracleModel(path)
        threads = int(os.environ["OMP_NUM_THREADS"])
        n_cpu = 2
        t0 = time.perf_counter()
        ok = True
        op = om.default_params()
        op.suppress_nst = 1
        for i in range(n_cpu):
            ro = om.full(host[i], op)
            ok = ok and [t[0] for t in ro["tokens"]] == [t[0] for t in res[i]["tokens"]]
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": round(n_cpu * 30.0 / cdt, 3), "unit": "x real-time", "cores": threads, "kind": "port",
                               "sample": "%d of the %d clips of this workload through oracle/ (OpenMP, %d threads); tokens %s the GPU's"
                                         % (n_cpu, B, threads, "identical to" if ok else "DIFFER from")}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
