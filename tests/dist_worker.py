"""One rank of the Oneshot multi-GPU path (BASELINE configs[2] shape): its shard of the clips through the engine, then the one
all_gather of int32 token buffers; writes what THIS rank holds afterwards to --out.  Started as a fresh process per rank by
tests/test_gpu_dist.py (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* in the environment, as bench.py's launcher and torchrun set them)."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", required=True); ap.add_argument("--clips-per-rank", type=int, default=8); ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--backend", default="nccl"); ap.add_argument("--share-gpu", action="store_true"); ap.add_argument("--precision", default="exact"); ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import torch
    import torch.distributed as dist
    from streamkit_amd import engine, synth
    from streamkit_amd import dist as skd
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    dev = 0 if a.share_gpu else local
    torch.cuda.set_device(dev)
    if world > 1:
        if a.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(a.backend)
    m = engine.Model(a.model, device=dev)
    ctx = engine.Context(m, max_batch=a.clips_per_rank, max_samples=int(16000 * a.seconds))
    ctx.set_precision(a.precision)
    ids = skd.shard_clip_ids(a.clips_per_rank * world, rank, world)
    p = ctx.default_params(); p.suppress_nst = 1
    res = ctx.full_batch([synth.clip(c, int(16000 * a.seconds)) for c in ids], p)
    rows = skd.pack_tokens(res)
    table = skd.gather_tokens(rows, world, device=torch.device("cuda", dev) if (a.backend == "nccl" and world > 1) else None)
    json.dump({"rank": rank, "device": dev, "clip_ids": ids, "gather_shape": list(rows.shape), "gather_dtype": str(rows.dtype), "table": {str(k): v for k, v in table.items()}}, open(a.out, "w"))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
