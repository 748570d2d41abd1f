"""Multi-GPU sharding of the Oneshot batch path: clips are independent, so the only exchange is one
gather of fixed-size int32 token buffers (count, n_segments, token ids padded with -1) to every rank."""
import os

import numpy as np

TOKENS_PER_CLIP = 224
ROW = 2 + TOKENS_PER_CLIP

# ONE limit for every place in this repository that STARTS rank processes itself (bench.py's launch_ranks when no launcher set RANK, tests/test_gpu_dist.py):
# the GPU pool these run on allows at most six of a user's processes on its cards at once, so a self-started job never has more ranks than that.  Ranks started
# by a launcher (the driver's `python -m torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`) are the launcher's to count: bench.py then is ONE of the
# ranks and starts nothing.  SKW_SELF_STARTED_RANK_LIMIT overrides it on a node without that guard.
SELF_STARTED_RANK_LIMIT = 6


def self_started_rank_limit():
    return int(os.environ.get("SKW_SELF_STARTED_RANK_LIMIT", SELF_STARTED_RANK_LIMIT))


def shard_clip_ids(n_total, rank, world):
    """clip c -> rank c mod world (same per-clip seeds => results independent of the GPU count)."""
    return [c for c in range(n_total) if c % world == rank]


def pack_tokens(results):
    """results: list of dicts with 'tokens' [(id, ...)] and 'segments' -> int32 [n, 226]."""
    rows = np.full((len(results), ROW), -1, dtype=np.int32)
    for i, r in enumerate(results):
        ids = [t[0] for t in r["tokens"]][:TOKENS_PER_CLIP]
        rows[i, 0] = len(ids)
        rows[i, 1] = len(r["segments"])
        rows[i, 2:2 + len(ids)] = ids
    return rows


def unpack_tokens(rows):
    return [dict(n_segments=int(r[1]), ids=[int(x) for x in r[2:2 + int(r[0])]]) for r in rows]


def gather_tokens(local_rows, world, device=None):
    """all_gather of the per-rank buffers (RCCL when the tensors live on the GPU, gloo on CPU). Returns {clip_id: entry}."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(local_rows))
    if device is not None:
        t = t.to(device)
    if world == 1:
        parts = [t]
    else:
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(parts, t)
    out = {}
    n = local_rows.shape[0]
    for r, p in enumerate(parts):
        for j, e in enumerate(unpack_tokens(p.cpu().numpy())):
            out[j * world + r] = e          # inverse of shard_clip_ids for equal shard sizes
    assert len(out) == n * world
    return out
