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


class CGather:
    """include/skw_dist.h through ctypes: the same gather as gather_tokens, as the C ABI a non-Python host binds (libskw_dist.so, RCCL directly).
    CGather.local(devices): one process driving every listed GPU (ncclCommInitAll); CGather.rank(id_bytes, rank, world, device): one process per GPU."""
    _L = None

    @classmethod
    def lib(cls):
        import ctypes as C
        if cls._L is None:
            L = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libskw_dist.so"))
            L.skw_dist_create_local.restype = C.c_void_p; L.skw_dist_create_local.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_char_p, C.c_size_t]
            L.skw_dist_unique_id.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
            L.skw_dist_create_rank.restype = C.c_void_p; L.skw_dist_create_rank.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_size_t]
            L.skw_dist_world.argtypes = [C.c_void_p]; L.skw_dist_n_local.argtypes = [C.c_void_p]
            L.skw_dist_all_gather_tokens.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_int, C.POINTER(C.c_void_p)]
            L.skw_dist_last_error.restype = C.c_char_p; L.skw_dist_last_error.argtypes = [C.c_void_p]
            L.skw_dist_free.argtypes = [C.c_void_p]
            cls._L = L
        return cls._L

    def __init__(self, handle):
        self.h = handle

    @classmethod
    def local(cls, devices):
        import ctypes as C
        err = C.create_string_buffer(512); dv = (C.c_int * len(devices))(*devices)
        h = cls.lib().skw_dist_create_local(dv, len(devices), err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return cls(h)

    @classmethod
    def unique_id(cls):
        import ctypes as C
        err = C.create_string_buffer(512); buf = C.create_string_buffer(128)
        if cls.lib().skw_dist_unique_id(buf, err, 512) != 0:
            raise RuntimeError(err.value.decode())
        return buf.raw

    @classmethod
    def rank(cls, id_bytes, rank, world, device):
        import ctypes as C
        err = C.create_string_buffer(512)
        h = cls.lib().skw_dist_create_rank(id_bytes, rank, world, device, err, 512)
        if not h:
            raise RuntimeError(err.value.decode())
        return cls(h)

    def gather(self, rows_per_local_rank):
        """rows_per_local_rank: one int32 [n, 226] array per local rank -> one [world * n, 226] array per local rank (rank-major)"""
        import ctypes as C
        L = self.lib(); nl = L.skw_dist_n_local(self.h); world = L.skw_dist_world(self.h)
        assert len(rows_per_local_rank) == nl
        send = [np.ascontiguousarray(r, np.int32) for r in rows_per_local_rank]; n = send[0].shape[0]
        assert all(s.shape == (n, ROW) for s in send)
        recv = [np.empty((world * n, ROW), np.int32) for _ in range(nl)]
        sp = (C.c_void_p * nl)(*[s.ctypes.data for s in send]); rp = (C.c_void_p * nl)(*[r.ctypes.data for r in recv])
        if L.skw_dist_all_gather_tokens(self.h, sp, n, rp) != 0:
            raise RuntimeError(L.skw_dist_last_error(self.h).decode())
        return recv

    def close(self):
        if self.h:
            self.lib().skw_dist_free(self.h); self.h = None


def table_from_gathered(rows, world):
    """[world * n, 226] rank-major rows -> {clip_id: entry} (inverse of shard_clip_ids for equal shard sizes)"""
    n = rows.shape[0] // world
    out = {}
    for r in range(world):
        for j, e in enumerate(unpack_tokens(rows[r * n:(r + 1) * n])):
            out[j * world + r] = e
    return out
