"""Query-sharded brute-force matcher: one process per GPU, one RCCL all-gather of the per-shard best matches.

BASELINE.json north_star: "Shard only the brute-force descriptor match (query-split) across the 8 GPUs of one node with
an RCCL all-gather of per-shard best matches over xGMI; detection and BA stay single-GPU."  The reference has no
distributed code at all; the call it shards is FeatureMatcher.match_features' knnMatch
(reference src/v2/frame.py:23).

Partition: rank r owns queries [r*ceil(Q/W), (r+1)*ceil(Q/W)); the train set is replicated, so every
rank produces final (global train index, distance) pairs for its queries and no cross-rank tie-breaking exists.  The
exchange is a single all-gather of 16 bytes per query (idx[2], dist[2] as int32), padded to equal shard sizes.
At world_size 1 the same code path runs without a collective.
"""
import numpy as np


def shard_bounds(n_query, world_size, rank):
    """[begin, end) of the queries owned by `rank`, and the padded shard length ceil(Q/W)."""
    per = (n_query + world_size - 1) // world_size if world_size > 0 else n_query
    b = min(rank * per, n_query)
    e = min(b + per, n_query)
    return b, e, per


class ShardedMatcher:
    """knn2(query, train) -> (idx[Q,2], dist[Q,2]) identical on every rank.

    `local_knn2(q_shard, train) -> (idx, dist)` is the single-GPU kernel.  The default runs the HIP kernel on device
    tensors through the C ABI; tests inject a CPU callable to exercise the partition/gather logic under gloo.
    """

    def __init__(self, group=None, local_knn2=None):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.rank = dist.get_rank(group)
            self.world = dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        self._local = local_knn2 if local_knn2 is not None else self._hip_local
        self._ctx = None
        self._stream = None
        self._out = None
        self._packed = None
        self._gathered = None

    # ---- HIP path: torch device tensors in, torch device tensors out, no host copies, no synchronisation
    def torch_stream(self):
        """The context's HIP stream as a torch stream: make it current (`with torch.cuda.stream(...)`) so that torch
        copies, RCCL collectives and the match kernels are ordered on ONE stream."""
        import torch
        from .context import default_context
        if self._ctx is None:
            self._ctx = default_context()
        if self._stream is None:
            self._stream = torch.cuda.ExternalStream(self._ctx.stream, device=torch.device("cuda", self._ctx.device))
        return self._stream

    def _hip_local(self, q, t):
        import torch
        self.torch_stream()
        if not (q.is_cuda and t.is_cuda and q.dtype == torch.uint8 and t.dtype == torch.uint8):
            raise TypeError("the HIP matcher needs uint8 device tensors [n, 32]")
        nq, nt = q.shape[0], t.shape[0]
        if self._out is None or self._out[0].shape[0] < max(nq, 1) or self._out[0].device != q.device:
            self._out = (torch.empty((max(nq, 1), 2), dtype=torch.int32, device=q.device),
                         torch.empty((max(nq, 1), 2), dtype=torch.int32, device=q.device))
        idx, dist = self._out
        # NULL stream argument = the context's stream (the one torch_stream() wraps)
        self._ctx.hamming_knn2_dev(q.data_ptr(), nq, t.data_ptr(), nt, idx.data_ptr(), dist.data_ptr(), None)
        return idx[:nq], dist[:nq]

    def knn2_local_shard(self, q_shard, train):
        """The per-rank compute only (what bench.py times at world_size 1)."""
        return self._local(q_shard, train)

    def knn2(self, query, train):
        """query: the FULL query set (replicated input, as the reference's caller holds it); returns full results."""
        import torch
        nq = query.shape[0]
        b, e, per = shard_bounds(nq, self.world, self.rank)
        idx, dist = self._local(query[b:e], train)
        if self.world == 1:
            return idx, dist
        return self.gather_shards(idx, dist, nq)

    def gather_shards(self, idx, dist, n_query):
        """All-gather (idx, dist) of this rank's shard; shards are padded to ceil(Q/W) rows so the collective is a plain
        equal-size all-gather (one RCCL call, 16 B per query)."""
        import torch
        b, e, per = shard_bounds(n_query, self.world, self.rank)
        dev = idx.device
        if self._packed is None or self._packed.shape[0] != per or self._packed.device != dev:
            self._packed = torch.empty((per, 4), dtype=torch.int32, device=dev)
            self._gathered = torch.empty((self.world * per, 4), dtype=torch.int32, device=dev)
        n = e - b
        self._packed[:n, 0:2] = idx
        self._packed[:n, 2:4] = dist
        if n < per:
            self._packed[n:] = -1
        self._dist.all_gather_into_tensor(self._gathered, self._packed, group=self.group)
        out = self._gathered[:n_query] if per * self.world >= n_query else self._gathered
        # rows of rank r sit at [r*per, r*per + len_r); with ceil partition only trailing ranks are short, so the
        # first n_query rows are exactly the queries in order
        return out[:, 0:2], out[:, 2:4]
