"""Query-sharded brute-force matcher: one process per GPU, one RCCL all-gather of the per-shard best matches.

BASELINE.json north_star: "Shard only the brute-force descriptor match (query-split) across the 8 GPUs of one node with
an RCCL all-gather of per-shard best matches over xGMI; detection and BA stay single-GPU."  The reference has no
distributed code at all; the call it shards is FeatureMatcher.match_features' knnMatch
(reference src/v2/frame.py:23).

Partition: rank r owns queries [r*ceil(Q/W), (r+1)*ceil(Q/W)); the train set is replicated, so every
rank produces final (global train index, distance) pairs for its queries and no cross-rank tie-breaking exists.  The
exchange is a single all-gather of 16 bytes per query -- the kernel writes (idx0, idx1, dist0, dist1) rows directly in
the gather layout (vs_hamming_knn2_packed_dev), shards padded to equal length.  At world_size 1 the same code path runs
without a collective.

Streaming use (bench.py --gpus N): `submit()` enqueues the local match and starts the all-gather asynchronously on
RCCL's stream; `collect()` of step k is called after `submit()` of step k+1, so the collective of one step overlaps
the kernels of the next (two rotating buffer sets).

Ordering of a step (enforced inside vs_hamming_knn2_sharded_dev, include/vslam_hip.h): before the kernel overwrites a
gather buffer its compute stream waits (1) for the `done` event of the previous step on that buffer -- the in-place
all-gather sends from and receives into those very rows -- and (2) for the caller's stream as of submit(): the consumers
of the buffer's previous results and the producers of the new inputs.  The inputs of a step must stay untouched until
that step has been collected.
"""


class RcclAllGather:
    """A communicator of our own (created next to torch's, bootstrapped through torch.distributed) and a dedicated HIP
    stream for the all-gather that vs_hamming_knn2_sharded_dev issues through the RCCL C API.  torch.distributed's
    Python path costs 30-50 us of host time per collective, more than half a 10k x 10k match step; the raw call costs a
    few microseconds, and running it on its own stream lets it overlap the next step's kernels.  (RCCL = the NCCL API on
    ROCm; the library torch itself loaded is used so only one RCCL lives in the process.)"""

    def __init__(self, group=None):
        import ctypes as C
        import os
        import torch
        import torch.distributed as dist
        self._C = C
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        dev = torch.device("cuda", torch.cuda.current_device())

        class UniqueId(C.Structure):
            _fields_ = [("internal", C.c_byte * 128)]

        # every step that can fail on one rank alone is followed by an agreement, so that no rank is ever left alone
        # inside a collective: (1) the library loads everywhere, (2) rank 0 obtained an id, then the collective init
        ok = 1
        try:
            path = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
            self.lib = C.CDLL(path if os.path.exists(path) else "librccl.so")
            self.lib.ncclGetUniqueId.argtypes = [C.POINTER(UniqueId)]
            self.lib.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
            self.lib.ncclCommDestroy.argtypes = [C.c_void_p]
            self.lib.ncclGetErrorString.restype = C.c_char_p
        except (OSError, AttributeError):
            ok = 0
        uid = UniqueId()
        if ok and self.rank == 0 and self.lib.ncclGetUniqueId(C.byref(uid)) != 0:
            ok = 0
        flag = torch.tensor([ok], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            raise RuntimeError("RCCL library / unique id not available on every rank")
        t = torch.tensor(list(bytes(uid)), dtype=torch.uint8, device=dev)
        dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        C.memmove(C.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
        # ncclCommInitRank is itself a collective: if one rank fails before joining, its peers would wait inside it for
        # ever.  It therefore runs on a helper thread with a time limit (VS_RCCL_INIT_TIMEOUT seconds, default 60); a
        # rank that times out reports failure to the agreement that follows in ShardedMatcher._init_direct, and every
        # rank falls back to torch.distributed's collective together.
        import threading
        self.comm = C.c_void_p()
        result = {}

        def _init():
            torch.cuda.set_device(dev)
            result["rc"] = self.lib.ncclCommInitRank(C.byref(self.comm), self.world, uid, self.rank)

        th = threading.Thread(target=_init, daemon=True)
        th.start()
        th.join(float(os.environ.get("VS_RCCL_INIT_TIMEOUT", "60")))
        if th.is_alive():
            self.comm = None
            raise RuntimeError("ncclCommInitRank did not return within the time limit (a peer failed to join)")
        self._chk(result.get("rc", 1))
        self.stream = torch.cuda.Stream(device=dev)

    def _chk(self, rc):
        if rc != 0:
            raise RuntimeError("RCCL error %d: %s" % (rc, self.lib.ncclGetErrorString(rc).decode()))

    def count(self):
        """Number of ranks in THIS communicator as RCCL itself reports it (ncclCommCount) -- what bench.py puts on its JSON
        line as `rccl_ranks`, so that a scaling record can be checked against the collective that actually ran."""
        C = self._C
        n = C.c_int(-1)
        self.lib.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._chk(self.lib.ncclCommCount(self.comm, C.byref(n)))
        return int(n.value)

    def device(self):
        """HIP device ordinal the communicator is bound to (ncclCommCuDevice)."""
        C = self._C
        d = C.c_int(-1)
        self.lib.ncclCommCuDevice.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        self._chk(self.lib.ncclCommCuDevice(self.comm, C.byref(d)))
        return int(d.value)

    def close(self):
        if getattr(self, "comm", None):
            self.lib.ncclCommDestroy(self.comm)
            self.comm = None


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
    force_collective: run the all-gather even at world size 1 (rehearses the N > 1 code path on one GPU).

    Stream contract: inputs may be produced on, and results consumed from, torch's current stream -- submit / collect /
    knn2 insert the event waits between it and the library's stream (none are needed when the caller already works on
    `torch_stream()`).
    """

    def __init__(self, group=None, local_knn2=None, force_collective=False):
        import torch.distributed as dist
        self._dist = dist
        self.group = group
        if dist.is_available() and dist.is_initialized():
            self.rank = dist.get_rank(group)
            self.world = dist.get_world_size(group)
        else:
            self.rank, self.world = 0, 1
        self._local = local_knn2
        self._ctx = None
        self._stream = None
        self._bufs = {}   # (slot, rows) -> (packed [rows,4], gathered [W*rows,4])
        self._slot = 0
        self._rccl = None
        self._rccl_failed = False
        self._events = {}
        self._force_collective = force_collective
        # diagnostics for a scaling record (bench.py puts them on its JSON line, per rank): seconds spent creating the direct
        # communicator (unique id + broadcast + ncclCommInitRank + the agreement), and why this rank uses the path it uses
        self.bootstrap_s = None
        self.path_reason = "not decided yet (no step submitted)"

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

    def _buffers(self, rows, device, slot):
        import torch
        key = (slot, rows, str(device))
        if key not in self._bufs:
            packed = torch.empty((max(rows, 1), 4), dtype=torch.int32, device=device)
            gathered = torch.empty((self.world * max(rows, 1), 4), dtype=torch.int32, device=device) \
                if self.world > 1 or self._dist.is_initialized() else None
            self._bufs[key] = (packed, gathered)
        return self._bufs[key]

    def _order_after_caller(self):
        """Make the library's stream wait for everything the caller has enqueued on torch's CURRENT stream (the producers
        of q / t), unless that already is the library's stream."""
        import torch
        mine = self.torch_stream()
        cur = torch.cuda.current_stream(mine.device)
        if cur.cuda_stream != mine.cuda_stream:
            ev = torch.cuda.Event()
            ev.record(cur)
            mine.wait_event(ev)

    def _order_caller_after(self, event=None):
        """Make torch's CURRENT stream wait for the results (the all-gather's `event`, else the work enqueued so far on
        the library's stream), so the tensors handed back are safe to use on it."""
        import torch
        mine = self.torch_stream()
        cur = torch.cuda.current_stream(mine.device)
        if event is None:
            if cur.cuda_stream == mine.cuda_stream:
                return
            event = torch.cuda.Event()
            event.record(mine)
        cur.wait_event(event)

    def _local_packed(self, q, t, rows, slot):
        """Match this rank's queries; returns the packed [rows, 4] tensor (first len(q) rows valid)."""
        import torch
        nq, nt = q.shape[0], t.shape[0]
        packed, _ = self._buffers(rows, q.device, slot)
        if self._local is not None:  # injected kernel (CPU tests)
            idx, dist = self._local(q, t)
            packed[:nq, 0:2] = idx
            packed[:nq, 2:4] = dist
            return packed
        self.torch_stream()
        if not (q.is_cuda and t.is_cuda and q.dtype == torch.uint8 and t.dtype == torch.uint8):
            raise TypeError("the HIP matcher needs uint8 device tensors [n, 32]")
        self._order_after_caller()
        # NULL stream argument = the context's stream (the one torch_stream() wraps)
        self._ctx.hamming_knn2_packed_dev(q.data_ptr(), nq, t.data_ptr(), nt, packed.data_ptr(), None)
        return packed

    def knn2_local_shard(self, q_shard, train):
        """The per-rank compute only (what bench.py times at world size 1): (idx, dist) views of the packed rows, ordered
        for use on torch's current stream."""
        nq = q_shard.shape[0]
        packed = self._local_packed(q_shard, train, nq, 0)
        if self._local is None:
            self._order_caller_after()
        return packed[:nq, 0:2], packed[:nq, 2:4]

    def submit(self, q_shard, train, n_query):
        """Enqueue the local match of this rank's shard and start the all-gather (async).  Returns a ticket."""
        b, e, per = shard_bounds(n_query, self.world, self.rank)
        assert q_shard.shape[0] == e - b, "q_shard must be this rank's slice of the query set"
        slot = self._slot
        self._slot ^= 1
        collective = self._dist.is_initialized() and (self.world > 1 or self._force_collective)
        if collective and self._direct_ready(q_shard):
            return self._submit_direct(q_shard, train, n_query, per, slot)
        packed = self._local_packed(q_shard, train, per, slot)
        _, gathered = self._buffers(per, q_shard.device, slot)
        work = None
        if gathered is not None and collective:
            work = ("work", self._torch_all_gather(gathered, packed))
            out = gathered
        else:
            out = packed
        return (work, out, n_query)

    # ---- direct path: vs_hamming_knn2_sharded_dev = kernel into this rank's slot of the gather buffer + one in-place
    # ncclAllGather on RCCL's own stream, ordered by events inside the library (no host synchronisation)
    def _direct_ready(self, q):
        import os
        if self._local is not None or not q.is_cuda:
            self.path_reason = "torch.distributed collective: " + ("injected CPU kernel" if self._local is not None else "host tensors")
            return False
        if os.environ.get("VS_SHARDED_TORCH_COLLECTIVE", "0") == "1":
            self.path_reason = "torch.distributed collective: VS_SHARDED_TORCH_COLLECTIVE=1"
            return False
        if self._rccl_failed:
            return False  # (path_reason says why, from _init_direct)
        if self._rccl is None:
            self._init_direct(q.device)
        return self._rccl is not None

    def _submit_direct(self, q, t, n_query, per, slot):
        import torch
        _, gathered = self._buffers(per, q.device, slot)
        key = gathered.data_ptr()
        if key not in self._events:               # one `done` event per rotating buffer
            ev = torch.cuda.Event()
            ev.record(self._rccl.stream)          # torch creates the hipEvent_t lazily, on the first record
            self._events[key] = ev
        done = self._events[key]
        self._order_after_caller()
        # the library's stream first waits for `done` as the previous step on this buffer recorded it (inside the C call)
        self._ctx.hamming_knn2_sharded_dev(q.data_ptr(), q.shape[0], t.data_ptr(), t.shape[0], gathered.data_ptr(), per,
                                           self.rank, self.world, self._rccl.comm.value, None,
                                           self._rccl.stream.cuda_stream, done.cuda_event)
        return (("event", done), gathered, n_query)

    def _torch_all_gather(self, gathered, packed):
        """Fallback: torch.distributed's collective.  It is issued against torch's CURRENT stream, which must first wait
        for the match kernel on the library's stream."""
        if packed.is_cuda and self._local is None:
            self._order_caller_after()
        return self._dist.all_gather_into_tensor(gathered, packed, group=self.group, async_op=True)

    def _init_direct(self, device):
        """Create the direct communicator on every rank, then agree on the outcome: if any rank failed (or timed out
        inside ncclCommInitRank), all ranks use torch.distributed's collective -- a rank alone in ncclAllGather would
        wait for ever."""
        import time
        import torch
        ok, why = 1, None
        t0 = time.perf_counter()
        try:
            self._rccl = RcclAllGather(self.group)
            self._events = {}
        except (OSError, AttributeError, RuntimeError) as e:
            self._rccl, ok, why = None, 0, "%s: %s" % (type(e).__name__, e)
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        self._dist.all_reduce(flag, op=self._dist.ReduceOp.MIN, group=self.group)
        self.bootstrap_s = time.perf_counter() - t0
        if int(flag.item()) == 0:
            self._rccl_failed = True
            self.path_reason = "torch.distributed collective: " + (
                "this rank could not create its own communicator (%s)" % why if why else
                "another rank could not create its communicator (all ranks fall back together)")
            if self._rccl is not None:
                self._rccl.close()
                self._rccl = None
        else:
            self.path_reason = "rccl_direct: ncclCommInitRank of %d rank(s) succeeded on every rank in %.3f s" % (self.world, self.bootstrap_s)

    def collect(self, ticket):
        """Wait for a ticket's all-gather; returns (idx [Q,2], dist [Q,2]) ordered for use on torch's current stream.
        With the ceil partition only trailing ranks are short, so the first Q rows of the gathered buffer are exactly
        the queries in order."""
        work, out, n_query = ticket
        if work is not None:
            kind, h = work
            if kind == "event":
                self._order_caller_after(h)        # stream-side wait, the host does not block
                self.torch_stream().wait_event(h)  # the library's stream too: its next step may reuse the buffer
            else:
                h.wait()
        elif self._local is None and out.is_cuda:
            self._order_caller_after()
        if self._local is None and self._ctx is not None:
            self._ctx.match_status()   # (see _Plan.collect)
        return out[:n_query, 0:2], out[:n_query, 2:4]

    def plan(self, q_shard, train, n_query, single_stream=False, in_flight=2, buffers=None, static_inputs=False):
        """A pre-bound step for a fixed shape (bench.py, streaming callers): every ctypes argument of the rotating buffer
        sets is built once, so `submit()` is one C call (stream waits + kernel [+ event + in-place ncclAllGather] + done
        event) and `collect(slot)` one stream-side wait.  The caller works on `torch_stream()` (checked here); other
        callers use submit / collect / knn2, which order against torch's current stream themselves.
        buffers: rotating buffer sets (default: one per stream in flight, at least two).  static_inputs=True: the inputs are
        complete before the first submit() and never change, and the caller is done with a slot's results (synchronised, or
        waited for on every stream it uses) before the submit() that reuses the slot, `buffers` steps later -- the step's kernel
        is then not ordered behind the library's stream at all, and the exchange runs on that stream."""
        import torch
        mine = self.torch_stream()
        if torch.cuda.current_stream(mine.device).cuda_stream != mine.cuda_stream:
            raise RuntimeError("ShardedMatcher.plan: make torch_stream() the current stream first")
        if self._local is not None:
            raise RuntimeError("ShardedMatcher.plan needs the HIP path")
        return _Plan(self, q_shard, train, n_query, 1 if single_stream else int(in_flight), buffers, bool(static_inputs))

    def rccl_ranks(self):
        """Ranks of the matcher's own RCCL communicator (ncclCommCount), None when the direct path is not in use (world
        size 1 without force_collective, an injected CPU kernel, or the torch.distributed fallback)."""
        return None if self._rccl is None else self._rccl.count()

    def collective_path(self):
        """Which exchange a step uses: 'rccl_direct' (in-place ncclAllGather inside vs_hamming_knn2_sharded_dev),
        'torch_distributed' (fallback) or 'none' (one rank)."""
        if self._rccl is not None:
            return "rccl_direct"
        collective = self._dist.is_initialized() and (self.world > 1 or self._force_collective)
        if not collective:
            self.path_reason = "none: one rank, no collective asked for"
        return "torch_distributed" if collective else "none"

    def diagnostics(self):
        """{path, reason, bootstrap_s, rccl_ranks} of THIS rank (bench.py gathers them over the ranks)."""
        path = self.collective_path()
        return {"rank": self.rank, "path": path, "reason": self.path_reason, "bootstrap_s": self.bootstrap_s,
                "rccl_ranks": self.rccl_ranks()}

    def close(self):
        """Destroy the direct RCCL communicator (if one was created)."""
        try:
            if self._ctx is not None and self._local is None:
                import torch
                torch.cuda.synchronize()
                self._ctx.match_status()   # every step has completed: a lost chunk cannot go unreported past this point
        finally:
            if self._rccl is not None:
                try:
                    self._rccl.close()
                finally:
                    self._rccl = None

    def knn2(self, query, train):
        """query: the FULL query set (replicated input, as the reference's caller holds it); returns full results."""
        nq = query.shape[0]
        b, e, _ = shard_bounds(nq, self.world, self.rank)
        return self.collect(self.submit(query[b:e], train, nq))


class _Plan:
    """`in_flight` steps in flight (default two), each slot with a compute stream, a buffer set and a `done` event of its
    own, so the short tail of one launch -- the fold by the last-arriving workgroups, the kernel boundary -- overlaps the
    body of the next.  The library's stream (torch_stream(), the caller's) carries only the caller's own work: producers
    of the inputs before submit(), consumers of the results after collect().

    What orders a step (one C call, vs_hamming_knn2_sharded_dev):
        compute stream of the slot  <-  done[slot] as the slot's PREVIOUS step recorded it (its in-place all-gather reads
                                        and writes the buffer the kernel is about to overwrite)
                                    <-  the library's stream as of submit() (the producers of this step's q / t, the
                                        consumers of the slot's previous results)
        kernel -> [event -> RCCL stream: in-place ncclAllGather] -> done[slot]
        collect(slot): the library's stream waits for done[slot].
    Contract: the inputs of a step stay untouched until that step has been collected -- a streaming caller rotates at
    least `in_flight` input buffers and names them per step, submit(q=..., t=...).  in_flight == 1: everything on the
    library's stream."""

    def __init__(self, m, q, t, n_query, in_flight=2, buffers=None, static_inputs=False):
        import torch
        from . import _capi
        b, e, per = shard_bounds(n_query, m.world, m.rank)
        assert q.shape[0] == e - b, "q_shard must be this rank's slice of the query set"
        self.m, self.n_query, self.slot = m, n_query, 0
        self.lib, self.h = _capi.load(), m._ctx.handle
        collective = m._dist.is_initialized() and (m.world > 1 or m._force_collective)
        self.direct = collective and m._direct_ready(q)
        self.fallback = collective and not self.direct
        assert 1 <= in_flight <= 4
        # `in_flight` compute streams, `buffers` rotating buffer sets (at least as many; default: one per stream, two on one
        # stream).  More buffer sets than streams take the exchange off the chain of launches: a step's kernel has to wait for
        # the all-gather that last used ITS buffer set -- with two sets that is the step before the previous one, whose in-place
        # all-gather has only just been enqueued behind its kernel; with four sets it finished two steps ago.
        self.nslots = max(2, in_flight, int(buffers) if buffers else 0)
        assert self.nslots <= 8
        # static_inputs (contract in ShardedMatcher.plan): the step's kernel is NOT ordered behind the library's stream, which
        # carries the waits of earlier collect()s -- ordering every kernel behind them makes step k + 1 wait for the all-gather
        # of step k - 1
        self.static_inputs = static_inputs
        self.shape = (tuple(q.shape), tuple(t.shape))
        main = m._stream
        # where the in-place all-gather runs.  A stream made after the context (the communicator's own) may share a hardware
        # queue with one of the compute streams: the all-gather of step k, waiting there for kernel k, then holds back kernel
        # k + 1 behind it (measured at one rank, 10k x 10k: 65 us / step against 47 without the collective).  With static
        # inputs nothing else is enqueued on the library's stream between collect()s, and it owns a queue: 49 us / step.
        # With per-step inputs the library's stream carries their producers, which must not queue behind an all-gather.
        comm = main if static_inputs else (m._rccl.stream if self.direct else None)
        one_kernel_stream = None
        if in_flight == 1 and self.direct:
            # One launch at a time WITH a collective: compute on one stream, the exchange on another, and neither of them the
            # library's.  collect() orders the library's stream (the caller's: the consumers) behind a step's all-gather; were
            # the kernels launched on that stream too, kernel k + 2 would queue behind the all-gather of step k -- which a
            # kernel that fills the device lets in only at its tail (measured at one rank, 10k x 10k: 81 us per step against 54
            # without the collective).  So with static inputs the kernels run back to back on the context's first auxiliary
            # stream and the all-gathers on its second (both made with the context: hardware queues of their own); a kernel
            # waits only for the all-gather that last used ITS buffer set, `buffers` steps ago.  With per-step inputs the
            # kernels must follow their producers on the library's stream: they stay there, the exchange on an auxiliary stream.
            a0, a1 = m._ctx.aux_stream(0), m._ctx.aux_stream(1)
            if static_inputs and a0 and a1:
                one_kernel_stream = torch.cuda.ExternalStream(a0, device=q.device)
                comm = torch.cuda.ExternalStream(a1, device=q.device)
            else:
                comm = torch.cuda.ExternalStream(a0, device=q.device) if a0 else m._rccl.stream
            self.comm_stream = comm
        # compute streams of the slots: the context's own auxiliary streams first (created with the context, each on a
        # hardware queue of its own -- streams made later may share a queue with the library's and then never overlap it),
        # torch streams only beyond those
        pool = []
        for i in range(max(in_flight, 1)):
            aux = m._ctx.aux_stream(i) if in_flight > 1 else None
            pool.append((one_kernel_stream or main) if in_flight == 1 else torch.cuda.ExternalStream(aux, device=q.device) if aux
                        else torch.cuda.Stream(device=q.device))
        self.streams = [pool[slot % len(pool)] for slot in range(self.nslots)]
        self.args, self.outs, self.done, self.bufs, self.keep = [], [], [], [], []
        nq, nt = q.shape[0], t.shape[0]
        for slot in range(self.nslots):
            packed, gathered = m._buffers(per, q.device, 2 + slot)  # buffer sets of their own
            cs = self.streams[slot]
            ev = torch.cuda.Event()
            ev.record(cs)  # creates the hipEvent_t; complete by the time anything waits for it
            after = main.cuda_stream if cs is not main and not static_inputs else None
            if self.direct:   # this rank's rows go straight into its slot of the gather buffer, then the in-place all-gather
                args = [self.h, q.data_ptr(), nq, t.data_ptr(), nt, gathered.data_ptr(), per, m.rank, m.world,
                        m._rccl.comm.value, cs.cuda_stream, comm.cuda_stream, ev.cuda_event, after]
                out = gathered
            else:             # no collective inside the call (world 1, or torch's collective afterwards): packed rows only
                # everything on the library's own stream is ordered by the stream itself: no event between two launches
                # (an event record / wait pair keeps back-to-back kernels a few microseconds apart)
                args = [self.h, q.data_ptr(), nq, t.data_ptr(), nt, packed.data_ptr(), per, 0, 1,
                        None, cs.cuda_stream, None, ev.cuda_event if cs is not main else None, after]
                out = gathered if self.fallback else packed
            self.done.append(ev)
            self.args.append(args)
            self.bufs.append((packed, gathered))
            self.keep.append((q, t))
            self.outs.append((out[:n_query, 0:2], out[:n_query, 2:4]))
        self.work = [None] * self.nslots
        self.throttle = one_kernel_stream is not None

    def submit(self, q=None, t=None):
        """Enqueue one step; returns the slot to hand to collect().  q / t: this step's inputs (device tensors of the plan's
        shapes, produced on the library's stream); default: the tensors the plan was made with."""
        slot = self.slot
        self.slot = (slot + 1) % self.nslots
        args = self.args[slot]
        if q is not None or t is not None:
            if self.static_inputs:
                raise ValueError("ShardedMatcher.plan(static_inputs=True): per-step inputs need the ordering this plan gave up")
            kq, kt = self.keep[slot]
            q = kq if q is None else q
            t = kt if t is None else t
            if (tuple(q.shape), tuple(t.shape)) != self.shape or not (q.is_cuda and t.is_cuda):
                raise ValueError("ShardedMatcher.plan: a step's inputs must be device tensors of the planned shapes")
            args[1], args[3] = q.data_ptr(), t.data_ptr()
            self.keep[slot] = (q, t)  # alive until the slot's next step
        if self.work[slot] is not None:  # torch's collective of the slot's previous step was never collected
            self.work[slot].wait()
            self.work[slot] = None
        if self.throttle:
            # one kernel stream: the host stays at most `nslots` steps ahead of the device (it waits here for the step that last
            # used this buffer set -- rarely for long), so that step's `done` event is complete and the library enqueues no wait
            # for it in front of the kernel (a barrier packet there keeps back-to-back kernels apart)
            self.done[slot].synchronize()
        rc = self.lib.vs_hamming_knn2_sharded_dev(*args)
        if rc != 0:
            self.m._ctx._chk(rc)
        if self.fallback:
            import torch
            packed, gathered = self.bufs[slot]
            with torch.cuda.stream(self.streams[slot]):
                self.work[slot] = self.m._dist.all_gather_into_tensor(gathered, packed, group=self.m.group, async_op=True)
        return slot

    def collect(self, slot):
        """(idx [Q,2], dist [Q,2]) of the step submitted into `slot`, ordered on the library's stream."""
        if self.fallback:
            if self.work[slot] is not None:
                self.work[slot].wait()   # torch's current stream = the library's (plan() checked it)
                self.work[slot] = None
        elif self.streams[slot] is not self.m._stream or self.direct:
            self.m._stream.wait_event(self.done[slot])
        # a launch that lost a train chunk (bounded wait of the fold, ~0.5 s) raised a pinned flag: reported where results are
        # handed out.  The wait above is stream-side, so a flag of THIS step may not be up yet -- then the next collect(), or
        # close(), reports it; one read of a few pinned words per step
        rc = self.lib.vs_match_status(self.h)
        if rc != 0:
            self.m._ctx._chk(rc)
        return self.outs[slot]
