"""Data-parallel learner replicas (C1 — new capability, the reference has one learner per policy:
sample_factory/algo/utils/shared_buffers.py:26-32).

One process per GPU, envs partitioned by rank; rollout, inference, bootstrap values and the GAE/V-trace scans need no
communication (recurrences run along T only).  Per SGD step the replicas exchange:
  * 3 doubles  {sum(adv), sum(adv^2), n_valid}   before the loss  -> global per-minibatch advantage normalisation
  * 1 bucket   flat fp32 gradient (already scaled by the GLOBAL 1/n_valid) -> sum == gradient of the global mean loss
and once per dataset 3 doubles of return moments (returns normaliser) and the invalid count; loss / KL / entropy sums and
the max KL travel as ONE packed bucket (`reduce_sum_max`: gathered, summed locally, MAX for the max column) once per
epoch — per SGD step only for the per-minibatch KL-adaptive learning-rate schedule.  Adam then runs
redundantly on identical inputs, so weights never need an all-gather.  `torch.distributed` backend "nccl" is RCCL on
ROCm; the same code runs over gloo on CPU tensors (tests/test_dp_gloo.py).

cfg.dp_native_rccl (opt-in): the GRADIENT buckets go through the C-ABI instead (`sf_allreduce_grads`, csrc/sf_dp.hip:
one RCCL communicator per rank created from an id that rank 0 broadcasts over the torch process group, collectives
enqueued on a dedicated exchange stream) — the path a host without torch.distributed binds.  The small
scalar exchanges stay on torch.distributed.  Executed on hardware with ONE rank only so far (tests/test_gpu_dp.py), hence
not the default.
"""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


class _Done:
    def wait(self) -> None:
        return None


class _EventHandle:
    """wait(): order the CURRENT stream behind a collective that was enqueued on the exchange stream"""

    def __init__(self, event, group=None):
        self.event, self.group = event, group

    def wait(self) -> None:
        if self.group is not None:
            self.group._mark("grad_async_wait")
        with _Exposed(self.group):
            torch.cuda.current_stream().wait_event(self.event)


class _TimedWork:
    """torch.distributed Work whose wait() is bracketed by events on the compute stream (exposed wait time)"""

    def __init__(self, work, group):
        self.work, self.group = work, group

    def wait(self) -> None:
        self.group._mark("grad_async_wait")
        with _Exposed(self.group):
            self.work.wait()


class _Exposed:
    """event pair on the CURRENT (compute) stream around a point where it may have to wait for a collective: the elapsed
    time between the two is what the exchange cost the compute stream (0 when the collective had already finished)"""

    def __init__(self, group):
        self.group = group
        self.on = group is not None and group.timing is not None and torch.cuda.is_available()

    def __enter__(self):
        if self.on:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        return self

    def __exit__(self, *a):
        if self.on:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.group.timing["exposed"].append((self.e0, e1))
        return False


class ReplicaGroup:
    def __init__(self, process_group=None, force_collectives: bool = False, native_rccl: bool = False,
                 oneshot_bytes: int = 0):
        self.pg = process_group
        self.active = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if self.active else 1
        self.rank = dist.get_rank(process_group) if self.active else 0
        # cfg.dp_force_collectives: issue every collective even in a group of ONE rank — the only way to execute the
        # nccl (= RCCL) branch (dtypes, reduce ops, async bucket slices, stream ordering) on a single-GPU box
        self.on = self.active and (self.world > 1 or force_collectives)

        # gloo (CPU collectives) with device tensors: stage through the host.  Only used to exercise the replica
        # protocol on boxes with fewer GPUs than ranks (tests); production runs use nccl (= RCCL over xGMI).
        self._stage = self.active and dist.get_backend(process_group) == "gloo"
        self._comm = self._xstream = None
        # optional measurement (bench.py --gpus N): {"exposed": [(e0, e1)], "xchg": [(e0, e1)], "count": n}
        self.timing = None
        # optional order trace (tests/test_gpu_dp.py): [(tag, HIP event recorded on the CURRENT stream at that point)] —
        # where the gradient buckets are enqueued / waited for relative to the backward pass's launches
        self.trace = None
        if native_rccl and self.on:
            self._init_native()
        # cfg.dp_oneshot_bytes > 0 (opt-in): small SUM buckets through the one-shot mailbox exchange of csrc/sf_dp.hip
        self._os = None
        self._os_cap = 0
        if oneshot_bytes > 0 and self.on and torch.cuda.is_available():
            self._init_oneshot(int(oneshot_bytes))

    def _init_oneshot(self, cap: int) -> None:
        """every rank allocates its mailbox, the hipIpc handles travel over the torch process group (64 bytes per rank), every
        rank maps every peer's mailbox.  One context per group; buckets of at most `cap` bytes qualify."""
        from sample_factory_amd import lib
        ctx, handle = lib.dp_oneshot_create(self.world, self.rank, cap)
        dev = torch.device("cpu") if self._stage else torch.device("cuda", torch.cuda.current_device())
        mine = torch.frombuffer(bytearray(handle), dtype=torch.uint8).to(dev)
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        dist.all_gather(parts, mine, group=self.pg)
        lib.dp_oneshot_connect(ctx, b"".join(bytes(p.cpu().numpy().tobytes()) for p in parts))
        self._os, self._os_cap, self._oslib = ctx, cap, lib

    def _oneshot_ok(self, t: torch.Tensor) -> bool:
        return (self._os is not None and t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.float64)
                and 0 < t.numel() * t.element_size() <= self._os_cap)

    def _mark(self, tag: str) -> None:
        if self.trace is not None and torch.cuda.is_available():
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            self.trace.append((tag, ev))

    def enable_timing(self) -> None:
        """bracket every collective with HIP events: `exposed` = pairs on the compute stream around each point where it
        waits for (or runs) a collective; `xchg` = pairs on the exchange stream around the collective itself (native RCCL
        path only: torch.distributed runs its collectives on a stream of its own that cannot be instrumented from here)"""
        self.timing = dict(exposed=[], xchg=[], count=0)

    def timing_summary(self):
        """(exposed_ms, exchange_ms | None, collectives) accumulated since enable_timing(); synchronises"""
        if self.timing is None:
            return None
        torch.cuda.synchronize()
        ex = sum(a.elapsed_time(b) for a, b in self.timing["exposed"])
        xc = sum(a.elapsed_time(b) for a, b in self.timing["xchg"]) if self.timing["xchg"] else None
        return ex, xc, self.timing["count"]

    # ---- gradient buckets through the C-ABI (sf_allreduce_grads)
    def _init_native(self) -> None:
        from sample_factory_amd import lib
        dev = torch.device("cpu") if self._stage else torch.device("cuda", torch.cuda.current_device())
        ident = torch.zeros(lib.DP_UNIQUE_ID_BYTES, dtype=torch.uint8, device=dev)
        if self.rank == 0:
            ident.copy_(torch.frombuffer(bytearray(lib.dp_unique_id()), dtype=torch.uint8))
        dist.broadcast(ident, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
        self._comm = lib.dp_comm_create(bytes(ident.cpu().numpy().tobytes()), self.world, self.rank)
        assert lib.dp_comm_info(self._comm) == (self.world, self.rank)
        self._xstream = torch.cuda.Stream()
        self._lib = lib

    @property
    def native(self) -> bool:
        return self._comm is not None

    def _native_reduce(self, t: torch.Tensor) -> _EventHandle:
        """every gradient collective is enqueued on the ONE exchange stream, in program order (one communicator: same
        order on every rank), behind what the current stream has produced so far"""
        cur = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(cur)
        self._xstream.wait_event(ready)
        timed = self.timing is not None
        if timed:
            x0 = torch.cuda.Event(enable_timing=True)
            x0.record(self._xstream)
        self._lib.allreduce_grads(self._comm, t, self._xstream)
        done = torch.cuda.Event(enable_timing=timed)
        done.record(self._xstream)
        if timed:
            self.timing["xchg"].append((x0, done))
            self.timing["count"] += 1
        return _EventHandle(done, self)

    def all_reduce_grads(self, t: torch.Tensor) -> torch.Tensor:
        """SUM of a slice of the flat fp32 gradient over the replicas; the current stream continues behind the result"""
        self._mark("grad_sync")
        if self._oneshot_ok(t):  # the small head bucket (conv layers): one hop, nothing left to hide a ring behind
            return self.all_reduce_sum(t)
        if self.native:
            h = self._native_reduce(t)
            with _Exposed(self):
                torch.cuda.current_stream().wait_event(h.event)
            return t
        return self.all_reduce_sum(t)

    def all_reduce_grads_async(self, t: torch.Tensor):
        self._mark("grad_async_enqueue")
        if self.native:
            return self._native_reduce(t)
        return self.all_reduce_sum_async(t)

    def close(self) -> None:
        if self._comm is not None:
            torch.cuda.synchronize()
            self._lib.dp_comm_destroy(self._comm)
            self._comm = None
        if self._os is not None:
            torch.cuda.synchronize()
            if self.active:
                dist.barrier(group=self.pg)  # nobody unmaps a mailbox a peer's kernel may still read
            self._oslib.dp_oneshot_status(self._os)
            self._oslib.dp_oneshot_destroy(self._os)
            self._os = None

    def _collective(self, fn, t: torch.Tensor) -> torch.Tensor:
        if self.on:
            if self.timing is not None:
                self.timing["count"] += 1
            with _Exposed(self if t.is_cuda else None):
                if self._stage and t.is_cuda:
                    h = t.detach().cpu()
                    fn(h)
                    t.copy_(h)
                else:
                    fn(t)
        return t

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.on and self._oneshot_ok(t):
            if self.timing is not None:
                self.timing["count"] += 1
            with _Exposed(self):
                self._oslib.dp_oneshot_allreduce(self._os, t, "sum")
            return t
        return self._collective(lambda x: dist.all_reduce(x, op=dist.ReduceOp.SUM, group=self.pg), t)

    def all_reduce_sum_async(self, t: torch.Tensor):
        """Start summing `t` over the replicas and return a handle whose wait() orders the CURRENT stream behind the
        result (nccl/RCCL: the collective runs on the backend's own stream, behind everything already enqueued on the
        current one — the DDP bucket pattern).  Staged gloo runs have nothing to overlap with: reduced on the spot."""
        if not self.on or (self._stage and t.is_cuda):
            self.all_reduce_sum(t)
            return _Done()
        if self.timing is not None:
            self.timing["count"] += 1
        return _TimedWork(dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True), self)

    def all_reduce_max(self, t: torch.Tensor) -> torch.Tensor:
        return self._collective(lambda x: dist.all_reduce(x, op=dist.ReduceOp.MAX, group=self.pg), t)

    def broadcast(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        return self._collective(lambda x: dist.broadcast(x, src=src, group=self.pg), t)

    def env_shard(self, envs_per_rank: int) -> Tuple[int, int]:
        """[first, last) global env ids owned by this rank (weak scaling: per-rank work is fixed)."""
        return self.rank * envs_per_rank, (self.rank + 1) * envs_per_rank

    def reduce_sum_max(self, t: torch.Tensor, max_cols) -> torch.Tensor:
        """ONE collective for a small packed bucket whose entries need different reductions: every rank's copy is
        gathered ([world, ...] — a few hundred bytes), then summed locally, except the trailing-dimension columns in
        `max_cols`, which take the maximum over the ranks.  In place; every rank ends with identical values."""
        if not self.on:
            return t

        def fn(x):
            parts = [torch.empty_like(x) for _ in range(self.world)]
            dist.all_gather(parts, x.contiguous(), group=self.pg)
            g = torch.stack(parts)
            red = g.sum(0)
            if len(max_cols):
                red[..., list(max_cols)] = g[..., list(max_cols)].max(0).values
            x.copy_(red)
        return self._collective(fn, t)

    def loss_sums(self, sums: torch.Tensor) -> torch.Tensor:
        """sums[0..3] additive loss sums, sums[4] = max KL (needs MAX), rest additive: one packed exchange."""
        return self.reduce_sum_max(sums, (4,))
