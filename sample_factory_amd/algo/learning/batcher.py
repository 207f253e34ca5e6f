"""Row bookkeeping between sampler and learner — the protocol of sample_factory/algo/learning/batcher.py:89-271
(`Batcher`) without its copy, on an occupancy map instead of the reference's boundary dictionaries (:22-86).

The reference gathers the trajectory slices of a dataset into a separate training batch (batcher.py:192-212, a 3.8 GB
copy per dataset at config 2); here the learner trains on the slab rows in place, so a "training batch" is just a
contiguous row range and releasing it returns the rows to the sampler.

Every range that moves through this module is a whole number of GRANULES (granule = the smaller of "rows of one sampling
round of a unit" and "rows of one dataset"; BufferMgr guarantees one divides the other), so the state of the slab is a
small dense array with one entry per granule: 0 = not held by this ledger, otherwise the AGE STAMP of the run of held
granules it belongs to.  Adjacent held granules always belong to one run (adding rows next to a run fuses them under a
fresh stamp), runs are handed out oldest stamp first — which is exactly the hand-out order of the reference's
insertion-ordered dictionaries, pinned by the traces recorded from the reference in tests/golden/host_logic.json.
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np


class RowLedger:
    """Which rows of [0, num_rows) are held, as maximal runs with an age.

    add(start, stop)       rows become held; fuses with the runs touching them (the fused run is the youngest)
    take(n, exact=True)    rows [lo, lo+n) of the OLDEST run that is at least n rows long (exact) / the first
                           min(n, len) rows of the oldest run (exact=False); what is left of that run becomes the
                           youngest run; None if no run qualifies
    """

    def __init__(self, num_rows: int, granule: int = 1):
        assert num_rows % granule == 0
        self.granule = granule
        self.age = np.zeros(num_rows // granule, dtype=np.int64)  # per granule: 0 = free of this ledger, else run stamp
        self.clock = 0

    # ---- views
    @property
    def total_num(self) -> int:
        return int(np.count_nonzero(self.age)) * self.granule

    def runs(self):
        """[(start_row, stop_row, stamp)] of the held runs, in row order"""
        held = np.concatenate(([0], (self.age != 0).astype(np.int8), [0]))
        edges = np.flatnonzero(np.diff(held))  # alternating run starts / stops (in granules)
        g = self.granule
        return [(int(a) * g, int(b) * g, int(self.age[a])) for a, b in zip(edges[0::2], edges[1::2])]

    @property
    def run_starts(self) -> List[int]:
        return [r[0] for r in self.runs()]

    # ---- updates
    def _stamp(self, lo: int, hi: int) -> None:
        self.clock += 1
        self.age[lo:hi] = self.clock

    def add(self, start: int, stop: int) -> None:
        g = self.granule
        assert start % g == 0 and stop % g == 0 and 0 <= start < stop <= len(self.age) * g, (start, stop, g)
        lo, hi = start // g, stop // g
        assert not self.age[lo:hi].any(), f"rows [{start}, {stop}) are already held"
        while lo > 0 and self.age[lo - 1]:          # the run ending where ours begins
            lo -= 1
        while hi < len(self.age) and self.age[hi]:  # the run beginning where ours ends
            hi += 1
        self._stamp(lo, hi)

    def take(self, n: int, exact: bool = True) -> Optional[slice]:
        g = self.granule
        assert n % g == 0 and n > 0, (n, g)
        for start, stop, _ in sorted(self.runs(), key=lambda r: r[2]):
            if exact and stop - start < n:
                continue
            cut = min(stop, start + n)
            self.age[start // g:cut // g] = 0
            if cut < stop:
                self._stamp(cut // g, stop // g)
            return slice(start, cut)
        return None


class Batcher:
    """on_new_trajectories(slice) -> training slices (contiguous, exactly one dataset long) as soon as enough adjacent
    rows have arrived; on_training_batch_released(slice) -> sampling slices back to the BufferMgr queue."""

    def __init__(self, buffer_mgr, cfg):
        self.buffer_mgr, self.cfg = buffer_mgr, cfg
        self.traj_per_training_iteration = buffer_mgr.trajectories_per_training_iteration
        self.traj_per_sampling_iteration = buffer_mgr.sampling_trajectories_per_iteration
        granule = min(self.traj_per_training_iteration, self.traj_per_sampling_iteration)
        rows = buffer_mgr.num_buffers
        assert rows % granule == 0, f"slab of {rows} rows is not a whole number of {granule}-row granules"
        self.rows_for_training = RowLedger(rows, granule)   # written by the sampler, not yet with the learner
        self.rows_for_sampling = RowLedger(rows, granule)   # released by the learner, not yet back in the free queue
        self.in_flight = 0  # datasets handed to the learner and not yet released

    def on_new_trajectories(self, trajectory_slice: slice) -> List[slice]:
        self.rows_for_training.add(trajectory_slice.start, trajectory_slice.stop)
        return self.ready_batches()

    def ready_batches(self) -> List[slice]:
        """datasets that can go to the learner NOW: complete (exactly one training iteration of adjacent rows) and
        within the cap of max_batches_to_accumulate datasets in flight (batcher.py:170-218: no free training batch ->
        the rows wait in the ledger and, once the slab has no free slice left, the sampler pauses)"""
        out = []
        while self.in_flight + len(out) < self.buffer_mgr.max_batches_to_accumulate:
            s = self.rows_for_training.take(self.traj_per_training_iteration)
            if s is None:
                break
            out.append(s)
        self.in_flight += len(out)
        return out

    def on_training_batch_released(self, training_slice: slice) -> int:
        """returns the number of sampling slices put back into the free queue"""
        self.in_flight -= 1
        self.rows_for_sampling.add(training_slice.start, training_slice.stop)
        n = 0
        while (s := self.rows_for_sampling.take(self.traj_per_sampling_iteration)) is not None:
            self.buffer_mgr.release(s)
            n += 1
        return n
