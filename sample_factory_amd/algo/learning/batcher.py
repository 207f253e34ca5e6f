"""Slice bookkeeping between sampler and learner — sample_factory/algo/learning/batcher.py:22-86 (SliceMerger) and the
protocol of Batcher (:89-271) without its copy: the reference gathers the trajectory slices of a dataset into a
separate training batch (batcher.py:192-212, a 3.8 GB copy per dataset at config 2); here the learner trains on the
slab rows in place, so a "training batch" is just a contiguous row slice and releasing it returns the rows to the
sampler."""
from __future__ import annotations

from typing import Dict, List, Optional


def slice_len(s: slice) -> int:
    return s.stop - s.start


class SliceMerger:
    """Merges adjacent row slices into longer ones; hands out slices of at most / exactly a given length (dict order =
    insertion order, as in the reference, so the hand-out order is reproducible)."""

    def __init__(self):
        self.slice_starts: Dict[int, slice] = dict()
        self.slice_stops: Dict[int, slice] = dict()
        self.total_num = 0

    def _add_slice(self, s: slice) -> None:
        self.slice_starts[s.start] = s
        self.slice_stops[s.stop] = s
        self.total_num += slice_len(s)

    def _del_slice(self, s: slice) -> None:
        del self.slice_starts[s.start]
        del self.slice_stops[s.stop]
        self.total_num -= slice_len(s)

    def merge_slices(self, trajectory_slice: slice) -> None:
        while True:
            prev_slice = self.slice_stops.get(trajectory_slice.start)
            if prev_slice is not None:  # a slice ends where ours begins
                self._del_slice(prev_slice)
                trajectory_slice = slice(prev_slice.start, trajectory_slice.stop)
                continue
            next_slice = self.slice_starts.get(trajectory_slice.stop)
            if next_slice is not None:  # a slice begins where ours ends
                self._del_slice(next_slice)
                trajectory_slice = slice(trajectory_slice.start, next_slice.stop)
                continue
            self._add_slice(trajectory_slice)
            return

    def _extract_at_most(self, s: slice, batch_size: int) -> slice:
        n = slice_len(s)
        self._del_slice(s)
        if n > batch_size:
            self._add_slice(slice(s.start + batch_size, s.stop))
            s = slice(s.start, s.start + batch_size)
        return s

    def get_at_most(self, batch_size: int) -> Optional[slice]:
        for s in self.slice_starts.values():
            return self._extract_at_most(s, batch_size)
        return None

    def get_exactly(self, batch_size: int) -> Optional[slice]:
        for s in self.slice_starts.values():
            if slice_len(s) >= batch_size:
                return self._extract_at_most(s, batch_size)
        return None


class Batcher:
    """on_new_trajectories(slice) -> training slices (contiguous, exactly one dataset long) as soon as enough adjacent
    rows have arrived; on_training_batch_released(slice) -> sampling slices back to the BufferMgr queue."""

    def __init__(self, buffer_mgr, cfg):
        self.buffer_mgr, self.cfg = buffer_mgr, cfg
        self.traj_per_training_iteration = buffer_mgr.trajectories_per_training_iteration
        self.traj_per_sampling_iteration = buffer_mgr.sampling_trajectories_per_iteration
        self.slices_for_training = SliceMerger()
        self.slices_for_sampling = SliceMerger()
        self.in_flight = 0  # datasets handed to the learner and not yet released

    def on_new_trajectories(self, trajectory_slice: slice) -> List[slice]:
        self.slices_for_training.merge_slices(trajectory_slice)
        return self.ready_batches()

    def ready_batches(self) -> List[slice]:
        """datasets that can go to the learner NOW: complete (exactly one training iteration of adjacent rows) and
        within the cap of max_batches_to_accumulate datasets in flight (batcher.py:170-218: no free training batch ->
        the rows wait in the merger and, once the slab has no free slice left, the sampler pauses)"""
        out = []
        while self.in_flight + len(out) < self.buffer_mgr.max_batches_to_accumulate:
            s = self.slices_for_training.get_exactly(self.traj_per_training_iteration)
            if s is None:
                break
            out.append(s)
        self.in_flight += len(out)
        return out

    def on_training_batch_released(self, training_slice: slice) -> int:
        """returns the number of sampling slices put back into the free queue"""
        self.in_flight -= 1
        self.slices_for_sampling.merge_slices(training_slice)
        n = 0
        while True:
            s = self.slices_for_sampling.get_exactly(self.traj_per_sampling_iteration)
            if s is None:
                return n
            self.buffer_mgr.release(s)
            n += 1
