"""Data-parallel gradient exchange for the MicFormer step: whole CT+MR pairs are sharded across ranks (one process per
GPU), weights are replicated, and the ONLY data-path collective is one all-reduce(sum)/world of the flat fp32 gradient
buffer per step (RCCL over xGMI on MI355X via torch.distributed backend "nccl"; gloo on CPU for tests).

xGMI is point-to-point (7 links x ~153 GB/s per GPU): a few LARGE buckets keep every ring step bandwidth- rather than
latency-bound; the bucket size is a parameter.  Device-agnostic on purpose (CPU + gloo exercises the same code).
"""
import torch
import torch.distributed as dist


class FlatGradSync:
    def __init__(self, process_group=None, bucket_bytes=64 << 20, always=False):
        """always=True issues the collectives even on a one-rank group (a test hook: RCCL stream semantics on one GPU)."""
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_available() and dist.is_initialized() else 1
        self.always = bool(always) and dist.is_available() and dist.is_initialized()
        if (self.world > 1 or self.always) and self.pg is None:
            self.pg = dist.group.WORLD
        self.bucket_elems = max(int(bucket_bytes) // 4, 1)

    def broadcast_params(self, flat_params, src=0):
        """Rank-identical initial weights (the reference has a single process; DDP needs this once)."""
        if self.world > 1:
            dist.broadcast(flat_params, src=src, group=self.pg)

    def allreduce_sum_async(self, view):
        """Start sum-all-reduce of one contiguous slice of the flat gradient; returns a work handle (None on one rank).
        torch.distributed orders it after everything already enqueued on the current stream and runs it on the backend's own
        stream, so launches issued afterwards overlap with it; `handle.wait()` re-joins."""
        if self.world <= 1 and not self.always:
            return None
        return dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def max_over_ranks(self, value, device):
        t = torch.tensor([value], dtype=torch.float64, device=device)
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.pg)
        return float(t.item())


def flatten_views(tensors, align=4):
    """Offsets of `tensors` inside one flat buffer, each aligned to `align` elements (16 bytes for fp32)."""
    offs, total = [], 0
    for t in tensors:
        offs.append(total)
        total += (t.numel() + align - 1) // align * align
    return offs, total


def module_buckets(names, offsets, sizes, total, min_elems=1 << 20, depth=3):
    """Contiguous [start, end) slices of the flat buffer, cut where the first `depth` dotted components of the parameter name
    change (swin.layers.2 | swin.layers.3 | ...), tiny groups merged into their successor: the stages finish their weight
    gradients at different times of the step, so each slice can be all-reduced as soon as its last writer is done."""
    order = sorted(range(len(names)), key=lambda i: offsets[i])
    cuts, prev = [0], None
    for i in order:
        key = ".".join(names[i].split(".")[:depth])
        if prev is not None and key != prev and offsets[i] - cuts[-1] >= min_elems:
            cuts.append(offsets[i])
        prev = key
    cuts.append(total)
    return [(a, b) for a, b in zip(cuts, cuts[1:]) if b > a]


def last_writer_per_bucket(buckets, writes):
    """writes: iterable of (launch index, start offset, numel).  -> for every bucket the largest launch index that writes into it
    (-1: nothing queued writes there, it is complete when backward is)."""
    import bisect
    starts = [b[0] for b in buckets]
    last = [-1] * len(buckets)
    for gi, off, n in writes:
        lo = bisect.bisect_right(starts, off) - 1
        hi = bisect.bisect_right(starts, off + max(n, 1) - 1) - 1
        for b in range(lo, hi + 1):
            last[b] = max(last[b], gi)
    return last



class OverlappedGradReduce:
    """The data-parallel tail of a step, device-agnostic: the queued weight-gradient launches are issued group by
    group and every slice of the flat gradient is sum-all-reduced right after the launch that writes into it last (slices
    nothing queued writes to go first), so the collective runs under the remaining launches; then the optimiser consumes
    the SUM with grad_scale = 1/world (the mean is never materialised).  TrainEngine drives it with HIP launches + RCCL;
    tests/test_dist_gloo.py drives the same object with CPU tensors + gloo."""

    def __init__(self, sync, flat_g, buckets, bucket_last, wire=None):
        """wire: optional bfloat16 buffer of flat_g's size -- the slices then cross the links as bf16 (half the bytes: 123 MB instead
        of 247 MB at base): each slice is rounded into it, sum-reduced there, and widened back into flat_g after its wait."""
        self.sync, self.flat_g, self.wire = sync, flat_g, wire
        self.buckets, self.bucket_last = list(buckets), list(bucket_last)

    def _reduce_ready(self, gi, works):
        for (a, b), last in zip(self.buckets, self.bucket_last):
            if last == gi:
                if self.wire is not None and (self.sync.world > 1 or self.sync.always):
                    self.wire[a:b].copy_(self.flat_g[a:b])
                    works.append((self.sync.allreduce_sum_async(self.wire[a:b]), a, b))
                    continue
                w = self.sync.allreduce_sum_async(self.flat_g[a:b])
                if w is not None:
                    works.append((w, None, None))

    def run(self, ngroups, launch_group, pre=None):
        works = []
        if pre is not None:
            pre()
        self._reduce_ready(-1, works)
        for gi in range(ngroups):
            launch_group(gi)
            self._reduce_ready(gi, works)
        for w, a, b in works:
            w.wait()
            if a is not None:
                self.flat_g[a:b].copy_(self.wire[a:b])

    def step_tail(self, ngroups, launch_group, optimizer_step, pre=None):
        """flush + reduce, then optimizer_step(grad_scale) with grad_scale = 1 / world."""
        self.run(ngroups, launch_group, pre)
        optimizer_step(1.0 / self.sync.world)
