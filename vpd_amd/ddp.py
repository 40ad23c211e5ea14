"""Data-parallel gradient exchange: one process per GPU, bucketed SUM all-reduce
of the flat gradient buffer with RCCL (torch.distributed backend "nccl"),
issued on a side stream as soon as a bucket is final so it overlaps the rest
of backward.  The reference is single-device; the semantics
are defined so that world_size 1 is the reference: the loss is a SUM over
crops (train_vpd_model.py:87), so gradients are SUMMED, not averaged.

The pure functions at the top have no GPU dependency and are what the gloo /
world_size-2 CPU tests exercise.
"""
import os

import torch
import torch.distributed as dist


def shard_sizes(n, world):
    """Contiguous split of n crops over `world` ranks (ragged last batch: some ranks may get 0)."""
    base, rem = divmod(n, world)
    return [base + (1 if r < rem else 0) for r in range(world)]


def shard_slice(n, rank, world):
    sizes = shard_sizes(n, world)
    start = sum(sizes[:rank])
    return slice(start, start + sizes[rank])


def wire_dtype():
    """VPD_DDP_WIRE=bf16 (optional): gradient messages travel -- and are summed by the collective -- in bf16, half the
    bytes on a per-link-bound xGMI ring; every rank receives the same sums, so replicas stay identical.  Default fp32."""
    w = os.environ.get("VPD_DDP_WIRE", "fp32")
    if w not in ("fp32", "bf16"):
        raise ValueError("VPD_DDP_WIRE must be fp32 or bf16, not %r" % w)
    return torch.bfloat16 if w == "bf16" else torch.float32


# wire buffers of the bf16 mode: one per (storage address, element count, dtype, device) for the life of the process -- the gradient
# views a reducer sends are the same ranges of the same flat buffers every step, so nothing is allocated on the comm stream after the
# first step (a fresh bf16 copy per bucket per step was allocator traffic next to the collectives)
_WIRE_BUFFERS = {}


def _wire_buffer(view, dtype):
    key = (view.data_ptr(), view.numel(), dtype, str(view.device))
    buf = _WIRE_BUFFERS.get(key)
    if buf is None:
        if len(_WIRE_BUFFERS) > 64:          # (plans were rebuilt: drop the buffers of the old workspaces)
            _WIRE_BUFFERS.clear()
        buf = _WIRE_BUFFERS[key] = torch.empty(view.shape, dtype=dtype, device=view.device)
    return buf


class _WireWork:
    """An all-reduce of `view` through a buffer of the wire dtype: wait(), then the summed values are copied back."""

    def __init__(self, view, group, async_op, dtype):
        self.view = view
        if dtype == view.dtype:
            self.buf = view
        else:
            self.buf = _wire_buffer(view, dtype)
            self.buf.copy_(view)
        self.work = dist.all_reduce(self.buf, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if not async_op:
            self._back()

    def _back(self):
        if self.buf is not self.view:
            self.view.copy_(self.buf)

    def wait(self):
        self.work.wait()
        self._back()


def all_reduce_buckets(flat, ranges, group=None, async_op=False):
    """SUM all-reduce of flat[off:off+numel] for every bucket, in the given order."""
    works = []
    dt = wire_dtype()
    for off, numel in ranges:
        if numel == 0:
            continue
        w = _WireWork(flat[off:off + numel], group, async_op, dt)
        if async_op:
            works.append(w)
    return works


def average_running_stats(bn_running, group=None):
    """VPD_DDP_AVG_BN=1 (optional): BatchNorm statistics are per rank (no SyncBN); at the end of an epoch every replica
    takes the mean over ranks of the running means / variances, so the checkpoint rank 0 writes -- and the validation pass -- see
    all shards.  In place; a collective: every rank calls it."""
    dist.all_reduce(bn_running, op=dist.ReduceOp.SUM, group=group)
    bn_running.div_(dist.get_world_size(group))
    return bn_running


def all_reduce_lazy(scratch_views, flat, small_idx, group=None, async_op=False):
    """Lazy gradients: SUM all-reduce of every bucket's scratch view (conv weight gradients in the kernels' own layout: a sum
    does not care) and of the other tensors of the flat buffer, gathered into one small message (BatchNorm, fc, motion head,
    stem: ~150 k of the 21.4 M elements).  Returns (works, finish): call finish() once the works are done to scatter the
    small message back."""
    works = []
    dt = wire_dtype()
    for v in scratch_views:
        if v.numel():
            w = _WireWork(v, group, async_op, dt)
            if async_op:
                works.append(w)
    # (the gathered message lives in a buffer kept per index table: the same address every step, so its wire buffer is reused too)
    key = ("small", small_idx.data_ptr(), small_idx.numel(), str(flat.device))
    small = _WIRE_BUFFERS.get(key)
    if small is None:
        small = _WIRE_BUFFERS[key] = torch.empty(small_idx.numel(), dtype=flat.dtype, device=flat.device)
    torch.index_select(flat, 0, small_idx, out=small)
    w = _WireWork(small, group, async_op, dt)
    if async_op:
        works.append(w)
    return works, (lambda: flat.index_copy_(0, small_idx, small))


class GradBucketReducer:
    """Overlaps the bucket all-reduces with backward: libvpdhip records one HIP
    event per bucket on the compute stream; the comm stream waits on it and
    launches that bucket's all-reduce; the compute stream re-joins before AdamW."""

    def __init__(self, engine, group=None):
        self.engine = engine
        self.group = group
        self.comm_stream = torch.cuda.Stream(device=engine.device)
        self.events = []                # one per gradient bucket of the plan (vpd_plan_num_buckets), made on first use

    def _ensure_events(self, n):
        while len(self.events) < n:
            e = torch.cuda.Event()
            e.record(torch.cuda.current_stream(self.engine.device))      # materialises the hipEvent_t handle
            self.events.append(e)

    def event_handles(self, nbuckets):
        self._ensure_events(nbuckets)
        return [e.cuda_event for e in self.events[:nbuckets]]

    def reduce(self, plan, lazy=None):
        """Which buffers travel is the PLAN's state, not the caller's choice: after a lazy backward (engine.backward(events,
        lazy=True)) the conv weight gradients are in the plan's scratch (vpd_plan_grads_pending() == 1) and the bucket
        messages are the scratch ranges, the rest of the flat buffer travelling as one small message behind the last;
        otherwise the flat buffer's bucket ranges.  `lazy`, if given, must agree with the plan (a flat-buffer all-reduce
        after a lazy backward would sum stale conv ranges and leave the scratch the optimizer reads unreduced)."""
        from ._lib import lib
        pending = bool(lib().vpd_plan_grads_pending(plan.handle))
        if lazy is not None and bool(lazy) != pending:
            raise RuntimeError("GradBucketReducer.reduce(lazy=%s) but the plan's last backward was %s" %
                               (lazy, "lazy (gradients pending in the scratch)" if pending else "eager (flat buffer complete)"))
        lazy = pending
        cur = torch.cuda.current_stream(self.engine.device)
        flat = self.engine._grads
        self.overlap = os.environ.get("VPD_DDP_OVERLAP", "1") != "0"
        if not self.overlap:
            # diagnostic: no overlap -- all buckets reduced in line after backward (RCCL's persistent kernels
            # hold CUs; whether overlapping them with 1-block-per-CU conv kernels pays is a measurement, not a given)
            if lazy:
                _, finish = all_reduce_lazy(plan.scratch_views, flat, plan.small_idx, self.group, async_op=False)
                finish()
            else:
                all_reduce_buckets(flat, plan.buckets, self.group, async_op=False)
            return
        works = []
        with torch.cuda.stream(self.comm_stream):
            assert len(self.events) >= len(plan.buckets), "backward() was not given this plan's bucket events"
            finish = None
            for b, (ev, (off, numel)) in enumerate(zip(self.events, plan.buckets)):
                self.comm_stream.wait_event(ev)
                if lazy:
                    v = plan.scratch_views[b]
                    if v.numel():
                        works.append(_WireWork(v, self.group, True, wire_dtype()))
                    if b == len(plan.buckets) - 1:      # everything else, once the whole backward is done
                        w2, finish = all_reduce_lazy([], flat, plan.small_idx, self.group, async_op=True)
                        works += w2
                else:
                    works += all_reduce_buckets(flat, [(off, numel)], self.group, async_op=True)
            for w in works:
                w.wait()                 # comm stream waits for RCCL
            if finish is not None:
                finish()
        cur.wait_stream(self.comm_stream)

    def all_reduce_scalars(self, loss_sum, count):
        t = torch.tensor([loss_sum, float(count)], dtype=torch.float64, device=self.engine.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return float(t[0].item()), int(round(t[1].item()))
