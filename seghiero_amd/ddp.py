"""Data-parallel runtime: one process per GPU, gradients all-reduced over RCCL (xGMI).

The reference has no data parallelism at all (``train.py`` never calls ``init_process_group``) -- this is
new functionality required by the north star.  Sharding: rank r takes its own minibatch of B images (weak scaling),
parameters are replicated, every rank computes its local loss with local normalisers (standard DDP semantics = mean of
per-rank losses), and the only data-path exchange per step is the gradient all-reduce below (+ the 4-byte triplet
``class_count`` MIN-reduce inside the loss, ``hiera_triplet_loss.py:193-198``).

``GradSync`` keeps one flat fp32 arena for all gradients, split into buckets of ~``bucket_mb`` in backward order
(aux head, ASPP head, backbone from layer4 down to the stem).  The hand-scheduled backward nodes hand finished
gradients over while backward is still running (``early_flush``: end of the aux / head nodes, end of every ResNet
stage); each bucket that becomes complete is all-reduced with ONE RCCL call on a side HIP stream, overlapped with the
remaining backward kernels; the 1/world scaling is folded into the SGD kernel (``grad_scale``) instead of a separate pass.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun contract).
    Returns (rank, local_rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if dist.is_initialized():
        init_small_group()
    return rank, local, world


# Latency-class collectives (SyncBN statistics, coalesced per block: see ops.sync_batch; the 4-byte triplet class_count MIN-reduce)
# CAN get a process group -- i.e. an RCCL communicator and internal stream -- of their own, so that a stat reduction on the
# critical path of forward / backward never queues behind a 32 MB gradient bucket of the default group.  OFF by default
# (SEGHIERO_SMALL_GROUP=1 switches it on): torch.distributed documents concurrent collectives of two NCCL communicators on one device
# as unsafe unless one group's work has finished on the device before the other's is enqueued, and this path has never run on two
# GPUs (the round's box has one).  With the switch on, every small-group collective first makes its stream wait for an event
# recorded after the last gradient-bucket launch (_BUCKET_EVENT), which serialises the two communicators on the device -- the order
# torch.distributed asks for -- at the price of the very queueing the second group was meant to avoid; without the switch the small
# collectives simply use the default group (same communicator: RCCL orders them itself).
_SMALL = None
_BUCKET_EVENT = None           # recorded on the gradient-exchange stream after the latest bucket all-reduce (GradSync._launch)
SMALL_GROUP = os.environ.get("SEGHIERO_SMALL_GROUP", "0") == "1"
FORCE_COLLECTIVES = False      # tests: run every collective even at world size 1 (executes the RCCL calls on a one-GPU box)


def init_small_group():
    """Collective: every rank must call it (init_from_env and GradSync.__init__ do).  A no-op unless SMALL_GROUP is set."""
    global _SMALL
    if SMALL_GROUP and _SMALL is None and dist.is_available() and dist.is_initialized():
        _SMALL = dist.new_group()
    return _SMALL


def small_group():
    return _SMALL


def shutdown():
    global _SMALL, _ACTIVE, _BUCKET_EVENT
    _SMALL = None
    _ACTIVE = None
    _BUCKET_EVENT = None
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def collectives_on():
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or FORCE_COLLECTIVES)


def all_reduce_small(t, op=None):
    """All-reduce of a latency-critical small tensor on the small-message group (the default group unless SMALL_GROUP is on)."""
    if _SMALL is not None and _BUCKET_EVENT is not None and t.is_cuda:
        torch.cuda.current_stream(t.device).wait_event(_BUCKET_EVENT)      # never two communicators at work on the device at once
    return all_reduce(t, op=op, group=_SMALL)


def _via_host(group=None):
    """gloo has no device collectives in this build: stage through host memory (rehearsal / CPU tests only;
    the production backend is "nccl" = RCCL, which works on device buffers directly)."""
    return dist.get_backend(group) == "gloo"


def all_reduce(t, op=None, group=None, async_op=False):
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and _via_host(group):
        h = t.detach().cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
        return None
    return dist.all_reduce(t, op=op, group=group, async_op=async_op)


def broadcast_module_state(modules, src=0):
    """Make every rank start from rank `src`'s parameters and buffers (weights, BN running stats, `step`)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            if t.is_cuda and _via_host():
                h = t.detach().cpu()
                dist.broadcast(h, src=src)
                t.data.copy_(h)
            else:
                dist.broadcast(t.data, src=src)


# EXACT data-parallel mode (SURVEY 8e "optional exact mode"): the loss normalisers -- valid-pixel counts of the hierarchical BCE terms
# (hiera_triplet_loss.py:41-107), the all-pixel count of the CE wrapper (utils.py:20-21), the aux CE's valid count -- are all-reduced
# (ONE 24-byte message per step) and every rank divides its LOCAL numerators by the GLOBAL denominators: the per-rank losses then SUM to the
# full-batch loss, gradients are summed (not averaged) over the ranks, and with SyncBN on a sharded step equals the single-process
# step on the whole batch (strong scaling: global batch 16 as 8 x 2).  The triplet term stays per rank as the reference's DDP code
# intended (tree_triplet_loss.py:23-46 selects inside the local shard) and enters with weight 1 / world.  2-level loss + aux CE; the
# 3-level RMI loss keeps local normalisers.
EXACT = False
_COUNTS = None          # the all-reduced normalisers of the step in flight (device int64[3]), set by exact_counts()


def exact_counts(labels8, n_fine, hiera_index):
    """Exact mode: all-reduce this step's loss normalisers; the loss modules pick them up (current_counts) until end_step()."""
    global _COUNTS
    if not (EXACT and collectives_on()):
        _COUNTS = None
        return None
    from . import ops
    c = ops.label_counts(labels8, n_fine, hiera_index)
    all_reduce_small(c)
    _COUNTS = c
    return c


def current_counts():
    return _COUNTS


def end_step():
    global _COUNTS
    _COUNTS = None


_ACTIVE = None          # the GradSync that is collecting gradients of the backward pass in flight (set by begin())


def grad_buffer(p):
    """Where a backward node should write the gradient of parameter `p`: its slice of the active GradSync arena, else None.
    A slice is handed out ONCE per step: a second gradient of the same parameter (layers.GradMap.put accumulates) gets None, i.e. a
    fresh tensor, instead of overwriting the first one in place."""
    if _ACTIVE is not None and id(p) in _ACTIVE.views and id(p) not in _ACTIVE._in_arena and id(p) not in _ACTIVE._handed:
        _ACTIVE._handed.add(id(p))
        return _ACTIVE.grad_view(p)
    return None


def early_flush(pairs):
    """Called from inside the hand-scheduled backward nodes with (parameter, final gradient) pairs: lets the gradient
    exchange of finished buckets start while the rest of the backward is still running.  No-op without an active sync."""
    if _ACTIVE is not None:
        _ACTIVE.early(pairs)


class GradSync:
    """Bucketed gradient all-reduce over one flat fp32 arena laid out in backward order (aux, head, layer4 ... stem).

    ``begin()`` before ``loss.backward()``; the backward nodes hand finished gradients to ``early()`` (via ``early_flush``)
    at a few points -- end of the aux / head nodes, end of each ResNet stage -- which copies them into the arena and launches
    the all-reduce of every bucket that became complete on a side stream, overlapped with the remaining backward kernels;
    ``reduce()`` after backward sends what is left, waits, and points ``p.grad`` at the reduced arena (no copy back)."""

    ARENA_ALIGN = 64

    def __init__(self, params, bucket_mb=32.0, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.on = dist.is_initialized() and (self.world > 1 or FORCE_COLLECTIVES)
        if dist.is_initialized():
            init_small_group()
        # backward produces gradients roughly in reverse parameter order
        self.order = list(reversed(list(params)))
        dev = self.order[0].device
        # every parameter's slice starts on a 256-byte boundary (ARENA_ALIGN floats): the weight-gradient kernels write their
        # slice with 16-byte stores (odd-sized tensors -- cls_seg bias 13, aux BatchNorm 9 -- would misalign everything behind them);
        # the padding floats stay zero and travel with the bucket
        al = self.ARENA_ALIGN
        total = sum(-(-p.numel() // al) * al for p in self.order)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views, self.buckets, self.bucket_of = {}, [], {}
        off, start, limit = 0, 0, int(bucket_mb * (1 << 20) / 4)
        cur = []
        for p in self.order:
            self.views[id(p)] = (off, p.numel())
            self.bucket_of[id(p)] = len(self.buckets)
            cur.append(p)
            off += -(-p.numel() // al) * al
            if off - start >= limit:
                self.buckets.append((start, off, cur))
                start, cur = off, []
        if cur:
            self.buckets.append((start, off, cur))
        self.stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._in_arena, self._launched, self._works, self._handed = set(), set(), [], set()

    # ------------------------------------------------------------------ per-step protocol
    def begin(self):
        global _ACTIVE
        self._in_arena, self._launched, self._works, self._handed = set(), set(), [], set()
        _ACTIVE = self if self.on else None

    def grad_view(self, p):
        """The arena slice of parameter `p` shaped / strided like `p`: the backward kernels write weight gradients straight
        into it (layers.new_grad), so nothing is copied when the bucket leaves."""
        o, n = self.views[id(p)]
        return self.flat[o:o + n].as_strided(p.shape, p.stride())

    def _stage(self, pairs):
        srcs, dsts = [], []
        for p, g in pairs:
            if g is None or id(p) not in self.views or id(p) in self._in_arena:
                continue
            o, n = self.views[id(p)]
            self._in_arena.add(id(p))
            if g.data_ptr() == self.flat.data_ptr() + 4 * o and g.shape == p.shape and g.stride() == p.stride():
                continue                                  # already written in place by the backward kernels
            if g.shape != p.shape:
                g = g.reshape(p.shape)
            if g.stride() != p.stride():
                g = torch.empty_like(p).copy_(g)
            srcs.append(g.as_strided((n,), (1,)) if not g.is_contiguous() else g.reshape(-1))
            dsts.append(self.flat[o:o + n])
        if srcs:
            torch._foreach_copy_(dsts, srcs)

    def _launch(self, bi):
        global _BUCKET_EVENT
        start, end, _ = self.buckets[bi]
        buf = self.flat[start:end]
        self._launched.add(bi)
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                work = all_reduce(buf, group=self.group, async_op=True)
                self._works.append(work)
                if _SMALL is not None:
                    if work is not None:
                        work.wait()                      # (device-side: the exchange stream waits for the collective, not the host)
                    _BUCKET_EVENT = torch.cuda.Event()
                    _BUCKET_EVENT.record(self.stream)
        else:
            self._works.append(all_reduce(buf, group=self.group, async_op=True))

    def early(self, pairs):
        if not self.on:
            return
        self._stage(pairs)
        for bi, (_, _, plist) in enumerate(self.buckets):
            if bi not in self._launched and all(id(p) in self._in_arena for p in plist):
                self._launch(bi)

    def reduce(self, params):
        """All-reduce (sum) every gradient in place; returns the scale (1/world) the optimizer must apply."""
        global _ACTIVE
        _ACTIVE = None
        if not self.on:
            return 1.0
        self._stage([(p, p.grad) for p in params])
        for bi, (_, _, plist) in enumerate(self.buckets):
            if bi not in self._launched and any(id(p) in self._in_arena for p in plist):
                self._launch(bi)
        for w in self._works:
            if w is not None:
                w.wait()
        if self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)
        for p in params:                                # point the grads at the reduced arena (no copy back)
            if p.grad is not None and id(p) in self._in_arena:
                o, n = self.views[id(p)]
                p.grad = self.flat[o:o + n].as_strided(p.shape, p.stride())
        self._works = []
        return 1.0 if EXACT else 1.0 / self.world      # exact mode: every rank's loss already carries the global denominators -- gradients are summed
