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
    return rank, local, world


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


_ACTIVE = None          # the GradSync that is collecting gradients of the backward pass in flight (set by begin())


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

    def __init__(self, params, bucket_mb=32.0, group=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # backward produces gradients roughly in reverse parameter order
        self.order = list(reversed(list(params)))
        dev = self.order[0].device
        total = sum(p.numel() for p in self.order)
        self.flat = torch.zeros(total, device=dev, dtype=torch.float32)
        self.views, self.buckets, self.bucket_of = {}, [], {}
        off, start, limit = 0, 0, int(bucket_mb * (1 << 20) / 4)
        cur = []
        for p in self.order:
            self.views[id(p)] = (off, p.numel())
            self.bucket_of[id(p)] = len(self.buckets)
            cur.append(p)
            off += p.numel()
            if off - start >= limit:
                self.buckets.append((start, off, cur))
                start, cur = off, []
        if cur:
            self.buckets.append((start, off, cur))
        self.stream = torch.cuda.Stream(device=dev) if dev.type == "cuda" else None
        self._in_arena, self._launched, self._works = set(), set(), []

    # ------------------------------------------------------------------ per-step protocol
    def begin(self):
        global _ACTIVE
        self._in_arena, self._launched, self._works = set(), set(), []
        _ACTIVE = self if self.world > 1 else None

    def _stage(self, pairs):
        srcs, dsts = [], []
        for p, g in pairs:
            if g is None or id(p) not in self.views or id(p) in self._in_arena:
                continue
            o, n = self.views[id(p)]
            if g.shape != p.shape:
                g = g.reshape(p.shape)
            if g.stride() != p.stride():
                g = torch.empty_like(p).copy_(g)
            srcs.append(g.as_strided((n,), (1,)) if not g.is_contiguous() else g.reshape(-1))
            dsts.append(self.flat[o:o + n])
            self._in_arena.add(id(p))
        if srcs:
            torch._foreach_copy_(dsts, srcs)

    def _launch(self, bi):
        start, end, _ = self.buckets[bi]
        buf = self.flat[start:end]
        self._launched.add(bi)
        if self.stream is not None:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                self._works.append(all_reduce(buf, group=self.group, async_op=True))
        else:
            self._works.append(all_reduce(buf, group=self.group, async_op=True))

    def early(self, pairs):
        if self.world == 1:
            return
        self._stage(pairs)
        for bi, (_, _, plist) in enumerate(self.buckets):
            if bi not in self._launched and all(id(p) in self._in_arena for p in plist):
                self._launch(bi)

    def reduce(self, params):
        """All-reduce (sum) every gradient in place; returns the scale (1/world) the optimizer must apply."""
        global _ACTIVE
        _ACTIVE = None
        if self.world == 1:
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
        return 1.0 / self.world
