"""Step arena: the large device buffers of one training step, allocated ONCE per module and reused by every step.

Why: the step touches three HIP streams.  A block that torch's caching allocator handed out on one
stream and that was used on another can only be reused after an event on the second stream has completed, and with the
host enqueueing a step or more ahead of the GPU, whether that event has completed when the next step asks for the same
size depends on host timing.  When it has not, the allocator calls hipMalloc for another multi-GB block in the middle
of a step.  Round 2's driver run measured 20.5 ms/step against 18.2 on the builder's boxes with the whole excess inside
the bracket that allocated two 4.15 GB buffers.  Buffers taken from the arena never go back to the allocator: no
hipMalloc, no hipFree, no event bookkeeping, the same addresses every step (which is also what a HIP graph needs).

Safety.  Reuse across steps relies on stream ORDER, not on events: every step begins with `side.wait_stream(main)` and
its backward ends with main joined to every side stream, so all work of step k that touches an arena buffer is ordered
before any work of step k+1 (model/ifnet.py).  Two forward passes whose backward passes are both still pending cannot
share the buffers: the arena is LEASED by a forward pass and handed back by its backward (or when the autograd graph is
dropped); a forward that finds the arena leased falls back to ordinary allocations.
"""
import torch


class _Lease:
    """Held by the autograd context of the forward pass that uses the arena; released by its backward, or by the
    garbage collector when the graph is dropped without one."""

    def __init__(self, arena):
        self.arena = arena
        self.on_release = []      # callbacks: state that is only valid while this forward's graph is (prepared weight planes)

    def release(self):
        a, self.arena = self.arena, None
        if a is not None:
            a._leased = False
            cbs, self.on_release = self.on_release, []
            for cb in cbs:
                cb()

    def __del__(self):
        self.release()


class StepArena:
    def __init__(self):
        self._bufs = {}
        self._leased = False
        self.grown = 0          # how often a buffer had to be (re)allocated: constant once the shapes have been seen

    def lease(self):
        """-> a _Lease, or None when a previous forward still holds the arena."""
        if self._leased:
            return None
        self._leased = True
        return _Lease(self)

    def get(self, name, shape, dtype, device):
        """A view of the persistent buffer `name` with the given shape (contents are whatever the last user left)."""
        n = 1
        for s in shape:
            n *= int(s)
        t = self._bufs.get(name)
        if t is None or t.dtype != dtype or t.device != torch.device(device) or t.numel() < n:
            if t is not None:
                # the old buffer may still be read by queued work on any stream: drain the device before it is freed
                torch.cuda.synchronize(t.device)
            t = torch.empty(max(n, 1), dtype=dtype, device=device)
            self._bufs[name] = t
            self.grown += 1
        return t[:n].view(*shape)

    def nbytes(self):
        return sum(t.numel() * t.element_size() for t in self._bufs.values())

    def owns(self, t):
        """Does tensor `t` live in one of the arena's buffers?  (Arena memory never returns to an allocator pool, so a
        tensor it owns needs no record_stream bookkeeping when another stream reads it; any other tensor does.)"""
        p = t.data_ptr()
        for b in self._bufs.values():
            q = b.data_ptr()
            if q <= p < q + b.numel() * b.element_size():
                return True
        return False

    def release(self):
        """Free every buffer (after draining the device: queued work on any stream may still read them).  The next
        training step allocates them again.  Refused while a forward pass holds the lease."""
        if self._leased:
            raise RuntimeError("StepArena.release: a forward pass whose backward is pending still holds the arena")
        if self._bufs:
            for dev in {t.device for t in self._bufs.values() if t.is_cuda}:
                torch.cuda.synchronize(dev)
            self._bufs.clear()


def alloc(arena, name, shape, dtype, device):
    """torch.empty from the arena when there is one, from the caching allocator otherwise."""
    if arena is not None:
        return arena.get(name, shape, dtype, device)
    return torch.empty(*shape, dtype=dtype, device=device)
