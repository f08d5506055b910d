"""Data-parallel gradient exchange over flat fp32 arenas (RCCL over xGMI on MI355X; gloo on CPU for tests).

The reference gets data parallelism from torch DDP inside BasicSR: bucketed
NCCL all-reduce of G grads during G backward, TWO all-reduces of D grads (one per backward()), a
broadcast of the spectral-norm u/v buffers before each D forward and a reduce of 7 loss scalars.
MI355X-first re-design:
  * gradients live in ONE flat arena per network, so the exchange is a handful of large all-reduces
    (chunked so RCCL can pipeline them over the 7 xGMI links) issued on a SIDE stream;
  * the generator exchange is launched as soon as G's backward retires and overlaps the whole
    discriminator phase (D real+fake forward/backward do not depend on the updated G weights);
    G's Adam update waits only for G's exchange (an event behind it on the side stream) and runs under D's;
  * D grads are reduced ONCE after both backward passes (avg_real+avg_fake == avg(real+fake));
  * u/v stay bit-identical across ranks without any broadcast: every rank applies the same
    deterministic power iteration to the same (all-reduced) weights;
  * averaging (1/world) is folded into the fused Adam kernel's grad_scale.

Exchange algorithm (SSR_DP_ALGO, read when the context is built; a one-command A/B for the first session on an 8-GPU node):
  allreduce (default)  chunked dist.all_reduce: RCCL picks ring / tree per message size;
  rsag                 explicit reduce_scatter_tensor + all_gather_into_tensor over the (padded) arena: each rank reduces 1/world
                       of the arena and gathers the rest - the direct all-links form SURVEY.md section 5 prefers on point-to-point
                       xGMI (7 links x ~153 GB/s per GPU), same sums up to the reduction order inside RCCL.
Diagnostics (DPContext.timing = True; bench.py --gpus N switches it on): an event pair around every exchange on the comm stream ->
comm_busy_ms(), so that a scaling result can be split into compute, exposed communication and host enqueue time from one run.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def init_distributed(backend: Optional[str] = None) -> "DPContext":
    """One process per GPU, env:// rendezvous (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*), mirroring
    basicsr.utils.dist_util.init_dist('pytorch') as called from /root/reference/ssr/utils/options.py:65-74."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # SSR_DP_FORCE=1: run the data-parallel branch (phase graphs, side-stream exchanges over the real backend) even with
    # one rank.  An all-reduce over one rank is the identity, so the result equals the single-process step; this is how the
    # RCCL code path is exercised on a one-GPU box (tests/test_dp_gpu.py).
    force = os.environ.get("SSR_DP_FORCE", "0") == "1" and "RANK" in os.environ
    if world <= 1 and not force:
        return DPContext(None, 0, 1)
    rank = int(os.environ["RANK"])
    local = int(os.environ.get("LOCAL_RANK", rank))
    if backend is None:   # SSR_DIST_BACKEND=gloo: test hook (two ranks sharing one GPU, tests/test_dp_gpu.py; RCCL needs one device per rank)
        backend = os.environ.get("SSR_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(local % torch.cuda.device_count())
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return DPContext(dist.group.WORLD, rank, world, force=force)


class DPContext:
    def __init__(self, group, rank: int, world: int, chunk_bytes: int = 32 << 20, force: bool = False, algo: Optional[str] = None):
        self.group, self.rank, self.world = group, rank, world
        self.force = force
        self.chunk_elems = max(1, chunk_bytes // 4)
        self.algo = (algo or os.environ.get("SSR_DP_ALGO", "allreduce")).lower()
        if self.algo not in ("allreduce", "rsag"):
            raise ValueError(f"SSR_DP_ALGO={self.algo!r}: expected 'allreduce' or 'rsag'")
        self._comm_stream = None
        self._pending: List = []
        self._scratch = {}          # rsag: padded staging arenas by (numel, device)
        self.timing = False         # event pairs around the exchanges on the comm stream (comm_busy_ms)
        self._timed: List = []

    @property
    def active(self) -> bool:
        return self.world > 1 or self.force

    @property
    def grad_scale(self) -> float:
        """DDP averages gradients: fold 1/world into the optimizer kernel."""
        return 1.0 / self.world

    def _stream(self):
        if self._comm_stream is None and torch.cuda.is_available():
            self._comm_stream = torch.cuda.Stream()
        return self._comm_stream

    def all_reduce_async(self, flat: torch.Tensor):
        """Sum-all-reduce a flat arena in chunks.  On GPU the collectives are enqueued on a side stream
        that first waits for the producer (current) stream.  Returns a handle for wait(handle): an event
        recorded behind this arena's collectives (GPU) / its work objects (CPU), so that the consumer of
        an earlier exchange does not have to wait for a later one."""
        if not self.active:
            return None
        if flat.is_cuda:
            cs = self._stream()
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                if self.timing:
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(cs)
                if self.algo == "rsag":
                    self._rsag(flat)
                else:
                    for off in range(0, flat.numel(), self.chunk_elems):
                        dist.all_reduce(flat[off:off + self.chunk_elems], op=dist.ReduceOp.SUM, group=self.group)
                ev = torch.cuda.Event(enable_timing=self.timing)
                ev.record(cs)
                if self.timing:
                    self._timed.append((e0, ev, flat.numel() * flat.element_size()))
            return ev
        if self.algo == "rsag":
            self._rsag(flat)
            return []
        works = [dist.all_reduce(flat[off:off + self.chunk_elems], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for off in range(0, flat.numel(), self.chunk_elems)]
        self._pending.extend(works)
        return works

    def _rsag(self, flat: torch.Tensor):
        """sum-all-reduce of a flat arena as reduce-scatter + all-gather (in place; on the current stream).  The arena is used in
        place when its length is a multiple of the world size, else through a zero-padded staging copy."""
        n, w = flat.numel(), self.world
        per = (n + w - 1) // w
        if per * w == n:
            buf = flat
        else:
            key = (per * w, flat.device, flat.dtype)
            buf = self._scratch.get(key)
            if buf is None:
                buf = self._scratch[key] = torch.zeros(per * w, dtype=flat.dtype, device=flat.device)
            buf[:n].copy_(flat)
        shard = buf[self.rank * per:(self.rank + 1) * per]
        if w == 1:
            return
        # in-place reduce-scatter into this rank's shard of the same buffer is not allowed by every backend: a shard-sized temporary
        tmp = torch.empty_like(shard)
        if hasattr(dist, "reduce_scatter_tensor") and dist.get_backend(self.group) != "gloo":
            dist.reduce_scatter_tensor(tmp, buf, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(buf, tmp, group=self.group)
        else:   # gloo (CPU tests) has no reduce_scatter: the same data movement from primitives it has - reduce each shard to its owner
            for r in range(w):
                dist.reduce(buf[r * per:(r + 1) * per], dst=r, op=dist.ReduceOp.SUM, group=self.group)
            tmp.copy_(shard)
            parts = [torch.empty_like(tmp) for _ in range(w)]
            dist.all_gather(parts, tmp, group=self.group)
            for r in range(w):
                buf[r * per:(r + 1) * per].copy_(parts[r])
        if buf is not flat:
            flat.copy_(buf[:n])

    def comm_busy_ms(self, reset: bool = True):
        """(total ms the comm stream spent inside exchanges, bytes exchanged, exchanges) since the last reset; needs timing = True and a
        device synchronisation by the caller"""
        ms = sum(a.elapsed_time(b) for a, b, _ in self._timed)
        nbytes = sum(c for _, _, c in self._timed)
        k = len(self._timed)
        if reset:
            self._timed.clear()
        return ms, nbytes, k

    def wait(self, handle=None):
        """Make the current stream (GPU) / the caller (CPU) wait for one exchange (`handle` from
        all_reduce_async) or, without a handle, for everything issued so far."""
        if not self.active:
            return
        if handle is not None:
            if isinstance(handle, list):
                for w in handle:
                    w.wait()
                self._pending = [w for w in self._pending if all(w is not h for h in handle)]
            else:
                torch.cuda.current_stream().wait_event(handle)
            return
        for w in self._pending:
            w.wait()
        self._pending.clear()
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)

    def reduce_scalars(self, t: torch.Tensor) -> torch.Tensor:
        """reduce_loss_dict (BasicSR): mean over ranks of the logged scalars (rank 0 reads them)."""
        if not self.active:
            return t
        t = t.clone()
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t / self.world

    def broadcast_(self, t: torch.Tensor, src: int = 0):
        """DDP constructor semantics: parameters/buffers start identical to rank 0's."""
        if self.active:
            dist.broadcast(t, src=src, group=self.group)

    def barrier(self):
        if self.active:
            dist.barrier(group=self.group)
