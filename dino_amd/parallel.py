"""Data-parallel fine-tuning over the GPUs of one node.

One process per GPU (``torch.distributed``; backend "nccl" = RCCL over xGMI on ROCm, "gloo" on CPU for tests).
Frames are independent, so the global batch is split contiguously, every rank runs the native training step on
its slice, and the only exchange is one sum-all-reduce of the gradients per optimiser step, divided by the world
size (valid because every rank holds the same number of patches and the loss is a mean,
pl_torch_modules.py:265).  The reference itself is single-GPU (``Trainer(gpus=1)``, pl_torch_modules.py:396,417):
the parity target is "equal to one process on the full batch" up to fp32 round-off.

Gradients are packed into a few flat fp32 buckets in reverse parameter order (head first, embeddings last: the
order backward produces them), each reduced with one collective: large messages keep all 7 xGMI links of the
full mesh busy, and 5 buckets of <= 8 MiB cover the 22.1 MiB of the unfrozen 3-block model.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_bounds(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) slice of the global batch owned by `rank`; requires equal shares."""
    if global_batch % world != 0:
        raise ValueError(f"global batch {global_batch} is not divisible by world size {world}")
    per = global_batch // world
    return rank * per, (rank + 1) * per


def shard_batch(x: torch.Tensor, y: torch.Tensor, rank: int, world: int):
    lo, hi = shard_bounds(x.shape[0], rank, world)
    return x[lo:hi], y[lo:hi]


def make_flat_buckets(named_params: Sequence[Tuple[str, torch.Tensor]], stage_of: Callable[[str], int] = None,
                      bucket_bytes: int = 8 << 20) -> dict:
    """Flat fp32 gradient buckets with one view per parameter.

    Parameters are taken in REVERSE registration order -- head, final norm + last block, ..., embeddings: the order the
    backward finishes them -- and packed into flat buffers of at most `bucket_bytes`; every view starts on a 256-byte
    boundary.  The gradient kernels write straight into the views and the all-reduce works on the flat buffers: no
    torch.cat, no copy-back.  Returns {'buckets': [{'flat', 'names', 'stage'}], 'views': {name: view}}; 'stage' is the last
    backward stage any member waits for (DINOSeg.grad_stage / dinoseg_stream_wait_grad_stage)."""
    groups, cur, size = [], [], 0
    for n, p in reversed(list(named_params)):
        padded = (p.numel() + 63) // 64 * 64
        if cur and (size + padded) * 4 > bucket_bytes:
            groups.append(cur)
            cur, size = [], 0
        cur.append((n, p, size))
        size += padded
    if cur:
        groups.append(cur)
    buckets, views = [], {}
    for g in groups:
        n_last, p_last, off_last = g[-1]
        flat = torch.zeros(off_last + (p_last.numel() + 63) // 64 * 64, dtype=torch.float32, device=p_last.device)
        names = []
        for n, p, off in g:
            views[n] = flat[off:off + p.numel()].view_as(p)
            names.append(n)
        buckets.append({"flat": flat, "names": names, "stage": max(stage_of(n) for n in names) if stage_of else 0})
    return {"buckets": buckets, "views": views}


def make_buckets(named_grads: Sequence[Tuple[str, torch.Tensor]], bucket_bytes: int = 8 << 20) -> List[List[Tuple[str, torch.Tensor]]]:
    """Group gradients, in REVERSE registration order, into buckets of about `bucket_bytes`."""
    buckets, cur, size = [], [], 0
    for name, g in reversed(list(named_grads)):
        nbytes = g.numel() * g.element_size()
        if cur and size + nbytes > bucket_bytes:
            buckets.append(cur)
            cur, size = [], 0
        cur.append((name, g))
        size += nbytes
    if cur:
        buckets.append(cur)
    return buckets


def allreduce_gradients(named_grads: Iterable[Tuple[str, torch.Tensor]], world: int = None, bucket_bytes: int = 8 << 20,
                        average: bool = True, group=None) -> int:
    """Sum-all-reduce the gradients bucket by bucket (fixed order => deterministic reduction), then divide by the
    world size.  Returns the number of collectives issued."""
    named = [(n, g) for n, g in named_grads if g is not None]
    if not dist.is_initialized() or (world or dist.get_world_size(group)) == 1:
        return 0
    world = world or dist.get_world_size(group)
    n_coll = 0
    for bucket in make_buckets(named, bucket_bytes):
        flat = torch.cat([g.reshape(-1) for _, g in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        if average:
            flat.div_(world)
        off = 0
        for _, g in bucket:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        n_coll += 1
    return n_coll


class DataParallelFineTuner:
    """Fine-tune loop body for one rank: shard -> native fused training step -> bucketed gradient all-reduce overlapped with the
    rest of backward -> optimiser.

    Stands in for the Lightning ``Trainer.fit`` inner loop of the reference (pl_torch_modules.py:365-432), which is out
    of scope; metrics / checkpointing callbacks are not reproduced.

    The model's ``.grad`` tensors are views into a few flat buckets (``model.grad_buckets()``).  After the step has been
    enqueued, each bucket is reduced with ONE collective on a side stream that waits (on the device, not the host) for the
    backward stage that finishes the bucket -- head first, embeddings last -- so the reduction of the head / last-block
    buckets runs under the backward of the earlier blocks; the compute stream then waits for the collectives before the
    optimiser.  ``collective='rs_ag'`` splits each all-reduce into reduce-scatter + all-gather (every rank reduces 1/world of
    the bucket: on the point-to-point xGMI mesh all 7 links of a GPU carry 1/8 of the bucket per phase instead of a ring's
    single-link 2*(7/8) of it).  Neither form has been timed on a multi-GPU node from here.

    `model` needs: fused_training_step(batch) -> {'loss': ...}, grad_buckets() -> list of {'flat','names','stage'},
    stream_wait_grad_stage(stage, stream) (may be a no-op), fused_adam_step(grad_scale=...) or configure_optimizers()."""

    def __init__(self, model, fused_optimizer: bool = True, bucket_bytes: int = 8 << 20, collective: str = "allreduce",
                 overlap: bool = True, group=None):
        if collective not in ("allreduce", "rs_ag"):
            raise ValueError("collective must be 'allreduce' or 'rs_ag'")
        self.model = model
        self.fused = fused_optimizer
        self.bucket_bytes = bucket_bytes
        self.collective = collective
        self.overlap = overlap
        self.group = group
        # the model's step binds the parameters' .grad to flat buckets of ITS current bucket size: fix that size before the first
        # step, or the first reduce would build (and reduce) a second, unbound set of buckets
        model.grad_buckets(bucket_bytes)
        self.torch_opt = None if fused_optimizer else model.configure_optimizers()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self._comm_stream = None
        self.last_collectives = 0

    def _reduce_bucket(self, flat: torch.Tensor):
        """Enqueue the sum-reduction of one flat bucket; returns the async work handles."""
        if self.collective == "allreduce" or flat.numel() % self.world != 0:
            return [dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        shard = flat.numel() // self.world
        mine = flat[self.rank * shard:(self.rank + 1) * shard]
        w1 = dist.reduce_scatter_tensor(mine, flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        w1.wait()        # orders the all-gather behind the reduce-scatter on the current (communication) stream
        return [dist.all_gather_into_tensor(flat, mine, group=self.group, async_op=True)]

    def check_bound(self, buckets) -> None:
        """The buckets about to be reduced must be the memory the step wrote its gradients into (every parameter's .grad is a
        view into them): reducing a freshly built, unbound set would silently average zeros."""
        named = dict(self.model.named_parameters()) if hasattr(self.model, "named_parameters") else {}
        for b in buckets:
            lo = b["flat"].data_ptr()
            hi = lo + b["flat"].numel() * b["flat"].element_size()
            for name in b["names"]:
                p = named.get(name)
                if p is not None and p.requires_grad and (p.grad is None or not lo <= p.grad.data_ptr() < hi):
                    raise RuntimeError(f"the .grad of '{name}' does not live in the bucket about to be reduced: the step and the "
                                       f"reduction disagree about the bucket layout (bucket_bytes)")

    def reduce_gradients(self) -> None:
        """Sum the gradient buckets over the ranks (the mean's 1/world is applied by the optimiser step)."""
        self.last_collectives = 0
        if self.world == 1:
            return
        buckets = self.model.grad_buckets(self.bucket_bytes)
        self.check_bound(buckets)
        on_gpu = buckets and buckets[0]["flat"].is_cuda
        works = []
        if on_gpu and self.overlap:
            dev = buckets[0]["flat"].device
            if self._comm_stream is None or self._comm_stream.device != dev:
                self._comm_stream = torch.cuda.Stream(device=dev)
            for b in buckets:       # reverse registration order = the order backward finishes them
                self.model.stream_wait_grad_stage(b["stage"], self._comm_stream)
                with torch.cuda.stream(self._comm_stream):
                    works += self._reduce_bucket(b["flat"])
                self.last_collectives += 1
        else:
            for b in buckets:
                works += self._reduce_bucket(b["flat"])
                self.last_collectives += 1
        for w in works:
            w.wait()                # GPU: the compute stream waits for the collective; CPU (gloo): blocks the host

    def step(self, x_global: torch.Tensor, y_global: torch.Tensor) -> torch.Tensor:
        x, y = shard_batch(x_global, y_global, self.rank, self.world)
        out = self.model.fused_training_step((x, y), 0)
        self.reduce_gradients()
        if self.fused:
            self.model.fused_adam_step(grad_scale=1.0 / self.world)
        else:
            if self.world > 1:
                for b in self.model.grad_buckets(self.bucket_bytes):
                    b["flat"].div_(self.world)
            self.torch_opt.step()
        loss = out["loss"].detach().clone()
        if self.world > 1:
            dist.all_reduce(loss, op=dist.ReduceOp.SUM, group=self.group)
            loss /= self.world
        return loss
