"""Batch sharding over the GPUs of one node.

Image pairs are independent all the way through the path (the batch index is
the outermost index of every kernel), so the batch is cut into contiguous
slices, one per rank, with NO collective on the data path.  The only exchange
is the one the reference performs in ``average_gradients``
(``src/e2eflow/core/train.py:446-480``): a sum of the per-rank loss scalars and
of the flat gradient buffer -- here one RCCL all-reduce each
(``torch.distributed`` backend ``nccl`` is RCCL on ROCm; ``gloo`` on CPU tests).

Every loss of ``unflow_amd.losses`` takes the GLOBAL normalisation
(``norm_batch`` / ``norm_count``), so the SUM over ranks equals the single-GPU
value of the whole batch.
"""
import os

import torch
import torch.distributed as dist


def init_process_group(backend=None):
    """One process per GPU; rank / world size / master address from the launcher's env."""
    if dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if world == 1:
        return 0, 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kwargs = {}
    if backend == "nccl":
        local = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(local)
        kwargs["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world


def shard_range(global_batch, rank, world):
    """Contiguous slice [start, stop) of the batch owned by ``rank``; the first
    ``global_batch % world`` ranks take one extra pair (ragged batches)."""
    if not 0 <= rank < world:
        raise ValueError("rank %d outside world of %d" % (rank, world))
    base, extra = divmod(global_batch, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def shard(tensor, rank, world):
    start, stop = shard_range(tensor.shape[0], rank, world)
    return tensor[start:stop]


def all_reduce_sum_(*tensors):
    """In-place SUM all-reduce of each tensor (no-op for a single process)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        for t in tensors:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tensors


def reduce_losses(losses):
    """Sum a dict of per-rank loss scalars into the global losses with ONE
    all-reduce (the scalars are packed into one buffer)."""
    keys = sorted(losses)
    if not keys:
        return {}
    packed = torch.stack([losses[k].detach().reshape(()) for k in keys])
    all_reduce_sum_(packed)
    return {k: packed[i] for i, k in enumerate(keys)}


class FlatGradBuffer:
    """The flat fp32 gradient buffer that stands in for the network weights:
    one large all-reduce over xGMI instead of one per tensor."""

    def __init__(self, numel, device):
        self.buffer = torch.zeros(int(numel), device=device, dtype=torch.float32)
        self._comm_stream = None

    def all_reduce(self):
        all_reduce_sum_(self.buffer)
        return self.buffer

    def all_reduce_async(self, after=None):
        """Starts the SUM all-reduce and returns a handle whose ``wait()`` makes the CURRENT stream wait for
        it (no host synchronisation).  ``after``: an event (``record_event()``) the buffer's contents depend
        on; the collective is issued from a side stream that first waits for it, so nothing enqueued on the
        launch stream after the event delays the exchange and the exchange cannot start before the event --
        the way a bucketed data-parallel trainer hands a finished gradient bucket to its communication
        stream.  None for a single process."""
        if not (dist.is_initialized() and dist.get_world_size() > 1):
            return None
        if after is None or not self.buffer.is_cuda:
            return dist.all_reduce(self.buffer, op=dist.ReduceOp.SUM, async_op=True)
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=self.buffer.device)
        self._comm_stream.wait_event(after)
        with torch.cuda.stream(self._comm_stream):
            work = dist.all_reduce(self.buffer, op=dist.ReduceOp.SUM, async_op=True)
        return _StreamWork(work, self._comm_stream)


class _StreamWork:
    """``wait()``: the current stream waits for the collective (and for the side stream it was issued from)."""

    def __init__(self, work, stream):
        self.work, self.stream = work, stream

    def wait(self):
        with torch.cuda.stream(self.stream):
            self.work.wait()                     # the side stream waits for the collective's own stream
        torch.cuda.current_stream().wait_stream(self.stream)


def record_event():
    """An event behind everything enqueued so far on the current stream (None without a GPU)."""
    if not torch.cuda.is_available():
        return None
    ev = torch.cuda.Event()
    ev.record()
    return ev
