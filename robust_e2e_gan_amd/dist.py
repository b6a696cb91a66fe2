"""C1: data-parallel gradient exchange -- one process per GPU, RCCL all-reduce over xGMI.

The reference has no collective (single device, joint_train.py:55).  Utterance minibatches shard
across ranks; every rank holds full replicas; after (and, for the ASR net, during) the G-backward
the flat fp32 gradient buffer of each network (optim.FlatOptimizer.grad) is averaged with ONE
all-reduce -- 116.6 MB (ASR) + 11.3 MB (enhancer) + 11.1 MB (D) at the config-4 architecture.
xGMI is point-to-point (7 links x ~153 GB/s), so a few large buffers beat many small buckets; the
ASR all-reduce is launched from an autograd hook the moment d(enhance_feat) exists (all ASR
gradients are complete by then) and overlaps with the enhancer BLSTM backward."""
import os

import torch
import torch.distributed as dist


def init_from_env():
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl':
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def world_size():
    return dist.get_world_size() if dist.is_initialized() else 1


def shard_indices(n, rank, world):
    """Rank r takes utterances r::N of the length-sorted batch (keeps Tmax balanced)."""
    return list(range(rank, n, world))


# Communication timeline (bench.py --gpus N): with COMM_TIMING on, every gradient all-reduce leaves a record
# (bytes, event at issue on the issuing stream, event after the collective on RCCL's stream, seen from a stream that waits for it).
COMM_TIMING = False
COMM_EVENTS = []


class _TimedWork:
    """A collective's work handle plus the event that marks its completion on the device."""

    def __init__(self, work, rec):
        self.work, self.rec = work, rec

    def wait(self):
        self.work.wait()                                 # the current stream now runs behind the collective ...
        if self.rec.get('done') is None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()                                  # ... so this event is its completion as the step sees it
            self.rec['done'] = ev
            self.rec['host_wait_called'] = True


def comm_report(t0_event):
    """[(bytes, issue_ms, done_ms)] of the recorded all-reduces relative to ``t0_event``; clears the list.  Call after a device sync."""
    out = []
    for r in COMM_EVENTS:
        if r.get('done') is None:
            continue
        out.append((r['bytes'], round(t0_event.elapsed_time(r['issue']), 3), round(t0_event.elapsed_time(r['done']), 3)))
    del COMM_EVENTS[:]
    return out


def allreduce_mean_(flat, async_op=False):
    """In-place mean over ranks of a flat gradient buffer.  No-op at world_size 1 (no RCCL needed)."""
    if world_size() == 1:
        return None
    if dist.get_backend() == 'nccl':
        if COMM_TIMING and async_op:
            rec = {'bytes': flat.numel() * flat.element_size(), 'issue': torch.cuda.Event(enable_timing=True), 'done': None}
            rec['issue'].record()
            COMM_EVENTS.append(rec)
            return _TimedWork(dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=True), rec)
        return dist.all_reduce(flat, op=dist.ReduceOp.AVG, async_op=async_op)
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=False)     # gloo (CPU tests): SUM then scale
    flat.div_(world_size())
    return None


def allreduce_sum_(t):
    """In-place SUM over ranks of a small device tensor (synchronised BatchNorm statistics): enqueued behind the current stream, the
    current stream continues behind it.  No-op at world_size 1."""
    if world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def allreduce_max_(t):
    """In-place MAX over ranks of a small device tensor.  No-op at world_size 1 or without a process group."""
    if world_size() > 1 and dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return t


def any_rank(flag_value):
    """max over ranks of a small non-negative host integer (an error count / flag).  Collective: EVERY rank must call it at the
    same point.  Used so that a failure seen by one replica raises on all of them -- a lone raise would leave the others blocked
    in their next collective."""
    if world_size() == 1:
        return int(flag_value)
    dev = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend() == 'nccl' else torch.device('cpu')
    t = torch.tensor([float(flag_value)], device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return int(t.item())


def rank():
    return dist.get_rank() if dist.is_initialized() else 0


def broadcast_(t, src=0):
    """In-place broadcast of a tensor from ``src`` (no-op at world_size 1).  A CPU tensor under the nccl backend is
    staged through the current device."""
    if world_size() == 1:
        return t
    if dist.get_backend() == 'nccl' and not t.is_cuda:
        d = t.cuda()
        dist.broadcast(d, src)
        t.copy_(d.cpu())
        return t
    dist.broadcast(t, src)
    return t


def average_buffers_(module):
    """Mean over ranks of a module's floating-point buffers (the discriminator's BatchNorm running statistics: every
    replica normalises with its own batch statistics -- standard DDP semantics -- so the running estimates drift apart;
    averaging them before validation / checkpoints keeps replicas and saved models identical)."""
    if world_size() == 1:
        return
    bufs = [b for b in module.buffers() if b.dtype.is_floating_point]
    if not bufs:
        return
    flat = torch.cat([b.reshape(-1).float() for b in bufs])
    allreduce_mean_(flat)
    off = 0
    for b in bufs:
        n = b.numel()
        b.copy_(flat[off:off + n].view_as(b))
        off += n


class GradSync:
    """Overlapped gradient averaging for the joint step.

        sync = GradSync()
        sync.arm(enhance_feat, asr_opt)     # before loss.backward(): hook fires when ASR grads are complete
        loss.backward()
        sync.finish([enh_opt])              # remaining (small) buffers, then wait for everything
    """

    def __init__(self):
        self.pending = []

    def arm(self, boundary_tensor, early_opt, issue_stream=None, also_wait=()):
        """``issue_stream``: the stream that carries the weight-gradient kernels of ``early_opt``'s network when the step
        is multi-stream (ops.WGRAD_STREAM).  RCCL orders its own stream only behind the stream the collective is issued
        from, so the hook issues the all-reduce FROM that stream, after making it wait for the stream the backward runs
        on and for every stream in ``also_wait`` -- an in-place all-reduce issued from the backward's stream would race
        with gradient kernels still queued on the others."""
        self._early = None
        if world_size() == 1 or not boundary_tensor.requires_grad:
            self._early = None
            return
        self._early = early_opt

        def hook(grad):
            if issue_stream is not None and torch.cuda.is_available():
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream())
                issue_stream.wait_event(ev)
                for s in also_wait:
                    if s is not None and s is not issue_stream:
                        issue_stream.wait_stream(s)
                with torch.cuda.stream(issue_stream):
                    w = allreduce_mean_(early_opt.grad, async_op=True)
            else:
                w = allreduce_mean_(early_opt.grad, async_op=True)
            if w is not None:
                self.pending.append(w)
            self._early = None
            return grad

        boundary_tensor.register_hook(hook)

    def finish(self, opts):
        if world_size() == 1:
            return
        if getattr(self, '_early', None) is not None:        # hook never fired (boundary had no grad)
            opts = [self._early] + list(opts)
            self._early = None
        for o in opts:
            w = allreduce_mean_(o.grad, async_op=True)
            if w is not None:
                self.pending.append(w)
        for w in self.pending:
            w.wait()
        self.pending = []
