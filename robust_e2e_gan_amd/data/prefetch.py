"""Two-slot device prefetcher for the input side of the step (F1 / K1: data/mix_data_loader.py:264-302 collate, :198-237 features).

The reference collates on the host and ships five padded (B, T, F) tensors per batch with blocking ``.cuda()`` calls inside
the step.  Here the RAGGED samples of batch k+1 (or their raw Kaldi record bytes) are staged in pinned memory, copied with ONE
H2D transfer per feature stream on a dedicated copy stream and padded / decoded there by re2e_pack_pad / re2e_kaldi_decode_pad
while the GPU is still busy with step k; the consumer's stream only waits for the event behind the last of those kernels.
Pinned staging buffers are reused (two slots), so the steady state allocates nothing on the host either."""
import collections

import numpy as np
import torch

from ..lib import call


class PinnedPool(object):
    """Reusable pinned host buffers, ``slots`` per tag.  The HOST writes into a slot's buffers while it prepares a batch, so before a
    slot comes round again the H2D copies that read it must have completed ON THE GPU: ``next_slot`` blocks on the event recorded
    behind the slot's previous use (``mark_used``).  The copy stream runs ahead of the compute streams, so in steady state that
    event is long past."""

    def __init__(self, slots=3):
        self.slots, self.turn, self.bufs, self.events = slots, 0, {}, {}

    def next_slot(self):
        self.turn = (self.turn + 1) % self.slots
        ev = self.events.pop(self.turn, None)
        if ev is not None:
            ev.synchronize()

    def mark_used(self, event):
        self.events[self.turn] = event

    def get(self, tag, nbytes):
        key = (tag, self.turn)
        b = self.bufs.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(int(nbytes), 16), dtype=torch.uint8)
            if torch.cuda.is_available():
                b = b.pin_memory()
            self.bufs[key] = b
        return b[:nbytes]


class Staged(object):
    """One batch as a loader worker hands it over: the ragged rows of every selected feature stream in ONE pinned buffer each
    (sorted by length, descending), the (offset | length) table, and the host-side members of the reference's 10-tuple."""
    __slots__ = ('rows', 'meta', 'lens', 'F', 'utt_ids', 'spk_ids', 'targets', 'target_sizes')


def stage_batch(batch, pool=None, streams=(2, 4, 5)):
    """HOST half of the device collate (what the reference does in its DataLoader workers, mix_data_loader.py:264-302): sort the
    samples by length and write the ragged rows of the selected feature streams into pinned staging buffers (``pool``: reusable
    slots; None: fresh pinned tensors).  No device work."""
    batch = sorted(batch, key=lambda sample: sample[2].size(0), reverse=True)
    st = Staged()
    st.lens = [int(s[2].size(0)) for s in batch]
    B, st.F = len(batch), int(batch[0][2].size(1))
    rows = int(sum(st.lens))
    off = np.concatenate([[0], np.cumsum(st.lens)[:-1]]).astype(np.int32)
    pin = torch.cuda.is_available()

    def buf(tag, nbytes):
        if pool is not None:
            return pool.get(tag, nbytes)
        t = torch.empty(max(int(nbytes), 16), dtype=torch.uint8)
        return (t.pin_memory() if pin else t)[:nbytes]
    st.meta = buf('meta', 8 * B).view(torch.int32)
    st.meta[:B] = torch.from_numpy(off)
    st.meta[B:2 * B] = torch.tensor(st.lens, dtype=torch.int32)
    st.rows = {}
    for k in streams:
        host = buf('s%d' % k, rows * st.F * 4).view(torch.float32).view(rows, st.F)
        torch.cat([s[k] for s in batch], 0, out=host)
        st.rows[k] = host
    st.utt_ids, st.spk_ids = [s[0] for s in batch], [s[1] for s in batch]
    st.targets = torch.LongTensor([int(t) for s in batch for t in s[7]])
    st.target_sizes = torch.IntTensor([len(s[7]) for s in batch])
    return st


def upload_staged(st, device):
    """DEVICE half: one non-blocking H2D copy per feature stream and re2e_pack_pad (zero padding to (B, Tmax, F)) on the CURRENT
    stream.  Returns the reference's 10-tuple with the staged streams on ``device`` (the others None)."""
    B, T = len(st.lens), st.lens[0]
    cuda = torch.device(device).type == 'cuda'
    meta_d = st.meta.to(device, non_blocking=True)
    outs = [None] * 5
    for k, host in st.rows.items():
        flat = host.to(device, non_blocking=True)
        out = torch.empty(B, T, st.F, dtype=torch.float32, device=device)
        call('re2e_pack_pad', flat.data_ptr(), meta_d.data_ptr(), meta_d.data_ptr() + 4 * B, B, T, st.F, out.data_ptr())
        if cuda:
            flat.record_stream(torch.cuda.current_stream())
        outs[k - 2] = out
    if cuda:
        meta_d.record_stream(torch.cuda.current_stream())
    return (st.utt_ids, st.spk_ids, outs[0], outs[1], outs[2], outs[3], outs[4], st.targets, torch.IntTensor(st.lens), st.target_sizes)


def collate_device_pinned(batch, device, pool, streams=(2, 4, 5)):
    """``mix_data_loader.collate_device`` with pooled pinned staging (``stage_batch`` + ``upload_staged``); a batch that is already
    a ``Staged`` object (prepared by loader workers) is uploaded as it is.  Same 10-tuple, same values as ``_collate_fn`` for the
    selected streams."""
    if isinstance(batch, Staged):
        return upload_staged(batch, device)
    return upload_staged(stage_batch(batch, pool, streams), device)


class DevicePrefetcher(object):
    """Iterate ``batches`` (an iterable of un-collated sample lists, e.g. ``DataLoader(..., collate_fn=lambda b: b)``) and yield
    device-resident 10-tuples whose staging, transfer and padding ran up to ``depth`` batches ahead.

    ``collate(batch, device, pool)`` must enqueue all its device work on the CURRENT stream (``collate_device_pinned``, or
    ``lambda b, d, p: collate_kaldi_device(b, d, cmvn)`` for raw Kaldi records).  It runs in a WORKER THREAD under the copy
    stream: writing 3 x 22 MB of ragged rows into pinned (host-coherent) memory takes the host 30-50 ms per config-4 batch
    (measured, ~2 GB/s), which on the thread that also enqueues the step's ~1 400 launches would make the step host-bound -- the
    reference does this work in DataLoader worker processes for the same reason.  The consumer's stream waits for the event
    recorded behind the batch's last device kernel when the batch is handed out; its tensors are marked as used by that stream.
    ``stream``: the copy stream (default: a new one; a process has four hardware queues by default, so callers that already run
    several streams pass one of theirs that is idle at the start of a step, or set GPU_MAX_HW_QUEUES)."""

    def __init__(self, batches, device, collate=collate_device_pinned, depth=2, stream=None, threaded=True):
        self.batches, self.device, self.collate, self.depth = batches, torch.device(device), collate, max(1, int(depth))
        self.pool = PinnedPool(self.depth + 2)
        self.stream = stream
        self.threaded = bool(threaded) and self.device.type == 'cuda'

    def __len__(self):
        return len(self.batches)

    def _issue(self, batch):
        cuda = self.device.type == 'cuda'
        if not cuda:
            return self.collate(batch, self.device, self.pool), None
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=self.device)
        self.pool.next_slot()
        with torch.cuda.stream(self.stream):
            out = self.collate(batch, self.device, self.pool)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.pool.mark_used(ev)
        return out, ev

    def _hand_out(self, out, ev):
        if ev is not None:
            cur = torch.cuda.current_stream()
            cur.wait_event(ev)
            for t in out:
                if isinstance(t, torch.Tensor) and t.is_cuda:
                    t.record_stream(cur)
        return out

    def __iter__(self):
        if not self.threaded or all(isinstance(b, Staged) for b in (self.batches if isinstance(self.batches, (list, tuple)) else [None])):
            it = iter(self.batches)
            q = collections.deque()
            for b in it:
                q.append(self._issue(b))
                if len(q) >= self.depth:
                    break
            while q:
                out, ev = q.popleft()
                nxt = next(it, None)
                if nxt is not None:
                    q.append(self._issue(nxt))
                yield self._hand_out(out, ev)
            return
        import queue
        import threading
        q = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        dev = self.device

        def work():
            try:
                torch.cuda.set_device(dev)
                for b in self.batches:
                    if stop.is_set():
                        break
                    q.put(self._issue(b))
                q.put(None)
            except BaseException as e:          # hand the failure to the consumer instead of dying silently
                q.put(e)
        th = threading.Thread(target=work, daemon=True)
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                yield self._hand_out(*item)
        finally:
            stop.set()
            while th.is_alive():                 # unblock a producer waiting on a full queue
                try:
                    q.get_nowait()
                except queue.Empty:
                    th.join(0.01)
