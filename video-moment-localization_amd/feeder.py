"""Batch feeder for the data-parallel train step (reference dataset.py:129-187 + main.py:118-133).

The reference builds eleven mask / target tensors per sample on the host (Python loops over CPU tensors), collates them and
ships thirteen tensors per batch to the device with blocking ``.to(device)`` calls.  Here a rank stages only what cannot be
derived -- the sampled clip features, the query word vectors and five scalars per sample (sampled frames, query length, ground
truth start / end, duration) -- in pinned host buffers, copies them on a dedicated HIP stream while the previous step computes,
and one kernel (csrc/labels.hip) writes every mask and target on the device.  Two slots alternate; a slot is refilled only
after the step that consumed it has been queued, which an event orders without host synchronisation.
"""
import torch

from . import _lib
from ._lib import call, ptr

_BATCH_KEYS = ("video_features", "video_mask", "query_features", "query_mask", "length_mask", "moment_mask",
               "sm", "ym", "ss", "ys", "se", "ye", "ya")


def build_targets_hip(times, duration, nfeats, qlen, T, L, Nq, stream=None):
    """All masks and targets of a batch in one launch (device tensors in, dict of device tensors out; dtypes and shapes as
    main.py reads them: video_mask / query_mask uint8 (B, T, 1) / (B, Nq, 1), length_mask / moment_mask / y* bool)."""
    if not times.is_cuda:
        raise _lib.SminHipError("build_targets_hip runs on a HIP device only (got a CPU tensor); there is no CPU fallback")
    dev, B = times.device, times.shape[0]
    u8 = lambda *s: torch.empty(s, dtype=torch.uint8, device=dev)
    f32 = lambda *s: torch.empty(s, dtype=torch.float32, device=dev)
    out = dict(video_mask=u8(B, T, 1), query_mask=u8(B, Nq, 1) if qlen is not None else None, length_mask=u8(B, L), moment_mask=u8(B, L, L),
               sm=f32(B, L, L), ym=u8(B, L, L), ss=f32(B, L), ys=u8(B, L), se=f32(B, L), ye=u8(B, L), ya=u8(B, L))
    # annotation times that still carry their double precision: 2 sigma^2 in double, rounded once (dataset.py:116-119 does the same
    # arithmetic in Python floats); fp32 times: the kernel forms it in double from what it is given
    den = None
    if times.dtype == torch.float64:
        den = (2.0 * ((times[:, 1] - times[:, 0]) / 5.0) ** 2).float().contiguous()
    times, duration = times.float().contiguous(), duration.float().contiguous()
    nfeats = nfeats.to(torch.int32).contiguous()
    qlen32 = qlen.to(torch.int32).contiguous() if qlen is not None else None
    with torch.cuda.device(dev):
        s = _lib.stream() if stream is None else stream
        call("smin_build_targets", s, ptr(times), ptr(duration), ptr(nfeats), ptr(qlen32), B, T, L, Nq, ptr(out["video_mask"]), ptr(out["query_mask"]),
             ptr(out["length_mask"]), ptr(out["moment_mask"]), ptr(out["sm"]), ptr(out["ym"]), ptr(out["ss"]), ptr(out["ys"]), ptr(out["se"]),
             ptr(out["ye"]), ptr(out["ya"]), ptr(den))
    for k in ("length_mask", "moment_mask", "ym", "ys", "ye", "ya"):
        out[k] = out[k].view(torch.bool)                      # same bytes; the reference's tensors are BoolTensors
    if qlen is None:
        del out["query_mask"]
    return out


class BatchFeeder:
    """Double-buffered host -> device feeder.  ``feed(sample_batches)`` takes an iterable of host batches
    ``dict(video_features (B,T,Din) float32, query_features (B,Nq,300) float32, nfeats (B,), qlen (B,), times (B,2), duration (B,))``
    (numpy arrays or CPU tensors) and yields device batches with the thirteen entries main.py's loop reads, one batch ahead of
    the consumer."""

    def __init__(self, T, L, Nq, device, depth=3):
        self.T, self.L, self.Nq, self.device = T, L, Nq, torch.device(device)
        if self.device.type != "cuda":
            raise _lib.SminHipError("BatchFeeder feeds a HIP device; there is no CPU path")
        self.copy_stream = torch.cuda.Stream(self.device)
        self.slots = [dict(host={}, ready=torch.cuda.Event(), consumed=None) for _ in range(depth)]

    def _pinned(self, slot, key, src):
        if src.is_pinned():                                                  # a loader with pin_memory=True: no staging copy
            return src
        buf = slot["host"].get(key)
        if buf is None or buf.shape != src.shape or buf.dtype != src.dtype:
            buf = slot["host"][key] = torch.empty(src.shape, dtype=src.dtype, pin_memory=True)
        buf.copy_(src)
        return buf

    def _stage(self, slot, hb):
        """Host side of one batch: into the slot's pinned buffers, then H2D + target construction on the copy stream."""
        if "batch" in slot:
            slot["ready"].synchronize()                                      # the slot's previous H2D copies have left its pinned buffers
        t = {k: torch.as_tensor(v) for k, v in hb.items()}
        host = {k: self._pinned(slot, k, t[k].float() if k in ("video_features", "query_features", "times", "duration") else t[k].to(torch.int32))
                for k in ("video_features", "query_features", "nfeats", "qlen", "times", "duration")}
        with torch.cuda.device(self.device), torch.cuda.stream(self.copy_stream):
            if slot["consumed"] is not None:
                self.copy_stream.wait_event(slot["consumed"])              # the step that read this slot's tensors is queued before we overwrite
            d = {k: v.to(self.device, non_blocking=True) for k, v in host.items()}
            B = d["video_features"].shape[0]
            # (rows past a sample's sampled frames / words arrive as zeros, as the reference's loader makes them: dataset.py:72-73, 172)
            tg = build_targets_hip(d["times"], d["duration"], d["nfeats"], d["qlen"], self.T, self.L, self.Nq)
            batch = dict(video_features=d["video_features"], query_features=d["query_features"], **tg)
            slot["ready"].record(self.copy_stream)
        slot["batch"] = {k: batch[k] for k in _BATCH_KEYS}
        return B

    def feed(self, host_batches):
        """Generator of device batches.  A worker thread does the host side of every batch (pinned staging copy, enqueueing the
        H2D copies and the target kernel on the copy stream) up to ``depth`` batches ahead; the consumer's thread only makes its
        stream wait for the slot's event, so no step waits for a host memcpy."""
        import queue
        import threading
        free, ready = queue.Queue(), queue.Queue()
        for slot in self.slots:
            free.put(slot)
        stop = threading.Event()

        def worker():
            try:
                for hb in host_batches:
                    slot = free.get()
                    if stop.is_set():
                        return
                    self._stage(slot, hb)
                    ready.put(slot)
                ready.put(None)
            except BaseException as e:                                       # surfaces in the consumer
                ready.put(e)

        th = threading.Thread(target=worker, name="smin-batch-feeder", daemon=True)
        th.start()
        try:
            while True:
                slot = ready.get()
                if slot is None:
                    return
                if isinstance(slot, BaseException):
                    raise slot
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(slot["ready"])                                # device-side wait only
                for v in slot["batch"].values():
                    v.record_stream(cur)
                yield slot["batch"]
                slot["consumed"] = torch.cuda.Event()
                slot["consumed"].record(torch.cuda.current_stream(self.device))
                free.put(slot)
        finally:
            stop.set()
            free.put(self.slots[0])                                          # unblock a worker waiting for a slot
