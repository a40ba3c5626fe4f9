"""Plumbing shared by the host mirrors (field_utils.py, xie.py): staging of tensors on the compute device, per-(thread, device, stream)
scratch, the launch wrapper of the K1 / K2 entry points, the deferred Inf / NaN warnings of field_grad (field_utils.py:110-113 of the
reference) and the thread-local traces of the greedy drivers.  Nothing here computes; there is no CPU fallback (the library or a HIP
device missing -> _lib raises).  Split out of field_utils.py in round 5; every name is re-exported there."""
from typing import Optional

import atexit
import ctypes
import threading

import torch

from . import _lib


# ---------------------------------------------------------------------------------------------------
# plumbing
# ---------------------------------------------------------------------------------------------------
def _compute_device() -> torch.device:
    _lib.require_device()
    return torch.device("cuda", torch.cuda.current_device())


def _stage(t: torch.Tensor, dev: torch.device, dtype: torch.dtype) -> torch.Tensor:
    """Tensor on the compute device with unit inner stride (rows may be strided)."""
    if t.device != dev or t.dtype != dtype:
        t = t.to(device=dev, dtype=dtype)
    if t.dim() != 2:
        raise ValueError(f"expected a 2-d point tensor, got shape {tuple(t.shape)}")
    if t.shape[0] > 1 and (t.stride(1) != 1 or t.stride(0) < t.shape[1]):
        t = t.contiguous()
    elif t.shape[0] <= 1 and t.shape[1] > 1 and t.stride(1) != 1:
        t = t.contiguous()
    return t


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL = _NullCtx()


def _on_device(dev):
    """torch.cuda.device(dev) only when dev is not already current (the context manager costs ~10 us)."""
    return _NULL if dev.index is None or dev.index == torch.cuda.current_device() else torch.cuda.device(dev)


_tls = threading.local()


def _workspace(nbytes: int, dev: torch.device, stream_handle=None) -> torch.Tensor:
    """Scratch for one launch sequence, cached per (thread, device, stream): work on one stream is
    ordered, so the next call on that stream may overwrite it; other streams / threads get their own."""
    cache = getattr(_tls, "ws", None)
    if cache is None:
        cache = _tls.ws = {}
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream if stream_handle is None else stream_handle)
    buf = cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = cache[key] = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
    return buf


def _ld(t: torch.Tensor) -> int:
    return t.stride(0) if t.shape[0] > 1 else max(t.shape[1], 1)


def _work_dtype(*ts) -> torch.dtype:
    return torch.float64 if any(t.dtype == torch.float64 for t in ts) else torch.float32


def _idx(t: Optional[torch.Tensor], dev) -> Optional[torch.Tensor]:
    if t is None:
        return None
    return t.to(device=dev, dtype=torch.int64).contiguous()


_WS_BYTES = {}


def _pairs_into(kind: str, src, src_idx, tgt, tgt_idx, eps, max_pts, out, out_scatter=False, accumulate=False,
                nonfinite=(None, None), stream_handle=None):
    """Launch K1/K2 on staged device tensors.  src/tgt/out live on the compute device.  nonfinite = (device
    pointer, pinned host pointer) of the three-int warning slot of this call, as ctypes pointers or None;
    stream_handle = the device's current stream when the caller has looked it up already."""
    lib = _lib.require_device()
    S = src.shape[0] if src_idx is None else src_idx.shape[0]
    T = tgt.shape[0] if tgt_idx is None else tgt_idx.shape[0]
    f64 = src.dtype == torch.float64
    nbytes = _WS_BYTES.get((kind, S, T, max_pts))
    if nbytes is None:                                  # a pure function of the sizes: asked once per shape
        if len(_WS_BYTES) > 4096:
            _WS_BYTES.clear()
        if kind == "field":
            nbytes = lib.dnp_field_grad_workspace_bytes(S, T, max_pts)
        else:
            nbytes = lib.dnp_potential_workspace_bytes(S, T, max_pts)
        _WS_BYTES[(kind, S, T, max_pts)] = nbytes
    if stream_handle is None:
        stream_handle = torch.cuda.current_stream(src.device).cuda_stream
    ws = _workspace(nbytes, src.device, stream_handle)
    with _on_device(src.device):
        stream = ctypes.c_void_p(stream_handle)
        if kind == "field":
            fn = lib.dnp_field_grad_f64 if f64 else lib.dnp_field_grad_f32
            rc = fn(_lib.ptr(src), S, _ld(src), _lib.ptr(src_idx), _lib.ptr(tgt), T, _ld(tgt), _lib.ptr(tgt_idx),
                    float(eps), int(max_pts), _lib.ptr(out), out.stride(0) if out.shape[0] > 1 else 3,
                    int(bool(out_scatter)), int(bool(accumulate)), nonfinite[0], nonfinite[1], _lib.ptr(ws),
                    ws.numel(), stream)
        else:
            fn = lib.dnp_potential_f64 if f64 else lib.dnp_potential_f32
            rc = fn(_lib.ptr(src), S, _ld(src), _lib.ptr(src_idx), _lib.ptr(tgt), T, _ld(tgt), _lib.ptr(tgt_idx),
                    int(max_pts), _lib.ptr(out), 1, _lib.ptr(ws), nbytes, stream)
    _lib.check(rc)
    return out


# ---- the reference's "warning: %d inf in field_grad" prints (field_utils.py:110-113) -----------------------------
# The kernels count the Inf / NaN leaf components they zero into a three-int slot {inf, nan, spare} of a small
# device ring, one slot per field_grad call.  Nothing is copied per call: after every _WARN_BATCH calls ONE
# asynchronous copy moves that block of slots to pinned host memory behind the kernels (and re-arms the block), an
# event says when it has landed, and the lines are printed by the next call on this thread that finds it landed -
# at the latest by flush_warnings() or interpreter exit.  A field_grad call therefore never waits for the device,
# and never launches a copy of its own, just to find out that there is nothing to warn about.
_WARN_RING = 64
_WARN_BATCH = 16


class _WarnState:
    def __init__(self, dev):
        self.dev = dev
        self.ring = torch.zeros((_WARN_RING, 3), dtype=torch.int32, device=dev)
        self.host = torch.zeros((_WARN_RING, 3), dtype=torch.int32).pin_memory()
        self.view = self.host.numpy()
        self.dev_base = self.ring.data_ptr()
        self.head = 0             # next slot to hand out
        self.copied = 0           # slots below this have their copy to the host enqueued
        self.tail = 0             # oldest slot not yet reported
        self.batches = []         # (event, lo, hi) of the copies in flight, oldest first
        self.stream = None        # stream of the calls since the last copy; False once they used more than one
        self.lock = threading.Lock()   # flush_warnings() may drain this state from another thread
        torch.cuda.current_stream(dev).synchronize()                  # the ring is initialised before its first use

    def next_slot(self, stream):
        """(device pointer, None) of the slot of one call launched on `stream` (an integer handle)."""
        with self.lock:
            full = self.head - self.tail >= _WARN_RING                # every slot is in flight: wait for the oldest
        if full:
            self.drain(block=True)
        with self.lock:
            i = self.head % _WARN_RING
            self.head += 1
            if self.stream is None:
                self.stream = stream
            elif self.stream != stream:
                self.stream = False
        return (ctypes.c_void_p(self.dev_base + 12 * i), None)

    def _enqueue_copy(self):
        """Copy slots [copied, head) to the host behind the kernels that fill them, re-arm them, remember the event."""
        lo, hi = self.copied, self.head
        if lo == hi:
            return
        with _on_device(self.dev):
            if self.stream is False:                                  # calls on several streams: order them all first
                torch.cuda.synchronize(self.dev)
            a, b = lo % _WARN_RING, (hi - 1) % _WARN_RING + 1
            for x, y in ([(a, b)] if a < b else [(a, _WARN_RING), (0, b)]):
                self.host[x:y].copy_(self.ring[x:y], non_blocking=True)
                self.ring[x:y].zero_()
            ev = torch.cuda.Event()
            ev.record()
        self.batches.append((ev, lo, hi))
        self.copied, self.stream = hi, None

    def after_call(self):
        with self.lock:
            if self.head - self.copied >= _WARN_BATCH:
                self._enqueue_copy()

    def drain(self, block=False):
        with self.lock:
            if block:
                torch.cuda.synchronize(self.dev)                      # also orders calls made on other streams
                self._enqueue_copy()
            while self.batches and (block or self.batches[0][0].query()):
                ev, lo, hi = self.batches.pop(0)
                ev.synchronize()
                for k in range(lo, hi):
                    n_inf, n_nan = int(self.view[k % _WARN_RING, 0]), int(self.view[k % _WARN_RING, 1])
                    if n_inf:
                        print("warning: %d inf in field_grad" % n_inf)
                    if n_nan:
                        print("warning: %d nan in field_grad" % n_nan)
                self.tail = hi


_warn_states = []
_warn_lock = threading.Lock()


def _warn_state(dev) -> _WarnState:
    states = getattr(_tls, "warn", None)
    if states is None:
        states = _tls.warn = {}
    st = states.get(dev.index)
    if st is None:
        st = states[dev.index] = _WarnState(dev)
        with _warn_lock:
            _warn_states.append(st)
    return st


def flush_warnings() -> None:
    """Print every pending Inf/NaN warning of field_grad calls made so far (waits for the device)."""
    with _warn_lock:
        states = list(_warn_states)
    for st in states:
        st.drain(block=True)


atexit.register(lambda: flush_warnings() if _warn_states else None)


# ---------------------------------------------------------------------------------------------------
# traces
# ---------------------------------------------------------------------------------------------------
def _set_trace(kind: str, **items) -> None:
    """Remember the visit order / flips / chosen interactions of the calling THREAD's last driver call.
    Thread-local: the reference runs these drivers concurrently from Python threads (util.py:187-196,
    :308-327).  Values may be device tensors; they are converted when somebody asks (last_trace)."""
    store = getattr(_tls, "traces", None)
    if store is None:
        store = _tls.traces = {}
    store[kind] = items


def last_trace(kind: str) -> dict:
    """Trace of this thread's last call of a greedy driver as numpy arrays.  kind: "patches"
    (strongest_field_propagation), "reps" (..._reps), "points" (..._points), "sharded"
    (parallel.sharded_patch_propagation).  Keys: order, sigma (+-1 per patch), chosen, start."""
    items = getattr(_tls, "traces", {}).get(kind)
    if items is None:
        raise KeyError(f"no {kind!r} driver has run on this thread")
    out = {}
    for k, v in items.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
            if v.ndim == 1 and v.shape[0] == 1 and k == "start":
                v = int(v[0])
        out[k] = v
    return out
