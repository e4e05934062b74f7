"""Online repetition counting for many concurrent frame streams on one engine.

Counterpart of the reference's streaming consumers of the video model -- ``count_by_video_model``
(workoutdetector/utils/inference_count.py:285-339: a ``deque(maxlen=8)``, one inference per 8 queued
frames, ``input_queue.clear()``) and the WebSocket loop of ``app/inference.py:87-111`` / ``app/server.py:85-100``
(one client = one stream, 8 frames per window) -- re-shaped for the GPU: every stream only queues frames;
``StreamBatcher.step()`` collects the windows that are complete across ALL streams and pushes them through the
engine as ONE batch (fused HIP transform + ``tsm_forward``), then feeds each stream's incremental counter.
The reference runs one blocking batch-1 ``session.run`` per client window.

No transport here (the WebSocket/HTTP front is outside the hot path): a server
calls ``push`` from its receive loop and ``step`` on a timer or whenever ``ready()`` is large enough.
"""
from __future__ import annotations

from collections import deque
from dataclasses import dataclass, field
from typing import Callable, Deque, Dict, Hashable, List, Optional, Tuple

import numpy as np
import torch

from .counting import RepCounter, scores_to_preds
from .inference_count import NUM_SEGMENTS, _engine_device
from .transform import TestTransform, build_test_transform


@dataclass
class StreamState:
    counter: RepCounter
    frames: List[np.ndarray] = field(default_factory=list)        # frames of the window being filled (host path)
    windows: Deque[object] = field(default_factory=deque)         # complete [8,H,W,3] uint8 windows not yet run
    states: List[int] = field(default_factory=list)               # one state per processed window
    frames_seen: int = 0
    cur: Optional[torch.Tensor] = None                            # HIP path: page-locked window being filled
    n_cur: int = 0


class StreamBatcher:
    """``push(stream_id, frame)`` HWC uint8 RGB frames; ``step()`` runs every complete 8-frame window of every
    stream in batches of at most ``max_batch`` and returns ``{stream_id: [(window_index, state, count), ...]}``.

    Windows are non-overlapping with stride 8 like the reference's streaming loop; ``softmax``/``threshold``
    follow utils/eval.py:153-164; frame sizes may differ between streams (windows are transformed per source
    resolution, then batched at 224x224)."""

    def __init__(self, model, threshold: float = 0.5, softmax: bool = True, step: int = 8, max_batch: int = 32,
                 transform: Optional[TestTransform] = None,
                 on_window: Optional[Callable[[Hashable, int, int, int], None]] = None,
                 max_pinned_bytes: int = 1 << 30, max_free_per_shape: int = 64):
        self.model = model
        self.threshold, self.softmax, self.step_frames = threshold, softmax, step
        self.max_batch = max_batch
        self.transform = transform or build_test_transform(False)
        self.on_window = on_window
        self.streams: Dict[Hashable, StreamState] = {}
        # HIP path: every frame is copied into a page-locked [8,H,W,3] window buffer when it ARRIVES (push), so that a
        # complete window is one DMA away from the GPU when step() runs -- the 1.8-MB host gather of a 360x206 window
        # (~0.15 ms) leaves the window's critical path.  Buffers are recycled once their H2D copy has completed.
        self._dev = _engine_device(model) if hasattr(model, 'packed_layout') else None
        # Page-locked memory is bounded: at most ``max_pinned_bytes`` in window buffers (a producer that runs ahead of
        # step(), or many resolutions, falls back to pageable windows beyond it -- slower upload, no hipHostMalloc on
        # the frame-arrival path, no unbounded pinning), at most ``max_free_per_shape`` idle buffers per frame size.
        self._free: Dict[Tuple[int, ...], List[torch.Tensor]] = {}
        self._inflight: List[Tuple[object, torch.Tensor]] = []
        self.max_pinned_bytes, self.max_free_per_shape = int(max_pinned_bytes), int(max_free_per_shape)
        self.pinned_bytes = 0

    # ---- ingest -------------------------------------------------------------------------------------------
    def open(self, stream_id: Hashable) -> StreamState:
        if stream_id not in self.streams:
            self.streams[stream_id] = StreamState(RepCounter(self.step_frames))
        return self.streams[stream_id]

    def push(self, stream_id: Hashable, frame) -> None:
        st = self.open(stream_id)
        arr = np.asarray(frame)
        if arr.dtype != np.uint8 or arr.ndim != 3 or arr.shape[2] != 3:
            raise ValueError(f'frame must be uint8 [H,W,3], got {arr.dtype} {arr.shape}')
        st.frames_seen += 1
        if self._dev is not None:
            if st.cur is None:
                st.cur, st.n_cur = self._take_window(tuple(arr.shape)), 0
            elif tuple(st.cur.shape[1:]) != tuple(arr.shape):
                raise ValueError('frame size changed inside a window')
            st.cur[st.n_cur].copy_(torch.from_numpy(np.ascontiguousarray(arr)))
            st.n_cur += 1
            if st.n_cur == NUM_SEGMENTS:
                st.windows.append(st.cur)
                st.cur = None                   # input_queue.clear() of the reference
            return
        if st.frames and st.frames[0].shape != arr.shape:
            raise ValueError('frame size changed inside a window')
        st.frames.append(arr)
        if len(st.frames) == NUM_SEGMENTS:
            st.windows.append(np.stack(st.frames))
            st.frames = []                      # input_queue.clear() of the reference

    def _recycle(self, wait: bool = False) -> None:
        """Window buffers whose upload has finished go back to the free lists (capped per shape; the surplus and every
        pageable fallback buffer is simply dropped)."""
        still = []
        for ev, buf in self._inflight:
            if wait:
                ev.synchronize()
            if ev.query():
                self._give_back(buf)
            else:
                still.append((ev, buf))
        self._inflight = still

    def _give_back(self, buf: torch.Tensor) -> None:
        """An idle window buffer returns to its shape's free list while that list is below ``max_free_per_shape``; beyond
        the cap (and for pageable fallback buffers) it is dropped and its page-locked bytes leave the budget."""
        if not buf.is_pinned():
            return
        free = self._free.setdefault(tuple(buf.shape[1:]), [])
        if len(free) < self.max_free_per_shape:
            free.append(buf)
        else:
            self.pinned_bytes -= buf.numel()

    def _take_window(self, frame_shape: Tuple[int, ...]) -> torch.Tensor:
        """A uint8 [8,H,W,3] window buffer: a recycled page-locked one if one of this size is free, a newly pinned one
        while the pinned budget lasts, else pageable memory."""
        self._recycle()
        free = self._free.get(frame_shape)
        if free:
            return free.pop()
        nbytes = NUM_SEGMENTS * int(np.prod(frame_shape))
        if self.pinned_bytes + nbytes > self.max_pinned_bytes:
            self._trim_free(keep=frame_shape)           # idle buffers of other resolutions give their budget back first
        if self.pinned_bytes + nbytes > self.max_pinned_bytes:
            return torch.empty((NUM_SEGMENTS,) + frame_shape, dtype=torch.uint8)
        self.pinned_bytes += nbytes
        return torch.empty((NUM_SEGMENTS,) + frame_shape, dtype=torch.uint8, pin_memory=True)

    def _trim_free(self, keep: Optional[Tuple[int, ...]] = None) -> None:
        """Release the idle page-locked buffers of every frame size no open stream is filling (except ``keep``)."""
        live = {tuple(s.cur.shape[1:]) for s in self.streams.values() if s.cur is not None}
        for shape in list(self._free):
            if shape != keep and shape not in live:
                self.pinned_bytes -= sum(b.numel() for b in self._free.pop(shape))

    def ready(self) -> int:
        return sum(len(s.windows) for s in self.streams.values())

    def close(self, stream_id: Hashable) -> Tuple[int, List[int]]:
        """Drop a stream (an incomplete last window is discarded, like the reference) -> (count, reps)."""
        st = self.streams.pop(stream_id)
        for buf in ([st.cur] if st.cur is not None else []) + [w for w in st.windows if isinstance(w, torch.Tensor)]:
            self._give_back(buf)                             # half-filled / never-run windows: the same capped path as _recycle
        self._trim_free()
        return st.counter.count, list(st.counter.reps)

    def result(self, stream_id: Hashable) -> Tuple[int, List[int]]:
        st = self.streams[stream_id]
        return st.counter.count, list(st.counter.reps)

    # ---- compute ------------------------------------------------------------------------------------------
    def _logits(self, windows: List[object]):
        """[n,8,H,W,3] uint8 windows (possibly of different sizes) -> raw logits [n, num_class]: a CUDA tensor that is
        still being computed on the HIP path (the caller syncs once per step), an ndarray on the duck-typed CPU path."""
        dev = self._dev
        if dev is not None:
            from .engine import preprocess_frames
            layout = self.model.packed_layout
            by_shape: Dict[Tuple[int, ...], List[int]] = {}
            for i, w in enumerate(windows):     # one transform launch per source resolution
                by_shape.setdefault(tuple(w.shape), []).append(i)
            clips = None
            for shape, idx in by_shape.items():
                # the windows are already page-locked (filled at push time): one DMA each, straight into place
                if len(idx) == 1:
                    fr = windows[idx[0]].to(dev, non_blocking=True)
                else:
                    fr = torch.empty((len(idx) * shape[0],) + tuple(shape[1:]), dtype=torch.uint8, device=dev)
                    for j, i in enumerate(idx):
                        fr[j * shape[0]:(j + 1) * shape[0]].copy_(windows[i], non_blocking=True)
                done = torch.cuda.Event()
                done.record()
                self._inflight += [(done, windows[i]) for i in idx]
                pk = preprocess_frames(fr, resize=self.transform.size, crop=self.transform.crop,
                                       scale_255=self.transform.scale_255, layout=layout)
                pk = pk.view((len(idx), NUM_SEGMENTS) + tuple(pk.shape[1:]))
                if len(by_shape) == 1:
                    clips = pk
                else:
                    if clips is None:
                        clips = torch.empty((len(windows),) + tuple(pk.shape[1:]), dtype=pk.dtype, device=dev)
                    clips[torch.tensor(idx, device=dev)] = pk
            return self.model.forward_device(clips.contiguous(), layout=layout)
        xs = [self.transform(torch.from_numpy(w).permute(0, 3, 1, 2).float()) for w in windows]
        name = self.model.get_inputs()[0].name
        return np.asarray(self.model.run(None, {name: torch.stack(xs).numpy()})[0])

    def step(self) -> Dict[Hashable, List[Tuple[int, int, int]]]:
        """Run all complete windows (oldest first, round-robin over streams) and update the counters."""
        out: Dict[Hashable, List[Tuple[int, int, int]]] = {}
        order: List[Hashable] = []
        pending = []
        while self.ready():
            batch: List[Tuple[Hashable, np.ndarray]] = []
            progressed = True
            while len(batch) < self.max_batch and progressed:   # round-robin keeps per-stream order and fairness
                progressed = False
                for sid, st in self.streams.items():
                    if st.windows and len(batch) < self.max_batch:
                        batch.append((sid, st.windows.popleft()))
                        progressed = True
            # enqueue only: the host gathers / pins batch i+1 while the GPU still computes batch i
            pending.append(self._logits([w for _, w in batch]))
            order += [sid for sid, _ in batch]
        if pending:
            if isinstance(pending[0], torch.Tensor):
                # HIP path: softmax / arg-max / threshold on the GPU too (tsm_scores_to_states): one int32 per window
                # crosses PCIe, in the step's only host sync
                from .engine import scores_to_states
                states = scores_to_states(torch.cat(pending), threshold=self.threshold, softmax=self.softmax).cpu().tolist()
                self._recycle()     # that .cpu() was the step's sync: every upload queued before it has completed
            else:
                states = scores_to_preds(np.concatenate(pending).tolist(), threshold=self.threshold, softmax=self.softmax)
            for sid, state in zip(order, states):
                st = self.streams[sid]
                widx = len(st.states)
                st.states.append(state)
                count = st.counter.push(state)
                out.setdefault(sid, []).append((widx, state, count))
                if self.on_window is not None:
                    self.on_window(sid, widx, state, count)
        return out
