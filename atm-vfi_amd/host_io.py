"""Host boundary either side of the hot path: the reference's
``InputPadder`` (benchmark/utils.py:57-80), ``inference_2frame`` (demo_2x.py:54-87) and
``load_model_checkpoint`` (demo_2x.py:24-51), restated so that the reference's scripts
run unchanged on top of this package."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


class InputPadder:
    """Centre replicate-padding to a multiple of ``divisor`` (benchmark/utils.py:57-80)."""

    def __init__(self, dims, divisor: int = 16):
        self.ht, self.wd = dims[-2:]
        ph = (((self.ht // divisor) + 1) * divisor - self.ht) % divisor
        pw = (((self.wd // divisor) + 1) * divisor - self.wd) % divisor
        self._pad = [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2]

    def pad(self, *inputs):
        out = [F.pad(x, self._pad, mode="replicate") for x in inputs]
        return out[0] if len(out) == 1 else out

    def unpad(self, *inputs):
        out = [self._unpad(x) for x in inputs]
        return out[0] if len(out) == 1 else out

    def _unpad(self, x):
        ht, wd = x.shape[-2:]
        l, r, t, b = self._pad
        return x[..., t:ht - b, l:wd - r]


def img2tensor(img):        # benchmark/utils.py:83-86
    if img.shape[-1] > 3:
        img = img[:, :, :3]
    return torch.tensor(img).permute(2, 0, 1).unsqueeze(0) / 255.0


def _hip_ops_of(model):
    """The HIP op backend of a ``Network`` living on the GPU (None for any other model: the generic path below is used)."""
    dev = next(model.parameters()).device
    if dev.type != "cuda" or not hasattr(model, "_ops"):
        return None, dev
    ops = model._ops(dev)
    return (ops if hasattr(ops, "frame_u8_to_f32") else None), dev


def inference_2frame(img0, img1, model, isBGR: bool = True, divisor: int = 64):
    """uint8 [H,W,3] frames -> uint8 [H,W,3] interpolated frame (demo_2x.py:54-87).  With the HIP backend the colour flip, /255,
    replicate padding, un-padding and np.round(x*255) run as two device kernels on the raw uint8 frames (bit-identical)."""
    ops, dev = _hip_ops_of(model)
    if ops is not None and img0.dtype == np.uint8 and img0.ndim == 3 and img0.shape[2] == 3 and img0.shape == img1.shape:
        pipe = FramePipeline(model, img0.shape[0], img0.shape[1], isBGR=isBGR, divisor=divisor, depth=1)
        return next(pipe.run([(img0, img1)]))
    if isBGR:
        img0 = img0[:, :, ::-1].copy()
        img1 = img1[:, :, ::-1].copy()
    t0 = (torch.tensor(img0.transpose(2, 0, 1)).to(dev) / 255.).unsqueeze(0)
    t1 = (torch.tensor(img1.transpose(2, 0, 1)).to(dev) / 255.).unsqueeze(0)
    padder = InputPadder(t0.shape, divisor=divisor)
    t0, t1 = padder.pad(t0, t1)
    pred = model.forward(t0, t1)["I_t"][0]
    pred = padder.unpad(pred).detach().cpu().numpy().transpose(1, 2, 0)
    pred = np.round(pred * 255).astype(np.uint8)
    if isBGR:
        pred = pred[:, :, ::-1].copy()
    return pred


class _Pending:
    """One submitted forward of ``PairStreams``: ``result()`` -> (outputs, event), re-raising what the forward raised."""
    __slots__ = ("_done", "out", "event", "error")

    def __init__(self):
        import threading
        self._done, self.out, self.event, self.error = threading.Event(), None, None, None

    def result(self):
        self._done.wait()
        if self.error is not None:
            raise self.error
        return self.out, self.event


class PairStreams:
    """K independent forwards in flight: K replicas of the model (``Network.replica()``: shared parameters and packed weights, own
    workspace + launch plan each) on K streams, frame pairs handed out round-robin.  The hot path's parallel axis is the pair axis
    (demo_2x.py:129-168: consecutive pairs; benchmark/test_vimeo90k.py:80-110: independent triplets), and a forward of a small frame
    is a chain of ~116 short kernels that each fill a few dozen of the 256 CUs: K chains side by side fill the chip.  There is NO
    cross-stream event inside a forward (a replica touches only its own buffers); the only ordering is at the ends -- the stream
    waits for the caller's stream once before a forward (its inputs) and an event marks the outputs ready.  Every result is
    bit-identical to ``model.forward`` on one stream.  Memory: K workspaces (13 GB each at 1080p, ~0.1 GB at 256 x 256).

    ``threads=True``: every stream gets a worker thread that issues its forwards (a planned forward is ONE C call that issues ~116
    kernel launches; ctypes drops the GIL for it).  Measured (tools/inflight_ab.py, profiles/r06_inflight_ab.txt): no consistent
    gain over one issuing thread once the per-forward allocator bookkeeping is off the path (256 x 256, K = 4: 2 081 frames/s from
    one thread, 1 720-2 180 with four), so it is off by default."""

    def __init__(self, model, k: int = 3, threads: bool = False):
        ops, dev = _hip_ops_of(model)
        if ops is None:
            raise RuntimeError("PairStreams needs an atm-vfi_amd Network on the GPU")
        if k < 1:
            raise ValueError("PairStreams: k >= 1")
        self.model, self.dev, self.k = model, dev, int(k)
        self.replicas = [model.replica() for _ in range(self.k)]
        self.streams = [torch.cuda.Stream(dev) for _ in range(self.k)]
        self._n = 0
        self._queues, self._workers = None, []
        if threads:
            import queue
            import threading
            self._queues = [queue.SimpleQueue() for _ in range(self.k)]
            for i in range(self.k):
                t = threading.Thread(target=self._work, args=(i,), name=f"atmvfi-pairstream-{i}", daemon=True)
                t.start()
                self._workers.append(t)

    def _sync_flags(self):
        for r in self.replicas:        # the switches a caller flips on the model between calls (demo_2x.py:126, the ensemble flag)
            r.global_motion, r.ensemble_global_motion = self.model.global_motion, self.model.ensemble_global_motion
            for name in ("local_motion_args", "global_motion_args"):
                getattr(r, name)["window_size"] = getattr(self.model, name)["window_size"]

    def _forward_on(self, i, im0, im1, in_event, consumer):
        st = self.streams[i]
        if in_event is not None:
            st.wait_event(in_event)
        with torch.cuda.stream(st):
            out = self.replicas[i].forward(im0, im1)
            ev = torch.cuda.Event()
            ev.record(st)
        for t in (im0, im1):
            t.record_stream(st)
        if consumer is not None:       # the caller will use (and free) the outputs on its own stream
            for v in out.values():
                for t in (v if isinstance(v, (list, tuple)) else (v,)):
                    if torch.is_tensor(t):
                        t.record_stream(consumer)
        return out, ev

    def _work(self, i):
        torch.cuda.set_device(self.dev)
        torch.set_grad_enabled(False)
        q = self._queues[i]
        while True:
            item = q.get()
            if item is None:
                return
            pend, im0, im1, in_event, consumer = item
            try:
                pend.out, pend.event = self._forward_on(i, im0, im1, in_event, consumer)
            except BaseException as e:          # handed to the caller by result()
                pend.error = e
            pend._done.set()

    def submit(self, im0, im1, wait_inputs: bool = True, record_outputs: bool = True) -> "_Pending":
        """Enqueue ``forward(im0, im1)`` on the next stream; ``.result()`` of the returned handle is ``(outputs, event)`` -- the outputs
        may be read (by the host, or by a stream that waited for the event) once the event has completed.  ``wait_inputs``: order the
        forward after everything queued so far on the caller's current stream (False when the frames are known to be resident and
        ready).  ``record_outputs``: tell torch's allocator that the caller's stream will use the output tensors (needed when they are
        consumed by kernels on that stream and freed while those are still queued; not when the host reads them)."""
        i = self._n % self.k
        self._n += 1
        cur = torch.cuda.current_stream(self.dev)
        in_event = None
        if wait_inputs:
            in_event = torch.cuda.Event()
            in_event.record(cur)
        pend = _Pending()
        if self._queues is None:
            try:
                pend.out, pend.event = self._forward_on(i, im0, im1, in_event, cur if record_outputs else None)
            except BaseException as e:
                pend.error = e
            pend._done.set()
        else:
            self._queues[i].put((pend, im0, im1, in_event, cur if record_outputs else None))
        return pend

    def map(self, pairs, wait_inputs: bool = True, record_outputs: bool = True):
        """``forward`` over an iterable of (im0, im1) with up to 2 K forwards submitted (K executing, K queued behind them); yields the
        output dicts in order, each complete."""
        from collections import deque
        self._sync_flags()
        q = deque()
        depth = 2 * self.k if self._queues is not None else self.k
        for a, b in pairs:
            if len(q) == depth:
                out, ev = q.popleft().result()
                ev.synchronize()
                yield out
            q.append(self.submit(a, b, wait_inputs, record_outputs))
        while q:
            out, ev = q.popleft().result()
            ev.synchronize()
            yield out

    def synchronize(self):
        for st in self.streams:
            st.synchronize()

    def release(self):
        """Stop the workers, wait for the streams, free the K workspaces."""
        if self._queues is not None:
            for q in self._queues:
                q.put(None)
            for t in self._workers:
                t.join()
            self._queues, self._workers = None, []
        self.synchronize()
        for r in self.replicas:
            r.release_workspace()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.release()
        return False


class FramePipeline:
    """uint8 frame pairs in host memory -> uint8 interpolated frames, with the transfers off the critical path.

    ``depth`` slots of pinned host buffers and device staging; slot i+1's host->device copy and slot i-1's device->host copy run
    on their own streams under slot i's forward.  Per pair over PCIe: 2 * H*W*3 bytes in, H*W*3 bytes out (uint8, 4x less than the
    fp32 tensors the reference moves).  ``run`` yields frames in order; a yielded array is a fresh copy.  Measured at 1080p on one
    MI355X (tools/bench_hostio.py, tools/diag_hostio.py): 25.0 ms resident, 26.9 ms per frame at depth 1 and 25.2 ms at depth 3 -- the
    pieces are small (H2D 1.0 ms, enqueue 2.8 ms of CPU, pre/post kernels < 0.1 ms) and at depth 3 the GPU queue never runs dry
    (gap between consecutive forwards 0.03 ms)."""

    def __init__(self, model, height: int, width: int, isBGR: bool = True, divisor: int = 64, depth: int = 3, streams: int = 1):
        ops, dev = _hip_ops_of(model)
        if ops is None:
            raise RuntimeError("FramePipeline needs an atm-vfi_amd Network on the GPU")
        # streams > 1: K forwards in flight (PairStreams: K replicas on K streams); pair i's whole chain -- pre-kernels, forward,
        # post-kernel -- runs on stream i % K, so nothing inside it crosses streams.  Small frames only fill the chip this way.
        self.lanes = PairStreams(model, streams) if streams > 1 else None
        depth = max(depth, streams + 1) if streams > 1 else depth
        self.model, self.ops, self.dev, self.bgr, self.depth = model, ops, dev, bool(isBGR), max(1, depth)
        self._k = 0
        self.h, self.w = height, width
        pad = InputPadder((1, 3, height, width), divisor=divisor)
        self.pad_left, _, self.pad_top, _ = pad._pad
        self.hp, self.wp = height + pad._pad[2] + pad._pad[3], width + pad._pad[0] + pad._pad[1]
        mk = lambda *s, dt: torch.empty(*s, dtype=dt, device=dev)
        self.slots = [{
            "h_in": torch.empty(2, height, width, 3, dtype=torch.uint8).pin_memory(),
            "h_out": torch.empty(height, width, 3, dtype=torch.uint8).pin_memory(),
            "d_in": mk(2, height, width, 3, dt=torch.uint8), "d_out": mk(height, width, 3, dt=torch.uint8),
            "f0": mk(1, 3, self.hp, self.wp, dt=torch.float32), "f1": mk(1, 3, self.hp, self.wp, dt=torch.float32),
            "in_ready": torch.cuda.Event(), "done": torch.cuda.Event(), "out_ready": torch.cuda.Event(),
        } for _ in range(self.depth)]
        for slot in self.slots:
            slot["h_in_np"] = slot["h_in"].numpy()
        self.copy_in, self.copy_out = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def _upload(self, slot, pair):
        a, b = pair
        if a.shape != (self.h, self.w, 3) or b.shape != (self.h, self.w, 3) or a.dtype != np.uint8 or b.dtype != np.uint8:
            raise ValueError(f"FramePipeline: expected two uint8 [{self.h},{self.w},3] frames")
        # numpy's single-threaded memcpy, not Tensor.copy_: ATen spreads a 6 MB copy over every core it sees (128 threads on the
        # GPU box), which trips the container's CPU quota and stalls the enqueueing thread for tens of ms (tools/diag_hostio.py)
        np.copyto(slot["h_in_np"][0], a)
        np.copyto(slot["h_in_np"][1], b)
        with torch.cuda.stream(self.copy_in):
            slot["d_in"].copy_(slot["h_in"], non_blocking=True)
            slot["in_ready"].record(self.copy_in)

    def _compute(self, slot):
        if self.lanes is not None:
            i = self._k % self.lanes.k
            self._k += 1
            with torch.cuda.stream(self.lanes.streams[i]):
                self._compute_on(slot, self.lanes.replicas[i])
        else:
            self._compute_on(slot, self.model)

    def _compute_on(self, slot, model):
        cur = torch.cuda.current_stream(self.dev)
        cur.wait_event(slot["in_ready"])
        self.ops.frame_u8_to_f32(slot["d_in"][0], slot["f0"][0], self.pad_top, self.pad_left, self.bgr)
        self.ops.frame_u8_to_f32(slot["d_in"][1], slot["f1"][0], self.pad_top, self.pad_left, self.bgr)
        it = model.forward(slot["f0"], slot["f1"])["I_t"]
        self.ops.frame_f32_to_u8(it[0], slot["d_out"], self.pad_top, self.pad_left, self.bgr)
        slot["done"].record(cur)
        self.copy_out.wait_event(slot["done"])
        with torch.cuda.stream(self.copy_out):
            slot["h_out"].copy_(slot["d_out"], non_blocking=True)
            slot["out_ready"].record(self.copy_out)

    def run(self, pairs):
        if self.lanes is not None:
            self.lanes._sync_flags()
        it = iter(pairs)
        inflight = []                      # slots whose compute has been enqueued, oldest first
        nxt = next(it, None)
        k = 0
        if nxt is not None:
            self._upload(self.slots[0], nxt)
        while nxt is not None or inflight:
            if nxt is not None:
                slot = self.slots[k % self.depth]
                self._compute(slot)
                inflight.append(slot)
                k += 1
                nxt = next(it, None)
                if nxt is not None:
                    if len(inflight) == self.depth:          # the next slot is still owned by the oldest pair: deliver it first
                        old = inflight.pop(0)
                        old["out_ready"].synchronize()
                        yield old["h_out"].numpy().copy()
                    self._upload(self.slots[k % self.depth], nxt)      # overlaps the forward just enqueued
                    continue
            old = inflight.pop(0)
            old["out_ready"].synchronize()
            yield old["h_out"].numpy().copy()


def interpolate_video_2x(frames, model, isBGR: bool = True, divisor: int = 64, depth: int = 3, streams: int = 1):
    """The frame loop of demo_2x.py:144-163 over any iterable of uint8 [H,W,3] frames (decoding / encoding stays with the caller):
    yields f0, I(f0,f1), f1, I(f1,f2), ..., f_{n-1} -- 2n-1 frames -- with the pairs running through ``FramePipeline``
    (``streams`` > 1: that many forwards in flight on streams of their own, for frames too small to fill the GPU one at a time)."""
    from collections import deque
    it = iter(frames)
    first = next(it, None)
    if first is None:
        return
    originals = deque([first])

    def pairs():
        prev = first
        for cur in it:
            originals.append(cur)
            yield prev, cur
            prev = cur
    pipe = FramePipeline(model, first.shape[0], first.shape[1], isBGR=isBGR, divisor=divisor, depth=depth, streams=streams)
    for pred in pipe.run(pairs()):
        yield originals.popleft()
        yield pred
    yield originals.popleft()          # the last frame is written once (demo_2x.py:160)


# cv2.VideoCapture property ids (cv2.CAP_PROP_*; the image has no OpenCV, and the adapters below only need the numbers)
CAP_PROP_FRAME_WIDTH, CAP_PROP_FRAME_HEIGHT, CAP_PROP_FPS, CAP_PROP_FRAME_COUNT = 3, 4, 5, 7


def capture_frames(cap):
    """The decode end of demo_2x.py:129-163 as an iterator: ``cap`` is a ``cv2.VideoCapture`` or anything with its ``isOpened()`` /
    ``read() -> (ok, frame)`` pair; yields uint8 [H,W,3] frames until a read fails (the reference's loop exit, :159-163).  Each frame
    is copied (``prev_frame = curr_frame.copy()``, :156: OpenCV may reuse its buffer)."""
    while cap.isOpened():
        ok, frame = cap.read()
        if not ok:
            break
        yield np.array(frame, copy=True)


def video_2x(cap, make_writer, model, isBGR: bool = True, divisor: int = 64, depth: int = 3, interpolator=None, **kw):
    """The video branch of demo_2x.py:129-168 end to end over a capture / writer pair: reads FPS, W, H from ``cap``
    (``cap.get(CAP_PROP_*)``, :130-133), opens the sink with ``make_writer(2 * FPS, (W, H))`` (:134-135: the processed video plays at
    twice the rate), writes f0, I(f0,f1), f1, ..., f_{n-1} -- every original once, the last frame once (:148-150, 160) -- and releases
    both ends (:165-166).  The codec stays with the caller: with OpenCV, ``cap = cv2.VideoCapture(path)`` and ``make_writer = lambda
    fps, size: cv2.VideoWriter(out, cv2.VideoWriter_fourcc(*'mp4v'), fps, size)``.  ``interpolator(frames, model, ...)`` defaults to
    ``interpolate_video_2x`` (the pipelined HIP path).  Returns ``{"fps_in", "fps_out", "size", "frames_in", "frames_out"}``.
    Not provided: ``--combine_video`` (cv2.putText drawing, :88-97).  An empty video writes nothing (the reference raises NameError)."""
    fps = int(cap.get(CAP_PROP_FPS))
    w, h = int(cap.get(CAP_PROP_FRAME_WIDTH)), int(cap.get(CAP_PROP_FRAME_HEIGHT))
    out = make_writer(2 * fps, (w, h))
    n_in = [0]

    def counted():
        for f in capture_frames(cap):
            if f.shape[:2] != (h, w):
                raise ValueError(f"video_2x: the capture announced {w}x{h} frames and delivered {f.shape[1]}x{f.shape[0]}")
            n_in[0] += 1
            yield f
    n_out = 0
    try:
        for frame in (interpolator or interpolate_video_2x)(counted(), model, isBGR=isBGR, divisor=divisor, depth=depth, **kw):
            out.write(frame)
            n_out += 1
    finally:
        cap.release()
        out.release()
    return {"fps_in": fps, "fps_out": 2 * fps, "size": (w, h), "frames_in": n_in[0], "frames_out": n_out}


def interpolate_video_2x_distributed(frames, model, rank: int, world: int, isBGR: bool = True, divisor: int = 64, block: int = 4,
                                     group=None):
    """``interpolate_video_2x`` over the GPUs of one node (one process per GPU, ``torch.distributed`` initialised by the caller):
    rounds of ``world * block`` consecutive pairs, rank r interpolating ``block`` consecutive ones; the uint8 predictions of a round
    are all-gathered one step behind the compute (``sharding.HostGather``: 6 MB per 1080p frame over xGMI instead of 25 MB of fp32;
    the collective and the device -> pinned-host copies run on a side stream), and EVERY rank yields the full 2n-1 sequence f0,
    I(f0,f1), f1, ... as uint8 [H,W,3] arrays, in order.  ``frames``: a sequence of uint8 [H,W,3] frames every rank can index.
    Uploads go through pinned slots on a copy stream, one frame ahead of the forward that needs it.  Inside a block consecutive
    pairs share a frame: its device copy is reused and so is everything ``forward`` computes per frame
    (``Network.enable_frame_cache``: encoder, fusions and, with the global branch on, its per-frame half)."""
    from . import sharding
    ops, dev = _hip_ops_of(model)
    if ops is None:
        raise RuntimeError("interpolate_video_2x_distributed needs an atm-vfi_amd Network on the GPU")
    n = len(frames)
    if n == 0:
        return
    h, w = frames[0].shape[:2]
    pad = InputPadder((1, 3, h, w), divisor=divisor)
    pad_left, _, pad_top, _ = pad._pad
    hp, wp = h + pad._pad[2] + pad._pad[3], w + pad._pad[0] + pad._pad[1]
    fbuf = [torch.empty(1, 3, hp, wp, dtype=torch.float32, device=dev) for _ in range(2)]
    out_u8 = torch.empty(h, w, 3, dtype=torch.uint8, device=dev)
    use_cache = hasattr(model, "enable_frame_cache") and not getattr(model, "ensemble_global_motion", False)
    if use_cache:
        model.enable_frame_cache(True)
    # Uploads as in FramePipeline: a ring of pinned host slots + device staging, filled on a copy stream ONE FRAME AHEAD of the
    # forward that needs it (this rank's pairs, in order, are known up front from the block schedule), so a frame's host -> device
    # copy runs under the previous pair's forward instead of in front of its own.
    mine = [i for _, spans in sharding.shard_blocks(n - 1, world, block) for i in range(*spans[rank])]
    need = []                          # frame indices in upload order: both frames of a block's first pair, then one per pair
    for k, i in enumerate(mine):
        if k == 0 or mine[k - 1] != i - 1:
            need.append(i)
        need.append(i + 1)
    depth = 3
    ring = [{"h": torch.empty(h, w, 3, dtype=torch.uint8).pin_memory(), "d": torch.empty(h, w, 3, dtype=torch.uint8, device=dev),
             "ready": torch.cuda.Event(), "free": torch.cuda.Event()} for _ in range(depth)]
    for slot in ring:
        slot["h_np"] = slot["h"].numpy()
    copy_in = torch.cuda.Stream(dev)
    state = {"cur": 0, "issued": 0, "used": 0}     # fbuf[cur] holds the previous pair's second frame

    def prefetch():
        """Issue uploads until ``depth - 1`` frames are in flight ahead of the consumer (one slot may still be read by a kernel)."""
        while state["issued"] < len(need) and state["issued"] - state["used"] < depth - 1:
            slot = ring[state["issued"] % depth]
            if state["issued"] >= depth:
                slot["free"].synchronize()                    # the pre-kernel that read this slot's device copy has run
            fr = frames[need[state["issued"]]]
            if fr.shape != (h, w, 3) or fr.dtype != np.uint8:
                raise ValueError(f"interpolate_video_2x_distributed: expected uint8 [{h},{w},3] frames")
            np.copyto(slot["h_np"], fr)                       # numpy's single-threaded memcpy (see FramePipeline._upload)
            with torch.cuda.stream(copy_in):
                slot["d"].copy_(slot["h"], non_blocking=True)
                slot["ready"].record(copy_in)
            state["issued"] += 1

    def to_device(idx, dst):
        assert need[state["used"]] == idx, "upload schedule out of step with the pair order"
        prefetch()
        slot = ring[state["used"] % depth]
        cur = torch.cuda.current_stream(dev)
        cur.wait_event(slot["ready"])
        ops.frame_u8_to_f32(slot["d"], dst[0], pad_top, pad_left, bool(isBGR))
        slot["free"].record(cur)
        state["used"] += 1
        prefetch()                                            # the next frame's upload overlaps this pair's forward

    pos = {"k": 0}

    def pair(fa, fb, reuse_first):
        i = mine[pos["k"]]
        pos["k"] += 1
        a, b = fbuf[state["cur"]], fbuf[state["cur"] ^ 1]
        if not reuse_first:
            to_device(i, a)
        to_device(i + 1, b)
        it = (model.forward(a, b, reuse_first=reuse_first) if use_cache else model.forward(a, b))["I_t"]
        ops.frame_f32_to_u8(it[0], out_u8, pad_top, pad_left, bool(isBGR))
        state["cur"] ^= 1             # fb is the next pair's fa
        return out_u8
    try:
        for item in sharding.interpolate_video_2x_sharded(frames, pair, rank, world, (h, w, 3), torch.uint8, block=block,
                                                          device=dev, group=group, host_gather=True):
            yield item
    finally:
        if use_cache:
            model.enable_frame_cache(False)


def forward_tta(model, im0, im1):
    """Flip test-time augmentation of benchmark/test_snufilm.py:135-139: average of the prediction and the un-flipped prediction
    on the frames flipped along H and W.  Returns ``I_t`` [B,3,H,W]."""
    pred = model.forward(im0, im1)["I_t"]
    pred_flip = model.forward(im0.flip(2).flip(3).contiguous(), im1.flip(2).flip(3).contiguous())["I_t"]
    return (pred + pred_flip.flip(2).flip(3)) / 2


def tile_plan(height: int, width: int, tile: tuple, overlap: int):
    """The tiles of ``forward_tiled``: the frame is cut into a grid of CORE rectangles of at most ``tile`` = (th, tw) pixels (equal
    sizes up to rounding, rows x columns = ceil(H / th) x ceil(W / tw)); a tile is its core grown by ``overlap`` pixels on every
    side that has a neighbour (clipped to the frame).  Returns [(y0, y1, x0, x1, cy0, cy1, cx0, cx1)]: the tile's rectangle in
    the frame and its core's rectangle in the frame, row-major."""
    th, tw = tile
    if th <= 0 or tw <= 0 or overlap < 0:
        raise ValueError("tile_plan: tile sizes must be positive and the overlap non-negative")
    ny, nx = -(-height // th), -(-width // tw)
    ys = [round(i * height / ny) for i in range(ny + 1)]
    xs = [round(j * width / nx) for j in range(nx + 1)]
    plan = []
    for i in range(ny):
        for j in range(nx):
            cy0, cy1, cx0, cx1 = ys[i], ys[i + 1], xs[j], xs[j + 1]
            plan.append((max(cy0 - overlap, 0), min(cy1 + overlap, height), max(cx0 - overlap, 0), min(cx1 + overlap, width),
                         cy0, cy1, cx0, cx1))
    return plan


def forward_tiled(forward, im0, im1, tile=(1088, 2048), overlap: int = 64, divisor: int = 64, streams: int = 1):
    """TILED inference for frames too large to run at once (BASELINE.json configs[4]: "Xiph-4K 2160x4096 tiled").  The reference
    has no tiling (its "4K" is a centre crop, test_xiph.py:115-123), so the mode is defined HERE, and its parity
    oracle is the reference's forward run on the identical tiles with the identical stitching -- which is what the tests
    do by passing the oracle's forward as ``forward``:
      * tiles = ``tile_plan(H, W, tile, overlap)``; every tile of both frames is replicate-padded to a multiple of ``divisor`` with
        the reference's own InputPadder (benchmark/utils.py:57-80), run through ``forward`` and un-padded;
      * the prediction's CORE rectangle is written into the output frame (cores partition the frame: no blending, no seams in the
        sense of double coverage; the overlap only gives every core pixel ``overlap`` pixels of real context).
    ``forward``: a ``Network`` (its ``I_t`` is taken; ``streams`` > 1 keeps that many tiles in flight through a ``PairStreams`` made
    for this call -- its replicas build their workspaces and plans first, so for repeated use pass a ``PairStreams`` you keep as
    ``forward`` instead), a ``PairStreams``, or any callable (im0, im1) -> I_t or the output dict.  Returns ``I_t`` [B,3,H,W] on the frames' device.  Per tile the workspace is that of the tile's
    size (14 GB at 1152 x 2112) instead of the whole frame's (52 GB at 2176 x 4096)."""
    if im0.shape != im1.shape or im0.dim() != 4:
        raise ValueError(f"forward_tiled: two [B,3,H,W] frames expected, got {tuple(im0.shape)} and {tuple(im1.shape)}")
    b, c, h, w = im0.shape
    plan = tile_plan(h, w, tile, overlap)
    out = torch.empty_like(im0, memory_format=torch.contiguous_format)
    is_net = hasattr(forward, "replica") and hasattr(forward, "forward")

    def pieces():
        for (y0, y1, x0, x1, *_core) in plan:
            a, bb = im0[..., y0:y1, x0:x1], im1[..., y0:y1, x0:x1]
            padder = InputPadder(a.shape, divisor=divisor)
            pa, pb = padder.pad(a, bb)
            yield padder, pa.contiguous(), pb.contiguous()

    def place(k, padder, pred):
        y0, y1, x0, x1, cy0, cy1, cx0, cx1 = plan[k]
        pred = padder.unpad(pred)
        out[..., cy0:cy1, cx0:cx1] = pred[..., cy0 - y0:cy1 - y0, cx0 - x0:cx1 - x0]

    own = is_net and streams > 1 and im0.is_cuda
    if own or isinstance(forward, PairStreams):
        padders = []

        def pairs_():
            for padder, pa, pb in pieces():
                padders.append(padder)
                yield pa, pb
        ps = PairStreams(forward, min(streams, len(plan))) if own else forward
        try:
            for k, res in enumerate(ps.map(pairs_())):
                place(k, padders[k], res["I_t"])
        finally:
            if own:
                ps.release()
        return out
    for k, (padder, pa, pb) in enumerate(pieces()):
        res = forward(pa, pb)
        place(k, padder, res["I_t"] if isinstance(res, dict) else res)
    return out


def psnr(a, b) -> float:
    """-10 log10(mean((a - b)^2)) on [0,1] images (benchmark/psnr_ssim.py:133-135, test_snufilm.py:147)."""
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    mse = ((a - b) ** 2).mean().item()
    return float("inf") if mse == 0 else -10.0 * float(np.log10(mse))


def save_checkpoint(model, path, optimizer=None, meta=None, train_metric=None, val_metric=None):
    """The trainer's checkpoint wire format (trainer.py:438-446): a 5-key dict that every loader of the reference understands."""
    torch.save({"model_state_dict": model.state_dict(),
                "optimizer_state_dict": optimizer.state_dict() if optimizer is not None else {},
                "meta_data": meta or {}, "train_metric": train_metric or {}, "val_metric": val_metric or {}}, path)


def strip_lazy_buffers(state):
    """Saved checkpoints carry the reference's lazily registered ``attn_mask``/``HW`` buffers
    (attention.py:304-305); every loader of the reference drops them (demo_2x.py:38-46)."""
    return {k: v for k, v in state.items() if "attn_mask" not in k and "HW" not in k}


def load_model_checkpoint(model, checkpoint_path, strict: bool = True, map_location=None):
    """demo_2x.py:24-51 -- accepts the trainer's 5-key dict or a bare state dict."""
    ck = torch.load(checkpoint_path, map_location=map_location or "cpu")
    optim = None
    if isinstance(ck, dict) and "model_state_dict" in ck:
        param = ck["model_state_dict"]
        optim = ck.get("optimizer_state_dict")
    else:
        param = ck
    model.load_state_dict(strip_lazy_buffers(param), strict=strict)
    return optim
