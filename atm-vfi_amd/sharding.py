"""Frame-batch data parallelism over the GPUs of one node.

Every frame pair is independent (no cross-sample op in ``forward``), so the path shards
with no data-path collective: rank r interpolates pairs ``r::world``.  The only exchange
is one all-gather of the per-rank output frames (RCCL over xGMI when the backend is
"nccl"; gloo on CPU in the tests).  Weights are replicated (206 MB for base)."""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch
import torch.distributed as dist


def shard_indices(n_pairs: int, rank: int, world: int) -> List[int]:
    return list(range(rank, n_pairs, world))


def gather_frames(local: torch.Tensor, n_pairs: int, rank: int, world: int) -> torch.Tensor:
    """all-gather ``[n_local,3,H,W]`` outputs back into pair order ``[n_pairs,3,H,W]``.
    Ranks may hold different counts (ragged tail): shorter ranks are padded for the
    collective and the padding is dropped afterwards."""
    if world == 1:
        return local
    per = (n_pairs + world - 1) // world
    shape = (per,) + tuple(local.shape[1:])
    send = local.new_zeros(shape)
    send[:local.shape[0]] = local
    parts = send.new_empty((world,) + shape)          # one receive buffer: every rank's frames land where they are read from
    dist.all_gather_into_tensor(parts, send.contiguous().unsqueeze(0))
    out = local.new_empty((n_pairs,) + tuple(local.shape[1:]))
    for r in range(world):
        idx = shard_indices(n_pairs, r, world)
        if idx:
            out[idx] = parts[r][:len(idx)]
    return out


def interpolate_sharded(forward: Callable, im0: torch.Tensor, im1: torch.Tensor, rank: int, world: int,
                        micro_batch: int = 1) -> torch.Tensor:
    """Run ``forward(im0[i], im1[i])['I_t']`` for this rank's pairs and all-gather the frames."""
    n = im0.shape[0]
    mine = shard_indices(n, rank, world)
    outs = []
    for s in range(0, len(mine), micro_batch):
        idx = mine[s:s + micro_batch]
        outs.append(forward(im0[idx], im1[idx])["I_t"])
    local = torch.cat(outs, 0) if outs else im0.new_zeros((0,) + tuple(im0.shape[1:]))
    return gather_frames(local, n, rank, world)


class PipelinedGather:
    """The benchmark's and the video path's collective: all-gather of per-rank output frames, ONE STEP BEHIND the compute.

    ``submit(local)`` copies this rank's frames of step k into a send buffer of its own (so the producer -- e.g. a captured HIP
    graph's static output -- may be overwritten by step k+1 straight away), issues the all-gather asynchronously (RCCL runs it on
    its own stream, under the forward of step k+1) and returns the gathered frames of step k-1, waiting for that older collective
    first; ``drain()`` returns the last step's.  Two send buffers alternate (a send buffer is reused two steps later, after its
    gather was waited for); THREE sets of receive buffers rotate, because the gather of step k+1 is issued by the very submit that
    hands out step k-1's frames... of another set: the frames returned by ``submit`` number k live in set (k-1) mod 3, which the
    gather of step k+2 overwrites.  A rank without a frame in the last (ragged) step submits ``None``: it sends zeros, and ``valid`` counts say
    how many ranks' frames are real.  ``encode`` (optional) maps the fp32 frames to the wire format before sending, e.g. rounding
    to uint8 (4x fewer bytes over xGMI); it must return a tensor of ``wire_shape`` / ``wire_dtype``.
    Frames come back as a list of ``world`` tensors in rank order: views of the receive buffers, valid until the NEXT-BUT-ONE
    ``submit`` (they survive one further submit; tests/test_sharding_gloo.py reads them after it).

    The collective is ``all_gather_into_tensor`` into ONE ``[world, *wire_shape]`` receive buffer per set: RCCL writes every rank's
    frame where it is read from (``all_gather`` into a tensor LIST gathers into a flat staging buffer and copies out per rank:
    ``world`` extra device copies per step); gloo takes the same call."""

    def __init__(self, world: int, wire_shape, device, wire_dtype=torch.float32, encode: Callable = None, group=None):
        self.world, self.encode, self.group = world, encode, group
        wire_shape = tuple(wire_shape)
        self._send = [torch.zeros((1,) + wire_shape, dtype=wire_dtype, device=device) for _ in range(2)]
        self.send = [t[0] for t in self._send]
        self._recv = [torch.empty((world,) + wire_shape, dtype=wire_dtype, device=device) for _ in range(3)]
        self.recv = [[t[i] for i in range(world)] for t in self._recv]       # per set: rank-order views of the one buffer
        self.pending = None            # (work, buffer set, valid count)
        self.k = 0

    def _issue(self, r: int, s: int):
        return dist.all_gather_into_tensor(self._recv[r], self._send[s], group=self.group, async_op=True)

    def _finish(self):
        if self.pending is None:
            return None
        work, r, valid = self.pending
        self.pending = None
        if work is not None:
            work.wait()
        return self.recv[r][:valid]

    def submit(self, local, valid: int = None):
        """local: this rank's frames of the step (wire_shape after ``encode``) or None; valid: ranks holding a real frame this
        step (default: all).  Returns the previous step's gathered frames (list, rank order) or None on the first call."""
        s, r = self.k & 1, self.k % 3
        self.k += 1
        if local is None:
            self.send[s].zero_()
        else:
            self.send[s].copy_(self.encode(local) if self.encode is not None else local)
        prev = self._finish()                       # the older collective first: its receive buffers are about to be handed out
        if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
            self.recv[r][0].copy_(self.send[s])
            work = None
        else:
            work = self._issue(r, s)
        self.pending = (work, r, self.world if valid is None else valid)
        return prev

    def drain(self):
        return self._finish()


class HostGather(PipelinedGather):
    """``PipelinedGather`` for a consumer on the HOST (the video loop): the all-gather and the device -> pinned-host copies of the
    gathered frames run on a stream of their own, behind an event recorded when this rank's frames were produced -- NOT behind
    whatever the compute stream has queued since.  ``submit`` / ``drain`` return the previous step's frames as numpy arrays (rank
    order), blocking the host only until THAT step's gather and copies are done, so the next round's forwards (already enqueued)
    keep the GPU busy meanwhile.  With ``t.cpu()`` on the compute stream instead, every round's delivery waited for the following
    round's whole compute and the GPU then idled while the host consumed the frames.  On CPU tensors (gloo tests) it degrades to
    the plain gather + ``numpy()``."""

    def __init__(self, world: int, wire_shape, device, wire_dtype=torch.float32, encode: Callable = None, group=None):
        super().__init__(world, wire_shape, device, wire_dtype, encode, group)
        self.cuda = torch.device(device).type == "cuda"
        if self.cuda:
            self.stream = torch.cuda.Stream(device)
            self.host = [[torch.empty(wire_shape, dtype=wire_dtype).pin_memory() for _ in range(world)] for _ in range(3)]

    def _finish(self):
        if self.pending is None:
            return None
        done, r, valid = self.pending
        self.pending = None
        if not self.cuda:
            if done is not None:
                done.wait()
            return [t.numpy().copy() for t in self.recv[r][:valid]]
        done.synchronize()
        return [t.numpy().copy() for t in self.host[r][:valid]]

    def submit(self, local, valid: int = None):
        if not self.cuda:
            s, r = self.k & 1, self.k % 3
            self.k += 1
            if local is None:
                self.send[s].zero_()
            else:
                self.send[s].copy_(self.encode(local) if self.encode is not None else local)
            prev = self._finish()
            work = None
            if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
                self.recv[r][0].copy_(self.send[s])
            else:
                work = self._issue(r, s)
            self.pending = (work, r, self.world if valid is None else valid)
            return prev
        s, r = self.k & 1, self.k % 3
        self.k += 1
        cur = torch.cuda.current_stream(self.send[s].device)
        if local is None:
            self.send[s].zero_()
        else:
            self.send[s].copy_(self.encode(local) if self.encode is not None else local)
        produced = torch.cuda.Event()
        produced.record(cur)
        prev = self._finish()                   # host waits for step k-1's gather + copies (set r of step k-1 is not this one)
        nv = self.world if valid is None else valid
        with torch.cuda.stream(self.stream):
            self.stream.wait_event(produced)
            if self.world == 1 and not (dist.is_available() and dist.is_initialized()):
                self.recv[r][0].copy_(self.send[s], non_blocking=True)
            else:
                work = self._issue(r, s)
                work.wait()                     # RCCL: this side stream waits for the collective; gloo: the host does
            for i in range(nv):
                self.host[r][i].copy_(self.recv[r][i], non_blocking=True)
            done = torch.cuda.Event()
            done.record(self.stream)
        self.pending = (done, r, nv)
        return prev


def shard_blocks(n_items: int, world: int, block: int):
    """Rounds of ``world * block`` consecutive items; in a round rank r owns the ``block`` consecutive items
    ``[round_start + r * block, +block)`` (clipped to ``n_items``).  Consecutive items on one rank let the video path reuse the
    shared frame of consecutive pairs.  Yields (round_start, [per-rank (start, stop)])."""
    step = world * block
    for start in range(0, n_items, step):
        spans = []
        for r in range(world):
            a = min(n_items, start + r * block)
            b = min(n_items, a + block)
            spans.append((a, b))
        yield start, spans


def interpolate_video_2x_sharded(frames: Sequence, interpolate_pair: Callable, rank: int, world: int, wire_shape, wire_dtype=torch.uint8,
                                 block: int = 1, decode: Callable = None, device="cpu", group=None, host_gather: bool = False):
    """demo_2x.py:129-168's frame loop over the GPUs of a node: pair i = (frames[i], frames[i+1]); rounds of ``world * block`` pairs,
    rank r interpolating ``block`` consecutive pairs of each round; the predictions of a round are all-gathered (one step behind the
    next round's compute, PipelinedGather) so that EVERY rank yields the full 2n-1 sequence f0, I(f0,f1), f1, ..., f_{n-1} in order.

    ``frames``: a sequence every rank can index (decoding stays with the caller).  ``interpolate_pair(f_a, f_b, reuse_first)`` returns
    this rank's prediction as a tensor of ``wire_shape`` / ``wire_dtype`` on ``device`` (e.g. the uint8 [H,W,3] frame of
    ``FramePipeline``: 4x fewer bytes over xGMI than fp32); ``reuse_first`` is True when f_a was the previous call's f_b on this
    rank, so an implementation may reuse that frame's encoder features (``Network.enable_frame_cache``).  ``decode`` converts a
    gathered wire tensor to what is yielded (default: the tensor itself).  ``host_gather``: gather through ``HostGather`` -- the
    frames arrive as numpy arrays from pinned host buffers filled on a side stream (``decode`` is then applied to those arrays).

    The collectives are issued from inside this generator, so EVERY rank must run it to the end (a rank that stops iterating early
    leaves the others waiting in their next all-gather until the process group's timeout -- create the group with one); when a
    consumer does abandon the generator (``close()``, an exception, garbage collection) the gather already in flight on this rank
    is waited for before the generator returns, so no un-waited work object outlives it."""
    n = len(frames)
    if n == 0:
        return
    gather = (HostGather if host_gather else PipelinedGather)(world, (block,) + tuple(wire_shape), device, wire_dtype, group=group)

    def emit(start, spans, got):
        i = start
        for r, (a, b) in enumerate(spans):
            for j in range(b - a):
                yield frames[i]
                f = got[r][j]
                yield decode(f) if decode is not None else (f.copy() if host_gather else f.clone())
                i += 1

    prev_meta = None
    last_b = None
    try:
        for start, spans in shard_blocks(n - 1, world, block):
            a, b = spans[rank]
            local = None
            if b > a:
                local = torch.zeros((block,) + tuple(wire_shape), dtype=wire_dtype, device=device)
                for i in range(a, b):
                    local[i - a].copy_(interpolate_pair(frames[i], frames[i + 1], last_b == i))
                    last_b = i + 1
            got = gather.submit(local)
            if prev_meta is not None:
                yield from emit(prev_meta[0], prev_meta[1], got)
            prev_meta = (start, spans)
        if prev_meta is not None:
            yield from emit(prev_meta[0], prev_meta[1], gather.drain())
        yield frames[n - 1]          # the last frame is written once (demo_2x.py:160)
    finally:
        gather.drain()               # abandoned early: do not leave an un-waited collective behind
