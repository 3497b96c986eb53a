"""CPU, world_size 2 over gloo: the multi-GPU frame-batch path -- shard the
pair axis across ranks, no data-path collective, one all-gather of the output frames."""
import importlib
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _fake_forward(a, b):
    # stands in for Network.forward on CPU: any per-pair function works for the sharding logic
    return {"I_t": 0.25 * a + 0.75 * b.flip(-1)}


def _worker(rank, world, port, n_pairs, q):
    sys.path.insert(0, ROOT)
    sharding = importlib.import_module("atm-vfi_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    im0 = torch.rand(n_pairs, 3, 8, 12, generator=g)
    im1 = torch.rand(n_pairs, 3, 8, 12, generator=g)
    out = sharding.interpolate_sharded(_fake_forward, im0, im1, rank, world, micro_batch=2)
    want = _fake_forward(im0, im1)["I_t"]
    ok = torch.equal(out, want) and sharding.shard_indices(n_pairs, rank, world) == list(range(rank, n_pairs, world))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def _pipelined_worker(rank, world, port, n_steps, q):
    """The function bench.py --gpus N and the sharded video loop run: PipelinedGather, one step behind, with a ragged last step
    and the uint8 wire format."""
    sys.path.insert(0, ROOT)
    sharding = importlib.import_module("atm-vfi_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    for as_u8 in (False, True):
        enc = (lambda x: (x * 255).round().clamp(0, 255).to(torch.uint8)) if as_u8 else None
        pg = sharding.PipelinedGather(world, (1, 3, 6, 10), "cpu", torch.uint8 if as_u8 else torch.float32, encode=enc)
        frame = lambda step, r: torch.full((1, 3, 6, 10), (7 * step + 3 * r + 1) / 64.0)
        static_out = torch.empty(1, 3, 6, 10)          # stands in for a captured graph's static output buffer
        got_steps = []
        held = None                                      # the views submit k returned, read again after submit k+1 (the contract)
        for step in range(n_steps):
            last_ragged = step == n_steps - 1
            mine = None
            valid = world
            if last_ragged:
                valid = 1                                # only rank 0 has a frame in the last step
                if rank == 0:
                    static_out.copy_(frame(step, rank)); mine = static_out
            else:
                static_out.copy_(frame(step, rank)); mine = static_out
            prev = pg.submit(mine, valid)
            static_out.fill_(-1.0)                       # the producer overwrites its output right away: the gather must not see it
            ok = ok and ((prev is None) == (step == 0))
            if held is not None:                         # frames of step - 2, handed out one submit ago: still intact
                for r, t in enumerate(held):
                    want = frame(step - 2, r)
                    ok = ok and torch.equal(t, enc(want) if as_u8 else want)
            held = prev
            if prev is not None:
                got_steps.append([t.clone() for t in prev])
        got_steps.append([t.clone() for t in pg.drain()])
        ok = ok and pg.drain() is None and len(got_steps) == n_steps
        for step, got in enumerate(got_steps):
            nvalid = 1 if step == n_steps - 1 else world
            ok = ok and len(got) == nvalid
            for r, t in enumerate(got):
                want = frame(step, r)
                want = enc(want) if as_u8 else want
                ok = ok and torch.equal(t, want)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def _video_worker(rank, world, port, n_frames, block, q, host_gather=False):
    sys.path.insert(0, ROOT)
    sharding = importlib.import_module("atm-vfi_amd.sharding")
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = [torch.full((4, 5, 3), 10 * i, dtype=torch.uint8) for i in range(n_frames)]
    calls = []

    def pair(fa, fb, reuse_first):
        calls.append((int(fa[0, 0, 0]) // 10, bool(reuse_first)))
        return ((fa.to(torch.int32) + fb.to(torch.int32)) // 2).to(torch.uint8)     # "interpolated" frame 10 i + 5
    out = list(sharding.interpolate_video_2x_sharded(frames, pair, rank, world, (4, 5, 3), torch.uint8, block=block,
                                                     host_gather=host_gather))
    vals = [int(t[0, 0, 0]) for t in out]
    ok = vals == [5 * k for k in range(2 * n_frames - 1)]                            # f0, I01, f1, ..., f_{n-1}: every rank, in order
    # this rank computed exactly its blocks, and reuse is announced exactly for consecutive pairs inside a block
    mine = [i for _, spans in sharding.shard_blocks(n_frames - 1, world, block) for i in range(*spans[rank])]
    ok = ok and [c[0] for c in calls] == mine
    ok = ok and all(reuse == (k > 0 and mine[k - 1] == mine[k] - 1) for k, (_, reuse) in enumerate(calls))
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, bool(ok)))


def _video_worker_host(rank, world, port, n_frames, block, q):
    _video_worker(rank, world, port, n_frames, block, q, host_gather=True)


def _run_world2(target, *args):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, 2, port) + args + (q,)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


@pytest.mark.parametrize("n_steps", [1, 2, 5])
def test_pipelined_gather_world2(n_steps):
    _run_world2(_pipelined_worker, n_steps)


@pytest.mark.parametrize("n_frames,block", [(2, 1), (6, 1), (8, 2), (7, 3), (1, 1)])
def test_sharded_video_frame_order_world2(n_frames, block):
    _run_world2(_video_worker, n_frames, block)


@pytest.mark.parametrize("n_frames,block", [(6, 1), (7, 3)])
def test_sharded_video_through_host_gather_world2(n_frames, block):
    """HostGather (the video path's collective: frames delivered as numpy arrays from host buffers) on CPU tensors over gloo."""
    _run_world2(_video_worker_host, n_frames, block)


@pytest.mark.parametrize("n_pairs", [8, 5, 1])
def test_sharded_interpolation_world2(n_pairs):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, True), (1, True)]


def test_single_rank_is_identity():
    sharding = importlib.import_module("atm-vfi_amd.sharding")
    a, b = torch.rand(3, 3, 4, 4), torch.rand(3, 3, 4, 4)
    assert torch.equal(sharding.interpolate_sharded(_fake_forward, a, b, 0, 1), _fake_forward(a, b)["I_t"])
