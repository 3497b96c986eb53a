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
