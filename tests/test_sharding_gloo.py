"""World size 2 over gloo: the N>1 host path of bench.py (contiguous batch sharding, output order, barrier +
max-over-ranks timing).  On the CPU (-m "not gpu") the operator is a hashlib stand-in; on the GPU box (-m gpu) the same
two ranks drive libcapyhip.so on the one visible card, as two of the driver's N ranks do on an 8-GPU node.  The GPU data
path has no collective; the in-library sharding (capy_set_devices) is tests/test_gpu_multidev.py."""
import os
import socket
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q, use_gpu=False):
    sys.path.insert(0, ROOT)
    import hashlib
    import time

    import torch
    import torch.distributed as dist

    from capycrypt_amd.sharding import sharded_map

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    items = [bytes([i % 251]) * (i * 37 % 5000) for i in range(1001)]  # every rank regenerates the same batch
    dist.barrier()
    t0 = time.perf_counter()
    if use_gpu:
        from capycrypt_amd import _lib, ops

        _lib.check(_lib.lib().capy_set_device(0))  # both ranks on the one card of the GPU box
        op = lambda ms: ops.sha3_batch(ms, 256)  # noqa: E731
    else:
        op = lambda ms: [hashlib.sha3_256(m).digest() for m in ms]  # noqa: E731
    lo, hi, res = sharded_map(op, items, rank, world)
    dist.barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(el, op=dist.ReduceOp.MAX)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, res))
    dist.destroy_process_group()
    if rank == 0:
        q.put((gathered, float(el.item())))


def test_two_rank_sharding_preserves_order():
    _run_two_ranks(False)


@pytest.mark.gpu
def test_two_rank_sharding_through_the_library():
    _run_two_ranks(True)


def _run_two_ranks(use_gpu):
    import hashlib

    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, use_gpu)) for r in range(2)]
    for p in procs:
        p.start()
    gathered, el = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    items = [bytes([i % 251]) * (i * 37 % 5000) for i in range(1001)]
    out = []
    expect_lo = 0
    for lo, hi, res in gathered:
        assert lo == expect_lo
        expect_lo = hi
        out.extend(res)
    assert expect_lo == len(items)
    assert out == [hashlib.sha3_256(m).digest() for m in items]
    assert el > 0
