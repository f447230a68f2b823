"""World-size-2 gloo test (CPU) of the data-parallel gradient exchange: each model is one flat bucket that is
all-reduced (averaged) from a post-accumulate-grad hook as soon as its last gradient lands (dpmn_amd/train/optim.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dpmn_amd.train.optim import FlatBucket
    torch.manual_seed(0)
    a = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
    b = torch.nn.Linear(3, 2)
    buckets = [FlatBucket(a, "a"), FlatBucket(b, "b")]
    for bk in buckets:
        bk.install_hooks(world)
    # parameters stay views of the flat buffers
    assert all(p.data_ptr() >= buckets[0].flat_p.data_ptr() for p in a.parameters())
    torch.manual_seed(100 + rank)        # different shard per rank
    x = torch.randn(4, 6)
    for bk in buckets:
        bk.zero_grad()
    loss = b(a(x)).pow(2).mean()
    loss.backward()                       # hooks fire: b's bucket first (its backward runs first), then a's
    for bk in buckets:
        bk.wait()
    mine = torch.cat([p.grad.reshape(-1) for bk in buckets for p in bk.params]).clone()   # views into the (padded) flat buffers
    # reference: local gradients recomputed without hooks, averaged with an explicit all_reduce
    a2 = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Linear(5, 3))
    b2 = torch.nn.Linear(3, 2)
    a2.load_state_dict(a.state_dict())
    b2.load_state_dict(b.state_dict())
    b2(a2(x)).pow(2).mean().backward()
    ref = torch.cat([p.grad.reshape(-1) for p in list(a2.parameters()) + list(b2.parameters())])
    dist.all_reduce(ref)
    ref /= world
    q.put((rank, float((mine - ref).abs().max()), float(ref.abs().max())))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, err, scale in res:
        assert scale > 0 and err < 1e-6 * max(1.0, scale), (rank, err, scale)
