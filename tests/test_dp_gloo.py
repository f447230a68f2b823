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


# ---------------------------------------------------------------------------------------------------------------------
# Trainer level: arena + CommGroups, ZeRO-1 (reduce-scatter -> sharded per-model clip + Adam -> all-gather), replica
# broadcast, and DIRECT-mode buckets (the path PGRM / CMM use) including a module invoked twice per step (--sr_share).
# The two optimizer kernels are GPU-only in the product; here a torch restatement is injected so the exchange logic,
# the shard arithmetic and the per-model clip can be checked on CPU.

def _torch_sumsq(g, out, part):
    out.copy_((g.double() ** 2).sum().float().reshape(1))


def _torch_adam_clip(p, g, m, v, normsq, max_norm, lr, b1, b2, eps, step, step_dev):
    coef = 1.0
    if max_norm > 0:
        coef = min(1.0, max_norm / (float(normsq[0]) ** 0.5 + 1e-6))
    gi = g * coef
    m.mul_(b1).add_(gi, alpha=1 - b1)
    v.mul_(b2).addcmul_(gi, gi, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    p.sub_((lr / bc1) * m / (v.sqrt() / bc2 ** 0.5 + eps))


class _DirectLinear(torch.nn.Module):
    """y = x W^T with an explicit backward that accumulates straight into the bucket's gradient view and reports
    completion itself -- the protocol of train/pgrm_train.py (grad_targets / finish_grads)."""
    direct_grad = True

    def __init__(self, i, o):
        super().__init__()
        self.weight = torch.nn.Parameter(torch.randn(o, i) * 0.3)

    def forward(self, x):
        m = self

        class F(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, w):
                ctx.save_for_backward(x)
                return x @ w.t()

            @staticmethod
            def backward(ctx, dy):
                (x,) = ctx.saved_tensors
                dx = dy @ m.weight.detach()
                if getattr(m, "_dpmn_bucket", None) is None:      # plain autograd (the single-process reference)
                    return dx, dy.t() @ x
                m.weight._dpmn_sink += dy.t() @ x
                m._dpmn_bucket.grads_ready()
                return dx, None
        if getattr(self, "_dpmn_bucket", None) is not None:
            self._dpmn_bucket.note_use()
        return F.apply(x, self.weight)


class ComplementationModulationModule(torch.nn.Module):     # same class NAME as the big model: scheduled first, own group
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(5, 700)
        self.b = torch.nn.Linear(700, 3)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _build(seed):
    torch.manual_seed(seed)
    return [_DirectLinear(6, 5), _DirectLinear(5, 5), ComplementationModulationModule(), torch.nn.Linear(3, 2)]


def _loss(models, x):
    d0, d1, big, tail = models
    h = d1(d1(d0(x)))                   # d1 is used twice in one step (shared module)
    return tail(big(h)).pow(2).mean() * 50


def _trainer_worker(rank, world, port, q, zero1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dpmn_amd.train import optim
    optim._sumsq, optim._adam_clip = _torch_sumsq, _torch_adam_clip
    models = _build(1000 + rank)          # DIFFERENT initial weights per rank: the trainer must broadcast rank 0's
    tr = optim.Trainer(models, lr=1e-2, beta1=0.5, max_norm=0.25, world_size=world, zero1=zero1, group_mb=0.005)
    assert len(tr.groups) >= 2 and tr.groups[0].buckets[0].module is models[2]
    ref = _build(1000)                    # single-process reference = rank 0's weights, averaged gradients
    for a, b in zip(models, ref):
        for p, r in zip(a.parameters(), b.parameters()):
            assert torch.equal(p.detach(), r.detach()), "replicas must start from rank 0's parameters"
    opts = [torch.optim.Adam(m.parameters(), lr=1e-2, betas=(0.5, 0.999)) for m in ref]
    worst = 0.0
    for step in range(3):
        xs = [torch.randn(4, 6, generator=torch.Generator().manual_seed(10 * step + r)) for r in range(world)]
        tr.zero_grad()
        _loss(models, xs[rank]).backward()
        assert all(g.launched for g in tr.groups), "every group's exchange starts inside backward"
        tr.step()
        tr.sync_params()
        for m in ref:
            m.zero_grad()
        for r in range(world):
            (_loss(ref, xs[r]) / world).backward()          # mean of per-shard means
        for m, o in zip(ref, opts):
            torch.nn.utils.clip_grad_norm_(m.parameters(), 0.25)    # per model, as super_resolution.py:272-277
            o.step()
        for a, b in zip(models, ref):
            for p, r in zip(a.parameters(), b.parameters()):
                worst = max(worst, float((p.detach() - r.detach()).abs().max()))
    flat = tr.flat_p.clone()
    other = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    same = all(torch.equal(other[0], o) for o in other)
    q.put((rank, worst, same))
    dist.barrier()
    dist.destroy_process_group()


def _run_trainer(zero1):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q, zero1)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, same in res:
        assert same, "ranks diverged (rank %d)" % rank
        assert worst < 2e-5, (rank, worst)


def test_trainer_allreduce_direct_mode_world2_gloo():
    _run_trainer(zero1=False)


def test_trainer_zero1_sharded_adam_world2_gloo():
    _run_trainer(zero1=True)


# ---------------------------------------------------------------------------------------------------------------------
# ZeRO-1 + checkpoint: the parameter all-gather of a step is asynchronous; save_checkpoint(trainer=...) must wait for it, so
# a checkpoint written right after trainer.step() holds step-t values for EVERY shard, on every rank.

def _ckpt_worker(rank, world, port, q, tmp):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from types import SimpleNamespace
    from dpmn_amd.train import optim
    from dpmn_amd.interfaces.base import TextBase
    optim._sumsq, optim._adam_clip = _torch_sumsq, _torch_adam_clip
    models = _build(1000)
    tr = optim.Trainer(models, lr=1e-2, beta1=0.5, max_norm=0.25, world_size=world, zero1=True, group_mb=0.005)
    ref = _build(1000)
    opts = [torch.optim.Adam(m.parameters(), lr=1e-2, betas=(0.5, 0.999)) for m in ref]
    fake = SimpleNamespace(vis_dir=os.path.join(tmp, "rank%d" % rank), args=SimpleNamespace(arch="tsrn"), batch_size=4, scale_factor=2,
                           voc_type="all")
    worst = 0.0
    for step in range(4):
        xs = [torch.randn(4, 6, generator=torch.Generator().manual_seed(10 * step + r)) for r in range(world)]
        tr.zero_grad()
        _loss(models, xs[rank]).backward()
        tr.step()
        # NO explicit sync_params here: the save itself must wait for the all-gather
        d = TextBase.save_checkpoint(fake, models, 0, step + 1, {}, {}, False, [], None, trainer=tr)
        for m in ref:
            m.zero_grad()
        for r in range(world):
            (_loss(ref, xs[r]) / world).backward()
        for m, o in zip(ref, opts):
            torch.nn.utils.clip_grad_norm_(m.parameters(), 0.25)
            o.step()
        # checkpoint.pth is overwritten by every model (base.py:358): the file holds the LAST model of the list
        sd = torch.load(os.path.join(d, "checkpoint.pth"))["state_dict_G"]
        for k, v in ref[-1].state_dict().items():
            worst = max(worst, float((sd[k] - v).abs().max()))
    # a best-model save writes one file per model: compare all of them with the reference and across ranks
    d = TextBase.save_checkpoint(fake, models, 0, 4, {}, {}, True, [], None, trainer=tr)
    vec = []
    for i, m in enumerate(ref):
        sd = torch.load(os.path.join(d, "model_best_sum_0_%d.pth" % i))["state_dict_G"]
        for k, v in m.state_dict().items():
            worst = max(worst, float((sd[k] - v).abs().max()))
            vec.append(sd[k].reshape(-1))
    flat = torch.cat(vec)
    other = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    q.put((rank, worst, all(torch.equal(other[0], o) for o in other)))
    dist.barrier()
    dist.destroy_process_group()


def test_zero1_checkpoint_right_after_step_world2_gloo(tmp_path):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_ckpt_worker, args=(r, 2, port, q, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, same in res:
        assert same, "checkpoints differ between ranks (rank %d)" % rank
        assert worst < 2e-5, (rank, worst)


# ---------------------------------------------------------------------------------------------------------------------
# Segmented exchange (train/optim.py SegmentedBucket): a large direct-mode model is exchanged in K collectives, one per backward-
# ordered parameter segment, each launched the moment the module's backward reports the segment -- i.e. BEFORE the backward returns --
# while the clip stays per MODEL (one norm over all segments).

class _SegmentedMLP(torch.nn.Module):
    """three direct-mode Linear layers; the explicit backward writes each layer's gradient straight into the bucket's sink and
    reports its segment (the protocol of train/cmm_train.py backward / seg_done)"""
    direct_grad = True

    def __init__(self):
        super().__init__()
        self.l0 = torch.nn.Parameter(torch.randn(600, 6) * 0.2)
        self.l1 = torch.nn.Parameter(torch.randn(600, 600) * 0.05)
        self.l2 = torch.nn.Parameter(torch.randn(3, 600) * 0.2)
        self.launched_inside = []       # per backward call: was segment 0's collective already launched when segment 1 was computed?

    def exchange_segments(self):
        return [[self.l2], [self.l1], [self.l0]]          # backward order

    def forward(self, x):
        m = self

        class F(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x, anchor):
                h0 = torch.tanh(x @ m.l0.t())
                h1 = torch.tanh(h0 @ m.l1.t())
                ctx.save_for_backward(x, h0, h1)
                return h1 @ m.l2.t()

            @staticmethod
            def backward(ctx, dy):
                x, h0, h1 = ctx.saved_tensors
                b = getattr(m, "_dpmn_bucket", None)
                seg = b is not None and hasattr(b, "segment_ready")
                sink = (lambda p: p._dpmn_sink) if b is not None else None
                g2 = dy.t() @ h1
                d1 = (dy @ m.l2.detach()) * (1 - h1 * h1)
                if b is None:
                    g1 = d1.t() @ h0
                    d0 = (d1 @ m.l1.detach()) * (1 - h0 * h0)
                    m.l2.grad, m.l1.grad, m.l0.grad = g2, g1, d0.t() @ x
                    return None, None
                sink(m.l2).add_(g2)
                if seg:
                    b.segment_ready(b.seg_of[id(m.l2)])
                    m.launched_inside.append(b.segments[b.seg_of[id(m.l2)]].group.launched)
                sink(m.l1).add_(d1.t() @ h0)
                if seg:
                    b.segment_ready(b.seg_of[id(m.l1)])
                d0 = (d1 @ m.l1.detach()) * (1 - h0 * h0)
                sink(m.l0).add_(d0.t() @ x)
                if seg:
                    b.segment_ready(b.seg_of[id(m.l0)])
                else:
                    b.grads_ready()
                return None, None
        b = getattr(self, "_dpmn_bucket", None)
        if b is not None:
            b.note_use()
            return F.apply(x, b.anchor)
        return F.apply(x, torch.zeros(1, requires_grad=True))


def _seg_worker(rank, world, port, q, zero1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dpmn_amd.train import optim
    optim._sumsq, optim._adam_clip = _torch_sumsq, _torch_adam_clip
    torch.manual_seed(500 + rank)
    big, tail = _SegmentedMLP(), torch.nn.Linear(3, 2)
    tr = optim.Trainer([big, tail], lr=1e-2, beta1=0.5, max_norm=0.25, world_size=world, zero1=zero1, group_mb=0.005)
    assert isinstance(tr.buckets[0], optim.SegmentedBucket) and len(tr.buckets[0].segments) == 3
    assert len({id(s_.group) for s_ in tr.buckets[0].segments}) >= 2, "the segments must not share one collective"
    # the arena holds the big model in BACKWARD order: l2, l1, l0
    offs = [s_.flat_g.data_ptr() for s_ in tr.buckets[0].segments]
    assert offs == sorted(offs)
    torch.manual_seed(500)
    rbig, rtail = _SegmentedMLP(), torch.nn.Linear(3, 2)
    for a_, b_ in ((big, rbig), (tail, rtail)):
        for p, r in zip(a_.parameters(), b_.parameters()):
            assert torch.equal(p.detach(), r.detach()), "replicas must start from rank 0's parameters"
    opts = [torch.optim.Adam(m_.parameters(), lr=1e-2, betas=(0.5, 0.999)) for m_ in (rbig, rtail)]
    worst = 0.0
    for step in range(3):
        xs = [torch.randn(8, 6, generator=torch.Generator().manual_seed(20 * step + r)) for r in range(world)]
        tr.zero_grad()
        (tail(big(xs[rank])).pow(2).mean() * 50).backward()
        assert big.launched_inside[-1], "segment 0's collective must be issued while the backward is still computing segment 1"
        tr.step()
        tr.sync_params()
        for m_ in (rbig, rtail):
            for p in m_.parameters():
                p.grad = None
        for r in range(world):
            (rtail(rbig(xs[r])).pow(2).mean() * 50 / world).backward()
            if r == 0:
                acc = [p.grad.clone() for p in rbig.parameters()]
            else:
                for a_, p in zip(acc, rbig.parameters()):
                    a_ += p.grad            # (the explicit backward of the reference ASSIGNS .grad: accumulate by hand)
        for a_, p in zip(acc, rbig.parameters()):
            p.grad = a_
        for m_, o in zip((rbig, rtail), opts):
            torch.nn.utils.clip_grad_norm_(m_.parameters(), 0.25)     # per MODEL: one norm over the three segments
            o.step()
        for a_, b_ in ((big, rbig), (tail, rtail)):
            for p, r in zip(a_.parameters(), b_.parameters()):
                worst = max(worst, float((p.detach() - r.detach()).abs().max()))
    flat = tr.flat_p.clone()
    other = [torch.zeros_like(flat) for _ in range(world)]
    dist.all_gather(other, flat)
    q.put((rank, worst, all(torch.equal(other[0], o) for o in other)))
    dist.barrier()
    dist.destroy_process_group()


def _run_seg(zero1):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_seg_worker, args=(r, 2, port, q, zero1)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=60) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, worst, same in res:
        assert same, "ranks diverged (rank %d)" % rank
        assert worst < 2e-5, (rank, worst)


def test_segmented_exchange_starts_inside_backward_allreduce_world2_gloo():
    _run_seg(zero1=False)


def test_segmented_exchange_starts_inside_backward_zero1_world2_gloo():
    _run_seg(zero1=True)
