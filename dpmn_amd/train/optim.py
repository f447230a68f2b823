"""Optimizer + data-parallel gradient exchange of the DPMN training step (interfaces/super_resolution.py:272-278,
base.py:200-223; DataParallel's reduce replaced per SURVEY.md section 5.8).

Each model (PGRM_k, DistillModule_j, CMM) becomes ONE flat bucket: parameters and gradients are views into contiguous
buffers, so
  * clip_grad_norm_(model.parameters(), 0.25) is one sum-of-squares kernel,
  * Adam(lr, betas=(beta1, 0.999)) is one fused kernel (clip coefficient applied on the fly, no host sync),
  * the RCCL all-reduce is one collective per model, launched as soon as the model's last gradient lands -- CMM's
    214 MB bucket goes first (its backward runs first) and overlaps with the PGRM backward.
PGRM and CMM (explicit backward kernels, train/pgrm_train.py, train/cmm_train.py) run in DIRECT mode: their backward
accumulates straight into the bucket's gradient views and reports completion itself, so autograd never allocates,
zero-fills or adds a per-parameter gradient tensor (that was ~1500 tiny launches per step).  Every other module keeps the
ordinary post-accumulate-grad hooks.
Gradient averaging over ranks = the single-device semantics of the reference (mean of per-shard means).
"""
import torch
import torch.distributed as dist

from .._abi import lib, check, dptr, stream


class FlatBucket:
    def __init__(self, module, name="", direct=False):
        self.name = name
        self.direct = direct
        self.params = [p for p in module.parameters()]
        align = 64      # floats: every parameter / gradient view starts on a 256-byte boundary (float4 + atomics friendly)
        n = sum((p.numel() + align - 1) // align * align for p in self.params)
        dev = self.params[0].device
        self.flat_p = torch.zeros(n, device=dev)
        self.flat_g = torch.zeros(n, device=dev)
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view(p.shape)
            p.grad = self.flat_g[off:off + k].view(p.shape)
            off += (k + align - 1) // align * align
        if direct:
            for p in self.params:
                p._dpmn_sink = p.grad
            module._dpmn_bucket = self
        self.n = n
        self.m = torch.zeros(n, device=dev)
        self.v = torch.zeros(n, device=dev)
        self.normsq = torch.zeros(1, device=dev)
        self.part = torch.empty(1024, device=dev)
        self.work = None
        self._pending = 0

    # ------------------------------------------------------------------ DP exchange
    def install_hooks(self, world_size, group=None):
        """all-reduce this bucket as soon as every parameter has accumulated its gradient in the current backward."""
        self.world, self.group = world_size, group
        if self.direct:
            return          # the module's backward calls grads_ready() itself
        trainable = [p for p in self.params if p.requires_grad]
        self._expect = len(trainable)

        def hook(_p):
            self._pending += 1
            if self._pending == self._expect:
                self._pending = 0
                self.launch_allreduce()
        for p in trainable:
            p.register_post_accumulate_grad_hook(hook)

    def grads_ready(self):
        """direct mode: called by the module's backward once every gradient of this bucket has been accumulated."""
        self.launch_allreduce()

    def launch_allreduce(self):
        if getattr(self, "world", 1) > 1:
            self.flat_g.div_(self.world)    # pre-scale: average (mean of per-shard means)
            self.work = dist.all_reduce(self.flat_g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def wait(self):
        if self.work is not None:
            self.work.wait()
            self.work = None

    # ------------------------------------------------------------------ optimizer
    def zero_grad(self):
        self.flat_g.zero_()

    def step(self, step, lr, beta1, beta2=0.999, eps=1e-8, max_norm=0.25, step_dev=None):
        if self.flat_g.is_cuda:
            check(lib.dpmn_sumsq_f32(dptr(self.flat_g), dptr(self.normsq), dptr(self.part), self.n, stream()))
            check(lib.dpmn_adam_clip_f32(dptr(self.flat_p), dptr(self.flat_g), dptr(self.m), dptr(self.v), dptr(self.normsq),
                                         max_norm, lr, beta1, beta2, eps, step, dptr(step_dev, True), self.n, stream()))
        else:
            raise RuntimeError("dpmn_amd: the optimizer kernels run on the GPU only")


class Trainer:
    """zero_grad / backward hooks / clip+Adam over a list of models, in the reference's order."""

    def __init__(self, models, lr=1e-3, beta1=0.5, max_norm=0.25, world_size=1, group=None):
        self.buckets = [FlatBucket(m, "model%d" % i, direct=getattr(m, "direct_grad", False)) for i, m in enumerate(models)]
        self.lr, self.beta1, self.max_norm = lr, beta1, max_norm
        self.t = 0
        self.t_dev = None       # device-side step counter, see device_step_counter()
        for b in self.buckets:
            b.install_hooks(world_size, group)

    def zero_grad(self):
        for b in self.buckets:
            b.zero_grad()

    def device_step_counter(self):
        """Keep Adam's step count in device memory from now on, so that a hipGraph capture of the training step replays
        the right bias corrections (host scalars are frozen into a graph at capture time)."""
        if self.t_dev is None:
            self.t_dev = torch.full((1,), float(self.t), device=self.buckets[0].flat_p.device)

    def step(self):
        self.t += 1
        if self.t_dev is not None:
            self.t_dev.add_(1.0)
        for b in self.buckets:
            b.wait()
            b.step(self.t, self.lr, self.beta1, max_norm=self.max_norm, step_dev=self.t_dev)
