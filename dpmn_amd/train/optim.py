"""Optimizer + data-parallel gradient exchange of the DPMN training step (interfaces/super_resolution.py:272-278,
base.py:200-223; nn.DataParallel's broadcast/reduce replaced per SURVEY.md section 5.8 and section 8(f)-2).

Storage: ONE parameter arena and ONE gradient arena for the whole trainer.  Each model (PGRM_k, DistillModule_j, CMM) is a
`FlatBucket` = a contiguous range of both arenas (every parameter view 256-byte aligned), so
  * zero_grad is one memset,
  * clip_grad_norm_(model.parameters(), 0.25) is one sum-of-squares kernel per model (the clip stays PER MODEL, as in the
    reference) and Adam(lr, betas=(beta1, 0.999)) one fused kernel (clip coefficient applied on the fly, no host sync).
Buckets are packed into `CommGroup`s, the unit of the RCCL exchange: consecutive models (in the trainer's order, CMM first
because its backward runs first) are coalesced until a group holds >= `group_mb` of gradients -- CMM's 214 MB is its own
group, the six 2.3 MB PGRMs and the 1 KB DistillModules share ONE (group_mb = 16: measured with the collectives forced at world
size 1, RCCL + ZeRO-1: 11 groups of >= 6 MB 30.2 ms, 2 groups 28.3 ms, plain step 27.0 -- every collective costs ~0.2 ms of launch /
stream hand-over whatever its size, and the PGRMs' 14 MB are exchanged in ~0.1 ms over xGMI even when issued after the last backward).
A group's collective is launched the moment the LAST gradient of its LAST member has been accumulated in the current
backward (async, on RCCL's stream, overlapping the rest of the backward) and is waited for right before its optimizer
kernels.  Two exchange modes:
  * zero1=False: all-reduce(AVG) of the group's gradient range; every rank runs the full clip+Adam.
  * zero1=True (default for world > 1; ZeRO-1): reduce-scatter(AVG) -> every rank owns 1/world of the group's range
    -> per-model ||g||^2 of the owned slices, one tiny all-reduce(SUM) of those scalars (the per-model clip needs the norm
    of the WHOLE model gradient) -> clip+Adam on the owned slices only (exp_avg / exp_avg_sq exist only for the owned
    1/world) -> async all-gather of the updated parameters, waited for by the first forward that uses a member model
    (CMM's 214 MB all-gather hides behind the next step's PSN + PGRM forward).
Gradient averaging over ranks = the single-device semantics of the reference (mean of per-shard means).
Replicas start identical: rank 0's parameters and buffers (BatchNorm running statistics) are broadcast at construction,
which is what nn.DataParallel's per-forward replicate() guarantees in the reference (base.py:160-162).

PGRM and CMM (explicit backward kernels, train/pgrm_train.py, train/cmm_train.py) run in DIRECT mode: their backward
accumulates straight into the bucket's gradient views and reports completion itself, so autograd never allocates,
zero-fills or adds a per-parameter gradient tensor.  A module invoked several times per step (--sr_share) reports once per
call; the bucket is ready when as many backward calls have finished as forward calls were made.  Every other module
keeps ordinary post-accumulate-grad hooks.
"""
import os
import torch
import torch.distributed as dist

from .._abi import lib, check, dptr, stream

ALIGN = 64          # floats: every parameter / gradient view starts on a 256-byte boundary (float4 + atomics friendly)


# 1: clip + Adam of a model group on a side stream as soon as its backward has reported.  Measured (B = 48, two runs each): 28.30 / 28.26 ms
# against 28.24 / 28.08 ms with every optimizer launch at the end of the step -- the optimizer leaves the tail (0.44 -> 0.01 ms) but
# its HBM-bound kernels slow the concurrent backward by as much: the GPU is full.  Kept as a switch, default off.
EARLY_OPT = os.environ.get("DPMN_EARLY_OPT", "0") != "0"


def _padded(n, a=ALIGN):
    return (n + a - 1) // a * a


def _sumsq(g, out, part):
    """out[0] = sum(g^2) on the GPU (csrc/conv_bwd.hip k_sumsq_partial).  Module-level so the CPU gloo tests can
    substitute a torch restatement for the two kernels below -- the product path has no CPU implementation."""
    if not g.is_cuda:
        raise RuntimeError("dpmn_amd: the optimizer kernels run on the GPU only")
    check(lib.dpmn_sumsq_f32(dptr(g), dptr(out), dptr(part), g.numel(), stream()))


def _adam_clip(p, g, m, v, normsq, max_norm, lr, beta1, beta2, eps, step, step_dev):
    if not p.is_cuda:
        raise RuntimeError("dpmn_amd: the optimizer kernels run on the GPU only")
    check(lib.dpmn_adam_clip_f32(dptr(p), dptr(g), dptr(m), dptr(v), dptr(normsq), max_norm, lr, beta1, beta2, eps, step,
                                 dptr(step_dev, True), p.numel(), stream()))


class FlatBucket:
    """One model's parameters / gradients as one contiguous range.  Stand-alone (`FlatBucket(module)`) it owns its storage;
    inside a Trainer the storage is a slice of the trainer's arenas (`storage=(flat_p, flat_g)`)."""

    def __init__(self, module, name="", direct=False, storage=None, params=None):
        """params: a subset of the module's parameters (one exchange segment of a large model, see SegmentedBucket); default: all."""
        self.name = name
        self.direct = direct
        self.module = module
        self.params = [p for p in module.parameters()] if params is None else list(params)
        n = self.size_of_params(self.params)
        dev = self.params[0].device
        if storage is None:
            self.flat_p = torch.zeros(n, device=dev)
            self.flat_g = torch.zeros(n, device=dev)
        else:
            self.flat_p, self.flat_g = storage
            assert self.flat_p.numel() == n and self.flat_g.numel() == n
        off = 0
        for p in self.params:
            k = p.numel()
            self.flat_p[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + k].view(p.shape)
            p.grad = self.flat_g[off:off + k].view(p.shape)
            off += _padded(k)
        if direct:
            for p in self.params:
                p._dpmn_sink = p.grad
            module._dpmn_bucket = self
            self.anchor = torch.zeros(1, device=dev, requires_grad=True)      # train/pgrm_train.py fn_inputs
        # caches keyed by the OLD gradient sinks / parameter list (train/pgrm_train.py UnpackQueue, params_of): a second Trainer or
        # bucket for the same module must not inherit them -- stale sink keys would keep the one-launch unpack from ever matching
        # again and hold the old slot workspaces alive
        module.__dict__.pop("_unpack_queue", None)
        module.__dict__.pop("_dpmn_plist", None)
        self.n = n
        self.normsq = torch.zeros(1, device=dev)
        self.part = torch.empty(1024, device=dev)
        self.m = self.v = None          # allocated on first stand-alone step(); a CommGroup owns them otherwise
        self.group = None
        self.uses = 0                   # direct mode: forward calls of the module in this step ...
        self.done = 0                   # ... and backward calls that have finished
        self._pending = 0
        self._expect = 0
        self.ready = False

    @staticmethod
    def size_of(module):
        return sum(_padded(p.numel()) for p in module.parameters())

    @staticmethod
    def size_of_params(params):
        return sum(_padded(p.numel()) for p in params)

    def param_slices(self):
        """(parameter, offset in this bucket's flat range, numel) -- every view starts on an ALIGN boundary"""
        off = 0
        for p in self.params:
            yield p, off, p.numel()
            off += _padded(p.numel())

    # ------------------------------------------------------------------ readiness (when may the exchange start?)
    def install_hooks(self, world_size=1, group=None):
        """Stand-alone use (tests, single bucket): wraps the bucket in its own CommGroup."""
        if self.group is None:
            CommGroup([self], world_size, group, zero1=False)
        return self

    def _install_ready_hooks(self):
        if self.direct:
            return          # the module's forward / backward call note_use() / grads_ready() themselves
        trainable = [p for p in self.params if p.requires_grad]
        self._expect = len(trainable)

        def hook(_p):
            self._pending += 1
            if self._pending == self._expect:
                self._pending = 0
                self._mark_ready()
        for p in trainable:
            p.register_post_accumulate_grad_hook(hook)

    def note_use(self):
        """direct mode: the module's training forward ran once more in this step."""
        self.uses += 1

    def grads_ready(self):
        """direct mode: one backward call of the module has accumulated all its gradients."""
        self.done += 1
        if self.done >= max(self.uses, 1):
            self._mark_ready()

    def _mark_ready(self, done=None):
        self.ready = True
        # the backward of the two refinement branches runs on two HIP streams (interfaces/super_resolution.py): the group's
        # collective is enqueued behind the stream of its LAST reporter only, so every member leaves an event for it to wait on
        if done is None:
            done = self.module.__dict__.pop("_dpmn_grads_done", None)
        elif done is False:       # (a segment of a SegmentedBucket reported without a side-stream event: the module-level event is not its own)
            done = None
        if done is not None:
            # the module's backward left part of its gradients on a side stream (train/cmm_train.py: the conv weight gradients and their
            # unpack) and did NOT make the calling stream wait for it: whoever consumes the bucket waits for this event instead
            self.ready_event = done
        elif self.group is not None and (self.group.multi or self.group.early_cb is not None) and self.flat_g.is_cuda:
            self.ready_event = torch.cuda.Event()
            self.ready_event.record()
        if self.group is not None:
            self.group.member_ready()

    def reset_step(self):
        self.uses = self.done = self._pending = 0
        self.ready = False

    # ------------------------------------------------------------------ stand-alone optimizer API
    def wait(self):
        if self.group is not None:
            self.group.wait_grads()
        self._wait_ready_event()

    def _wait_ready_event(self):
        """Order the current stream behind gradients the module's backward left on a side stream (_mark_ready)."""
        ev = getattr(self, "ready_event", None)
        if ev is not None and self.flat_g.is_cuda:
            torch.cuda.current_stream(self.flat_g.device).wait_event(ev)
            self.ready_event = None

    def zero_grad(self):
        # a backward that was never followed by step() (an exception, a skipped step) may have left leaf gradients in flight on a
        # side stream behind ready_event: the zero fill must not overtake them, and the promise of arm_early_step() ends here
        self._wait_ready_event()
        self.lazy_join = False
        self.flat_g.zero_()
        self.reset_step()
        if self.group is not None:
            self.group.launched = False

    def step(self, step, lr, beta1, beta2=0.999, eps=1e-8, max_norm=0.25, step_dev=None):
        """clip + Adam over the whole bucket (single-process semantics)."""
        if self.m is None:
            self.m = torch.zeros_like(self.flat_p)
            self.v = torch.zeros_like(self.flat_p)
        self._wait_ready_event()
        _sumsq(self.flat_g, self.normsq, self.part)
        _adam_clip(self.flat_p, self.flat_g, self.m, self.v, self.normsq, max_norm, lr, beta1, beta2, eps, step, step_dev)


class SegmentedBucket:
    """ONE direct-mode model whose gradients are exchanged in K collectives instead of one (the CMM: 214 MB, SURVEY.md 5.8 "launched
    as soon as a bucket is ready, overlapped with the remaining backward").  The model's parameters are laid out in the arenas in
    BACKWARD order and cut into K segments (the module's `exchange_segments()`: decoder first, then the bottleneck, then the encoder
    levels deepest first); every segment is a FlatBucket in a CommGroup of its own, and the module's backward reports
    `segment_ready(k, event)` the moment the last gradient of segment k is in place -- its reduce-scatter / all-reduce then runs
    under the backward of the segments still to come.  The clip stays per MODEL: the Trainer sums the segments' ||g||^2 slots into
    one value before any segment's Adam kernel reads it (one norm all-reduce for everything, as before).  Towards the module and the
    Trainer this object looks like the model's one bucket (note_use / grads_ready / lazy_join / reset_step ...)."""

    def __init__(self, module, segments, name=""):
        self.module, self.segments, self.name = module, segments, name
        self.direct = True
        self.anchor = segments[0].anchor
        self.params = [p for s_ in segments for p in s_.params]
        self.n = sum(s_.n for s_ in segments)
        self.uses = 0
        self.seg_done = [0] * len(segments)
        self.lazy_join = False
        self.seg_of = {id(p): k for k, s_ in enumerate(segments) for p in s_.params}
        module._dpmn_bucket = self

    flat_g = property(lambda self: self.segments[0].flat_g)
    ready = property(lambda self: all(s_.ready for s_ in self.segments))

    def note_use(self):
        self.uses += 1

    def segment_ready(self, k, done=None):
        """one backward call of the module has every gradient of segment k in place (done: an event that covers the streams they
        were written on, or None = complete in the calling stream's order)"""
        self.seg_done[k] += 1
        if self.seg_done[k] >= max(self.uses, 1) and not self.segments[k].ready:
            self.segments[k]._mark_ready(done if done is not None else False)

    def grads_ready(self):
        """whole-module completion: every segment the backward did not report by itself"""
        done = self.module.__dict__.pop("_dpmn_grads_done", None)
        for k, s_ in enumerate(self.segments):
            if not s_.ready:
                self.segment_ready(k, done)

    def reset_step(self):
        self.uses = 0
        self.seg_done = [0] * len(self.segments)
        for s_ in self.segments:
            s_.reset_step()

    def _wait_ready_event(self):
        for s_ in self.segments:
            s_._wait_ready_event()


# 1: gradient collectives issued from a stream of their own that waits for the members' ready events, instead of the stream of the last
# reporting member.  Measured with the collectives forced at world size 1 (RCCL, ZeRO-1): 30.10 / 30.07 vs 29.90 / 29.95 ms -- off.
COMM_STREAM = os.environ.get("DPMN_COMM_STREAM", "0") != "0"
_COMM = {}


def _comm_stream(dev):
    key = (dev.type, dev.index)
    if key not in _COMM:
        _COMM[key] = torch.cuda.Stream(device=dev)
    return _COMM[key]


class CommGroup:
    """Several buckets adjacent in the arenas, exchanged with one collective."""

    def __init__(self, buckets, world=1, pg=None, zero1=False, arena=None, force=False):
        self.buckets = buckets
        self.multi = world > 1 or force          # force: issue the collectives even at world size 1
        self.world, self.pg, self.zero1 = world, pg, bool(zero1) and self.multi
        self.rank = dist.get_rank(pg) if self.multi else 0
        if arena is None:      # stand-alone bucket: its own storage is the group's range (no padding to `world` needed)
            assert len(buckets) == 1 and not self.zero1
            self.flat_p, self.flat_g = buckets[0].flat_p, buckets[0].flat_g
        else:
            self.flat_p, self.flat_g = arena
        self.n = self.flat_g.numel()
        self.offsets = []
        off = 0
        for b in buckets:
            self.offsets.append(off)
            off += b.n
            b.group = self
            b._install_ready_hooks()
        dev = self.flat_p.device
        if self.zero1:
            assert self.n % world == 0
            self.shard_n = self.n // world
            self.lo = self.rank * self.shard_n
            self.g_shard = torch.zeros(self.shard_n, device=dev)      # reduce-scatter output (averaged gradients I own)
        else:
            self.shard_n, self.lo = self.n, 0
            self.g_shard = self.flat_g
        self.m = torch.zeros(self.shard_n, device=dev)
        self.v = torch.zeros(self.shard_n, device=dev)
        self.normsq = torch.zeros(len(buckets), device=dev)
        self.part = torch.empty(1024, device=dev)
        self.launched = False
        self.early_cb = None         # Trainer: called when the last member reports (single-process early optimizer step)
        self.stepped = False
        self.grad_work = None
        self.param_work = None
        self._post_scale = None
        if self.zero1:
            for b in buckets:      # the all-gather of step t is awaited by the first forward of step t+1 that needs it
                b.module.register_forward_pre_hook(lambda _m, _a: self.wait_params())

    # ------------------------------------------------------------------ gradient exchange
    def member_ready(self):
        if not self.launched and all(b.ready for b in self.buckets):
            self.launch()
            if self.early_cb is not None:
                self.early_cb(self)

    def launch(self):
        self.launched = True
        if not self.multi:
            return
        comm = None
        if self.flat_g.is_cuda:
            # the collective is issued from a stream of its own that waits for every member's ready event: the stream the last member's
            # backward runs on goes straight on to the next module's backward instead of waiting here for the other members' streams
            # (and, for a module with side-stream gradients, for that side stream: the CMM's weight-gradient unpack)
            comm = _comm_stream(self.flat_g.device) if COMM_STREAM else torch.cuda.current_stream()
            for b in self.buckets:
                ev = getattr(b, "ready_event", None)
                if ev is not None:
                    comm.wait_event(ev)
                    b.ready_event = None
        if comm is not None and COMM_STREAM:
            with torch.cuda.stream(comm):
                self._issue()
        else:
            self._issue()

    def _issue(self):
        avg = dist.ReduceOp.AVG
        self._post_scale = None
        if dist.get_backend(self.pg) != "nccl":        # test hook (gloo): no AVG / reduce-scatter guarantee on every build
            avg = dist.ReduceOp.SUM
            self._post_scale = 1.0 / self.world
        if self.zero1:
            try:
                self.grad_work = dist.reduce_scatter_tensor(self.g_shard, self.flat_g, op=avg, group=self.pg, async_op=True)
            except (RuntimeError, NotImplementedError):
                if dist.get_backend(self.pg) == "nccl":
                    raise
                self.grad_work = dist.all_reduce(self.flat_g, op=avg, group=self.pg, async_op=True)
                self._copy_shard = True
        else:
            self.grad_work = dist.all_reduce(self.flat_g, op=avg, group=self.pg, async_op=True)

    def wait_grads(self):
        if not self.launched:
            self.launch()       # a member never reported (e.g. a parameter without gradient): exchange now, still correct
        if self.grad_work is not None:
            self.grad_work.wait()
            self.grad_work = None
            if getattr(self, "_copy_shard", False):
                self.g_shard.copy_(self.flat_g[self.lo:self.lo + self.shard_n])
                self._copy_shard = False
            if self._post_scale is not None:
                self.g_shard.mul_(self._post_scale)

    def wait_params(self):
        if self.param_work is not None:
            self.param_work.wait()
            self.param_work = None

    # ------------------------------------------------------------------ optimizer
    def _owned(self, i):
        """intersection of bucket i with the range this rank owns, as (offset in the shard, length)."""
        a = max(self.offsets[i], self.lo)
        b = min(self.offsets[i] + self.buckets[i].n, self.lo + self.shard_n)
        return (a - self.lo, b - a) if b > a else (0, 0)

    def step(self, step, lr, beta1, beta2=0.999, eps=1e-8, max_norm=0.25, step_dev=None):
        self.step_norms()
        if self.zero1:
            dist.all_reduce(self.normsq, op=dist.ReduceOp.SUM, group=self.pg)     # len(buckets) floats
        self.step_update(step, lr, beta1, beta2, eps, max_norm, step_dev)

    def step_norms(self, zero=True):
        """First half of step(): wait for the exchanged gradients, ||g||^2 of the part of every member this rank owns into self.normsq
        (ZeRO-1: a partial sum, to be all-reduced -- by step(), or by the Trainer for ALL groups in one collective)."""
        self.wait_grads()
        self.wait_params()
        if self.flat_g.is_cuda:      # single process: events that launch() (multi only) did not consume -- a member's side-stream gradients
            for b in self.buckets:
                ev = getattr(b, "ready_event", None)
                if ev is not None:
                    torch.cuda.current_stream(self.flat_g.device).wait_event(ev)
                    b.ready_event = None
        owned = [self._owned(i) for i in range(len(self.buckets))]
        if self.zero1 and zero:
            self.normsq.zero_()
        for i, (o, k) in enumerate(owned):
            if k:
                _sumsq(self.g_shard[o:o + k], self.normsq[i:i + 1], self.part)

    def step_update(self, step, lr, beta1, beta2=0.999, eps=1e-8, max_norm=0.25, step_dev=None):
        """Second half of step(): clip + Adam on the owned shard with the (all-reduced) norms, then the parameter all-gather."""
        owned = [self._owned(i) for i in range(len(self.buckets))]
        p_shard = self.flat_p[self.lo:self.lo + self.shard_n]
        for i, (o, k) in enumerate(owned):
            if k:
                _adam_clip(p_shard[o:o + k], self.g_shard[o:o + k], self.m[o:o + k], self.v[o:o + k], self.normsq[i:i + 1],
                           max_norm, lr, beta1, beta2, eps, step, step_dev)
        if self.zero1:
            self.param_work = dist.all_gather_into_tensor(self.flat_p, p_shard, group=self.pg, async_op=True)


def broadcast_replicas(models, flat_p, pg=None):
    """All ranks start from rank 0's parameters and buffers (the reference replicates one model, base.py:160-162)."""
    dist.broadcast(flat_p, src=0, group=pg)
    for m in models:
        for buf in m.buffers():
            if buf.numel():
                dist.broadcast(buf, src=0, group=pg)


class Trainer:
    """zero_grad / gradient exchange / clip+Adam over a list of models, in the reference's order."""

    def __init__(self, models, lr=1e-3, beta1=0.5, max_norm=0.25, world_size=1, group=None, zero1=None, group_mb=16.0,
                 force_collectives=False):
        self.lr, self.beta1, self.max_norm = lr, beta1, max_norm
        self.world = world_size
        multi = world_size > 1 or force_collectives      # force_collectives: run the RCCL calls at world size 1 (1-GPU box check)
        group_mb = float(os.environ.get("DPMN_GROUP_MB", group_mb))      # experiment knob: size at which an exchange group is closed
        zero1 = multi if zero1 is None else (bool(zero1) and multi)
        self.zero1 = zero1
        self.t = 0
        self.t_dev = None       # device-side step counter, see device_step_counter()
        dev = next(models[0].parameters()).device
        # exchange order = expected backward order: the model list arrives as [PGRM_0.., CMM, Distill..] and the backward
        # runs CMM first, then the distill modules, then the PGRMs last-to-first
        # exchange UNITS: a model, or -- for a direct-mode model that offers `exchange_segments()` and holds >= 4 groups' worth of
        # gradients (the CMM) when collectives run at all -- its backward-ordered segments (SegmentedBucket)
        seg_on = os.environ.get("DPMN_SEGMENT_EXCHANGE", "1") != "0"
        units = []       # (model index, segment index or None, parameter list or None, padded size)
        for i in self._backward_order(models):
            m = models[i]
            segs = None
            if multi and seg_on and getattr(m, "direct_grad", False) and hasattr(m, "exchange_segments") and FlatBucket.size_of(m) * 4 >= 4 * group_mb * 2 ** 20:
                segs = m.exchange_segments()
                assert sorted(id(p) for sg in segs for p in sg) == sorted(id(p) for p in m.parameters()), "exchange_segments must cover every parameter once"
            if segs and len(segs) > 1:
                for k, sg in enumerate(segs):
                    units.append((i, k, sg, FlatBucket.size_of_params(sg)))
            else:
                units.append((i, None, None, FlatBucket.size_of(m)))
        plan, cur, cur_n = [], [], 0
        for u in range(len(units)):
            cur.append(u)
            cur_n += units[u][3]
            if cur_n * 4 >= group_mb * 2 ** 20:
                plan.append(cur)
                cur, cur_n = [], 0
        if cur:
            if plan and sum(units[u][3] for u in cur) * 4 < 2 ** 20 and len(plan) > 1:
                plan[-1].extend(cur)        # a sub-megabyte tail joins the previous small-model group
            else:
                plan.append(cur)
        unit = ALIGN * max(world_size, 1)
        gsize = [_padded(sum(units[u][3] for u in g), unit) for g in plan]
        total = sum(gsize)
        self.flat_p = torch.zeros(total, device=dev)
        self.flat_g = torch.zeros(total, device=dev)
        from ..model.packing import PackCache
        self.pack_cache = PackCache(self.flat_p)
        if zero1:
            self.pack_cache.before_refresh = self.sync_params
        self.buckets = [None] * len(models)
        self.groups = []
        seg_parts = {}
        off = 0
        for g, gs in zip(plan, gsize):
            o = off
            members = []
            for u in g:
                i, k, plist, n_u = units[u]
                st = (self.flat_p[o:o + n_u], self.flat_g[o:o + n_u])
                b = FlatBucket(models[i], "model%d" % i + ("" if k is None else ".s%d" % k), direct=getattr(models[i], "direct_grad", False), storage=st,
                               params=plist)
                if k is None:
                    self.buckets[i] = b
                else:
                    seg_parts.setdefault(i, []).append(b)
                members.append(b)
                o += n_u
            self.groups.append(CommGroup(members, world_size, group, zero1, arena=(self.flat_p[off:off + gs], self.flat_g[off:off + gs]),
                                         force=force_collectives))
            off += gs
        for i, parts in seg_parts.items():
            self.buckets[i] = SegmentedBucket(models[i], parts, "model%d" % i)
        # the clip norms of every group as slices of ONE vector: under ZeRO-1 the partial sums of all groups are all-reduced by one
        # collective in Trainer.step() (a tiny all-reduce per group between its norm and its Adam kernel was a stream round trip each:
        # 11 of them, ~1 ms of the step's serial tail)
        nb = sum(len(g.buckets) for g in self.groups)
        self.normsq_all = torch.zeros(nb, device=dev)
        o = 0
        slot = {}
        for g in self.groups:
            g.normsq = self.normsq_all[o:o + len(g.buckets)]
            for j, b in enumerate(g.buckets):
                slot[id(b)] = o + j
            o += len(g.buckets)
        # a segmented model's clip needs the norm of the WHOLE model: the slots of its segments, summed in step() before any Adam kernel
        self.seg_slots = [torch.tensor([slot[id(b)] for b in self.buckets[i].segments], device=dev) for i in sorted(seg_parts)]
        if multi:
            broadcast_replicas(models, self.flat_p, group)
        # single process: clip + Adam of a group run on a stream of their own as soon as its last member's backward has reported,
        # under the backward of the models that are still to come (the clip is per model, super_resolution.py:272-278: nothing of
        # another model is needed); Trainer.step() then only handles what is left and joins the stream
        self.early = EARLY_OPT and not multi and dev.type == "cuda"
        self.opt_stream = None
        if self.early:
            for g in self.groups:
                g.early_cb = self._early_step

    @staticmethod
    def _backward_order(models):
        idx = list(range(len(models)))
        big = [i for i in idx if type(models[i]).__name__ == "ComplementationModulationModule"]
        small = [i for i in idx if type(models[i]).__name__ == "DistillModule"]
        rest = [i for i in idx if i not in big and i not in small]
        return big + small + rest[::-1]

    def leaf_buckets(self):
        """every FlatBucket, the segments of a SegmentedBucket included (tests, diagnostics)"""
        out = []
        for b in self.buckets:
            out.extend(b.segments if isinstance(b, SegmentedBucket) else [b])
        return out

    def zero_grad(self):
        from ..model import packing
        self.disarm()       # a previous backward whose step() never came: join its side-stream gradients before the zero fill
        self.flat_g.zero_()
        for b in self.buckets:
            b.reset_step()
        for g in self.groups:
            g.launched = False
        # weight packs / transposes of this step: one multi-descriptor launch at their first use (model/packing.py PackCache)
        self.pack_cache.new_step()
        packing.ACTIVE = self.pack_cache

    def device_step_counter(self):
        """Keep Adam's step count in device memory from now on, so that a hipGraph capture of the training step replays
        the right bias corrections (host scalars are frozen into a graph at capture time)."""
        if self.t_dev is None:
            self.t_dev = torch.full((1,), float(self.t), device=self.flat_p.device)

    def sync_params(self):
        """Wait for outstanding parameter all-gathers (ZeRO-1); call before reading parameters outside a forward
        (checkpoints, state_dict clones) -- the gather runs on RCCL's stream and writes flat_p behind torch's back."""
        for g in self.groups:
            g.wait_params()

    def invalidate_packs(self):
        """The optimizer kernels write parameters through raw pointers, so torch's `_version` counters never move: every
        module that caches eval-mode weight packs keyed on them (model/cmm.py `_pack`) must drop the cache explicitly."""
        for b in self.buckets:
            if getattr(b.module, "_pack", None) is not None:      # CMM: eval packs; PGRM: "the workspace holds the folded weights"
                b.module._pack = None

    def state_snapshot(self):
        """Everything a step mutates in the optimizer (used to undo hipGraph warm-up steps)."""
        return dict(p=self.flat_p.clone(), t=self.t, t_dev=None if self.t_dev is None else self.t_dev.clone(),
                    mv=[(g.m.clone(), g.v.clone()) for g in self.groups])

    def state_restore(self, snap):
        self.flat_p.copy_(snap["p"])
        self.t = snap["t"]
        if self.t_dev is not None and snap["t_dev"] is not None:
            self.t_dev.copy_(snap["t_dev"])
        for g, (m, v) in zip(self.groups, snap["mv"]):
            g.m.copy_(m)
            g.v.copy_(v)
        self.invalidate_packs()

    def arm_early_step(self):
        """The caller promises that Trainer.step() follows the backward pass it is about to start (TextSR.train_step): groups may
        then be stepped as soon as their gradients are complete (DPMN_EARLY_OPT), and a module's backward may return while its leaf
        gradients are still being written on a side stream (the bucket's ready_event covers them; Trainer.step / the gradient
        exchange wait for it).  Without this call a backward never touches the parameters and every gradient is complete in the
        order of the stream the backward ran on."""
        self._armed = self.early
        for b in self.buckets:
            b.lazy_join = True      # a backward may leave leaf gradients on a side stream behind bucket.ready_event (train/cmm_train.py)

    def disarm(self):
        """End the promise of arm_early_step() without a step (an exception after the backward, a caller that wants to read the
        gradients): the current stream waits for every gradient a backward left on a side stream, and later backwards join their
        side streams themselves again.  Trainer.step() and zero_grad() call it; it is idempotent."""
        self._armed = False
        for b in self.buckets:
            b._wait_ready_event()
            b.lazy_join = False

    def _early_step(self, g):
        if not getattr(self, "_armed", False) or self.t_dev is not None or torch.cuda.is_current_stream_capturing():
            return          # hipGraph capture / device-side step counter: the step stays in Trainer.step()
        if self.opt_stream is None:
            self.opt_stream = torch.cuda.Stream(self.flat_p.device)
        for b in g.buckets:
            ev = getattr(b, "ready_event", None)
            if ev is not None:
                self.opt_stream.wait_event(ev)
                b.ready_event = None
        with torch.cuda.stream(self.opt_stream):
            g.step(self.t + 1, self.lr, self.beta1, max_norm=self.max_norm, step_dev=None)
        g.stepped = True

    def step(self):
        from ..model import packing
        self.t += 1
        if self.t_dev is not None:
            self.t_dev.add_(1.0)
        early = False
        pending = []
        for g in self.groups:
            if g.stepped:
                g.stepped, early = False, True
            else:
                pending.append(g)
        if (self.zero1 or self.seg_slots) and len(pending) == len(self.groups):
            # norms of all groups, ONE all-reduce, then every group's clip + Adam + parameter all-gather
            self.normsq_all.zero_()
            for g in pending:
                g.step_norms(zero=False)
            if self.zero1:
                dist.all_reduce(self.normsq_all, op=dist.ReduceOp.SUM, group=pending[0].pg)
            for idx in self.seg_slots:       # per-MODEL clip of a model exchanged in segments: every segment reads the model's norm
                self.normsq_all.index_copy_(0, idx, self.normsq_all.index_select(0, idx).sum().expand(idx.numel()).contiguous())
            for g in pending:
                g.step_update(self.t, self.lr, self.beta1, max_norm=self.max_norm, step_dev=self.t_dev)
        else:
            for g in pending:
                g.step(self.t, self.lr, self.beta1, max_norm=self.max_norm, step_dev=self.t_dev)
        self.disarm()       # (every ready_event was consumed by the group steps above: nothing left to wait for)
        if early:
            torch.cuda.current_stream(self.flat_p.device).wait_stream(self.opt_stream)
        packing.ACTIVE = None       # the packs are stale from here on
        self.invalidate_packs()
