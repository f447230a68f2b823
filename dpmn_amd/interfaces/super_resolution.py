"""Mirror of ``interfaces/super_resolution.py::TextSR`` for the SR hot path.

``refine`` is the forward stack shared by eval()/test() in the reference (super_resolution.py:370-449 /
628-704): frozen PSN -> branch 1 (text-prior PGRMs) -> branch 2 (mask-prior PGRMs) -> CMM -> alpha
blend with the PSN image.  Every tensor op of the SR path runs in libdpmn_hip.so.  Text priors come from a callable
(``text_prior_fn(cascade_images, k) -> (B,2,H,W)`` uint8-valued floats, quirk Q6): by default the batched VisionLAN +
glyph-atlas pipeline of interfaces/text_prior.py (--rec_path), or seeded noise priors with
--synthetic_prior.  On real TextZoom data (dataset/textzoom.py) --arch tatt gets its label_vecs from the frozen CRNN mirror
(model/crnn.py, super_resolution.py:165-169).
"""
import csv
import os

import torch
from .. import _streams

from . import base
from .. import ops
from ..model.cmm import ComplementationModulationModule
from ..utils import synth


TRAIN_BRANCH_STREAMS = os.environ.get("DPMN_TRAIN_BRANCH_STREAMS", "1") != "0"
BRANCH_STREAMS = os.environ.get("DPMN_BRANCH_STREAMS", "1") != "0"      # 0: branch 1 and branch 2 of refine() on one stream
EVAL_PIPELINE = os.environ.get("DPMN_EVAL_PIPELINE", "1") != "0"        # 0: TextSR.eval / test run one batch at a time
GRAPH_MULTISTREAM = os.environ.get("DPMN_GRAPH_MULTISTREAM", "1") != "0"    # 0: graphed_train_step captures the step on ONE stream
DISTILL_ON_BRANCH = os.environ.get("DPMN_DISTILL_ON_BRANCH", "1") != "0"      # 0: the DistillModules run on the main stream before the CMM


_SIDE_STREAMS = {}


def side_streams(device):
    """The two branch streams that belong to the CURRENT stream of a device, shared by every TextSR object of the process (per-stream
    workspaces and allocator pools are then warmed once).  One pair per main stream: batches that run concurrently on different
    lanes (RefinePipeline) must not meet on a shared branch stream."""
    fixed = _streams.pool(device)          # (creates the training step's streams first: dpmn_amd/_streams.py)
    cur = torch.cuda.current_stream(device)
    if cur == torch.cuda.default_stream(device):
        return fixed["branch"]
    key = (device.type, device.index, cur.cuda_stream)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = (torch.cuda.Stream(device), torch.cuda.Stream(device))
    return _SIDE_STREAMS[key]


class RefinePipeline:
    """Throughput mode of `TextSR.refine` for a stream of INDEPENDENT batches (eval / test loops, serving): `depth` lanes, each a HIP
    stream with its own pair of branch streams; batch i runs on lane i % depth.  A batch has two single-stream phases -- the frozen
    PSN before the two branches fork and the CMM after they join -- whose launch-, ramp- and tail-bound kernels leave CUs idle; with
    two batches in flight those phases overlap the other batch's work (measured at B = 48: 8.31 -> 7.72 ms per batch).  Results are
    the same tensors `refine` returns (every module keeps one workspace per stream); a lane's work is ordered behind the caller's
    current stream at submit time, and `out.record_stream` / `wait(out)` order the caller behind the lane.
    Not for training: consecutive optimisation steps depend on each other."""

    def __init__(self, sr, model_list, model_psn, depth=2):
        assert depth >= 1
        self.sr, self.models, self.psn = sr, model_list, model_psn
        dev = next(model_list[-1].parameters()).device
        _streams.pool(dev)
        self.lanes = [torch.cuda.Stream(dev) for _ in range(depth)]
        self.events = [None] * depth
        self.i = 0

    def submit(self, images_lr, label_vecs=None, text_prior_fn=None, text_priors=None):
        """Enqueue one batch; returns its output tensor (complete when the lane reaches it: `wait(out)` or `synchronize()`)."""
        k = self.i % len(self.lanes)
        self.i += 1
        lane = self.lanes[k]
        lane.wait_stream(torch.cuda.current_stream(lane.device))      # the inputs were produced on the caller's stream
        with torch.cuda.stream(lane):
            out = self.sr.refine(self.models, self.psn, images_lr, label_vecs, text_prior_fn=text_prior_fn, text_priors=text_priors)
            ev = torch.cuda.Event()
            ev.record(lane)
        self.events[k] = ev
        out._dpmn_ready = ev
        return out

    @staticmethod
    def wait(out):
        """Order the caller's current stream behind the batch that produced `out`."""
        ev = getattr(out, "_dpmn_ready", None)
        if ev is not None:
            torch.cuda.current_stream(out.device).wait_event(ev)
            out.record_stream(torch.cuda.current_stream(out.device))
        return out

    def synchronize(self):
        for lane in self.lanes:
            lane.synchronize()


class _CombineLosses(torch.autograd.Function):
    """100 * sum(terms) / n over 0-d loss terms as ONE autograd node: cat + sum + mul + div forward, div + mul backward (every term gets
    the same gradient 100 g / n, computed once).  The reference's chain `loss += term.mean() * 100` ... `loss / n`
    (super_resolution.py:205-262) is three tiny launches per term forward and two to three backward: ~55 of the step's ~115 torch ops."""

    @staticmethod
    def forward(ctx, n, *terms):
        ctx.n, ctx.k = n, len(terms)
        return torch.cat([t.reshape(1) for t in terms]).sum() * 100 / n

    @staticmethod
    def backward(ctx, g):
        gt = (g / ctx.n) * 100
        return (None,) + (gt,) * ctx.k


class TextSR(base.TextBase):
    def build_models(self, testing=False):
        """Model list in the reference's order (super_resolution.py:38-76): b1 PGRMs (mode=False), b2 PGRMs
        (mode=True), CMM last; plus the frozen PSN."""
        b1, b2 = self.args.stu_iter_b1, self.args.stu_iter_b2
        share = self.args.sr_share
        models = [self.generator_init(0, mode=False, hidden_size=3, testing=testing)['model']]
        if not share:
            for i in range(b1 - 1):
                models.append(self.generator_init(i + 1, mode=False, hidden_size=3, testing=testing)['model'])
        models.append(self.generator_init(b1, mode=True, hidden_size=3, testing=testing)['model'])
        if not share:
            for i in range(b1, b1 + b2 - 1):
                models.append(self.generator_init(i + 1, mode=True, hidden_size=3, testing=testing)['model'])
        psn = self.generator_init(0, psn=True)['model']
        for p in psn.parameters():
            p.requires_grad = False
        psn.eval()
        cmm = ComplementationModulationModule().to(self.device)
        if testing:
            # super_resolution.py:570-582: test() evaluates the trained CMM, model_best_cmm.pth next to the PGRM files
            if not (self.resume and os.path.isdir(self.resume)):
                raise RuntimeError("dpmn_amd: test() needs --resume <dir> holding model_best_{k}.pth and model_best_cmm.pth")
            path = os.path.join(self.resume, "model_best_cmm.pth")
            if not os.path.isfile(path):
                raise FileNotFoundError("dpmn_amd: %s is missing -- refusing to evaluate a randomly initialised CMM" % path)
            print('loading pre-trained model from %s ' % path)
            sd = torch.load(path, map_location=self.device)['state_dict_G']
            cmm.load_state_dict({(k[7:] if k.startswith('module.') else k): v for k, v in sd.items()})
        models.append(cmm)
        return models, psn

    def recogniser_text_prior(self, n=None, path=None, allow_random=False):
        """The reference's branch-1 prior source (super_resolution.py:100-111, 174-199): one VisionLAN per stage, driving the
        batched GPU pipeline of interfaces/text_prior.py.  path: the reference's --rec_path directory (recognizer_best_{i}.pth
        per stage) or one visionlan.pth-style file; without one this raises unless allow_random (tests / synthetic benches)."""
        from .text_prior import VisionLANTextPrior, build_recognizers
        n = self.args.stu_iter_b1 if n is None else n
        recs = build_recognizers(n, self.device, path if path is not None else self.rec_path, allow_random=allow_random)
        return VisionLANTextPrior(recs, self.device, font_path=getattr(self.args, "font_path", None))

    def default_text_prior(self):
        """--synthetic_prior (ours): seeded uint8-valued noise priors; otherwise the recogniser-driven prior (--tpg visionlan)."""
        if getattr(self.args, "synthetic_prior", True) or getattr(self.args, "tpg", "visionlan") != "visionlan":
            return self.synthetic_text_prior()
        return self.recogniser_text_prior()

    @staticmethod
    def synthetic_text_prior(seed=2):
        def fn(cascade, k):
            B, _, H, W = cascade.shape
            return torch.floor(synth.uniform("text_prior_%d" % k, (B, 2, H, W), 0.0, 256.0, seed)).to(cascade.device)
        return fn

    @torch.no_grad()
    def refine(self, model_list, model_psn, images_lr, label_vecs=None, text_prior_fn=None, text_priors=None,
               return_all=False):
        """super_resolution.py:370-449.  text_priors: optional precomputed list of b1 tensors (B,2,H,W)."""
        b1, b2 = self.args.stu_iter_b1, self.args.stu_iter_b2
        share = self.args.sr_share
        if self.args.arch in ('tsrn', 'tbsrn', 'tg'):
            images_lr_psn = model_psn(images_lr)
        elif self.args.arch == 'tatt':
            images_lr_psn, _ = model_psn(images_lr, label_vecs)
        elif self.args.arch == 'tpgsr':      # super_resolution.py:372-377: the PSN takes the recogniser's probabilities, returns the image
            images_lr_psn = model_psn(images_lr, label_vecs)
        else:
            raise NotImplementedError(self.args.arch)
        branch1, branch2 = [], []

        def run_branch1():
            cascade = images_lr_psn
            for k in range(b1):
                x_q = text_priors[k] if text_priors is not None else text_prior_fn(cascade, k)
                x_kv = cascade[:, :3]
                sr = model_list[0 if share else k](x_q, x_kv, branch1[:k])
                branch1.append(sr)
                cascade = sr

        def run_branch2():
            cascade = images_lr_psn
            for k in range(b1, b1 + b2):
                x_q = ops.to_mask(cascade)                      # batched toMask (util.py:27-35) on the GPU
                x_kv = cascade[:, :3]
                sr = model_list[0 if share else k](x_q, x_kv, branch2[:(k - b2)])   # slice quirk Q11
                branch2.append(sr)
                cascade = sr

        # The two branches only meet in the CMM: they run on two HIP streams, so that one branch's launch-, ramp- and tail-bound
        # kernels (the K = 96 GEMMs, gate, patch embedding: a third of a PGRM's launches) overlap the other branch's work.  Shared
        # modules (--sr_share) own ONE workspace and stay on one stream.
        if BRANCH_STREAMS and not share and images_lr.is_cuda:
            cur = torch.cuda.current_stream()
            s1, s2 = self._side_streams = side_streams(images_lr.device)
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            with torch.cuda.stream(s1):
                run_branch1()
            with torch.cuda.stream(s2):
                run_branch2()
            cur.wait_stream(s1)
            cur.wait_stream(s2)
            for t_ in branch1 + branch2:
                t_.record_stream(cur)
        else:
            run_branch1()
            run_branch2()
        fused = model_list[-1](branch1[-1], branch2[-1])
        out = ops.blend(fused, images_lr_psn, self.args.alpha)
        if return_all:
            return out, dict(psn=images_lr_psn, branch1=branch1, branch2=branch2, cmm=fused)
        return out

    @torch.no_grad()
    def eval(self, model_list, val_loader, index=0, rec=None, aster_info=None, rec_list=None, model_psn=None, crnn_psn=None,
             text_prior_fn=None):
        """super_resolution.py:340-513.  PSNR/SSIM always; recognition accuracy (lines 453-493) only when `rec` is a
        callable images (B,3,H,W) -> list[str] and the loader yields label strings as a 4th item -- the reference's
        ASTER / MORAN / CRNN recognisers are out of scope, so by default 'accuracy' is
        None ("not computed"), never a fake 0.0."""
        from ..utils.util import str_filt
        for m in model_list:
            m.eval()
        fn = text_prior_fn or self.default_text_prior()
        psnr, ssim, n = [], [], 0
        n_correct, n_labelled = 0, 0
        # the batches of an evaluation pass are independent: two of them in flight (RefinePipeline).  The loop is software-pipelined:
        # batch i + 1 is prepared and SUBMITTED before this stream waits for batch i and queues its metrics -- a lane is ordered
        # behind this stream at submit time, so metrics queued before the submit would chain the lanes one after the other
        # (not with the recogniser-driven prior: the VisionLAN mirrors keep one set of activations per module, not per stream)
        pipe = None
        if EVAL_PIPELINE and self.device.type == "cuda" and not self.args.sr_share and not hasattr(fn, "recognizers"):
            key = (tuple(id(m) for m in model_list), id(model_psn))
            cached = getattr(self, "_eval_pipe", None)
            if cached is None or cached[0] != key:      # one pipeline (lane + branch streams, hence one workspace set) per model list
                cached = self._eval_pipe = (key, RefinePipeline(self, model_list, model_psn, 2))
            pipe = cached[1]

        def finish(sr, images_hr, labels):
            nonlocal n, n_correct, n_labelled
            if pipe is not None:
                RefinePipeline.wait(sr)
            p, s = ops.psnr_ssim(sr, images_hr)
            psnr.append(p)
            ssim.append(s)
            n += sr.shape[0]
            if callable(rec) and labels is not None:
                for pred, target in zip(rec(sr[:, :3]), labels):
                    n_correct += int(pred == str_filt(target, 'lower'))
                n_labelled += len(labels)

        pending = None
        for data in val_loader:
            images_hr, images_lr = data[0].to(self.device), data[1].to(self.device)
            label_vecs = data[2].to(self.device) if len(data) > 2 and data[2] is not None else None
            if label_vecs is None and self.args.arch in ('tatt', 'tpgsr'):      # real data: TATT's label_vecs come from the frozen CRNN (lines 165-169)
                label_vecs = self.label_vecs_from_crnn(images_lr)
            if getattr(self.args, "rotate_test", 0):      # super_resolution.py:358-365 (angle range from rotate_train, as there)
                images_lr, images_hr = self.rotate_pair(images_lr, images_hr, self.args.rotate_train)
            labels = data[3] if len(data) > 3 else None
            if pipe is not None:
                sr = pipe.submit(images_lr, label_vecs, text_prior_fn=fn)
                if pending is not None:
                    finish(*pending)
                pending = (sr, images_hr, labels)
            else:
                finish(self.refine(model_list, model_psn, images_lr, label_vecs, fn), images_hr, labels)
        if pending is not None:
            finish(*pending)
        psnr_avg = float(torch.stack(psnr).mean().item())
        ssim_avg = float(torch.stack(ssim).mean().item())
        accuracy = round(n_correct / n_labelled, 4) if n_labelled else None
        return {'psnr': psnr, 'ssim': ssim, 'accuracy': accuracy, 'psnr_avg': round(psnr_avg, 6), 'ssim_avg': round(ssim_avg, 6)}

    # ------------------------------------------------------------------ training (super_resolution.py:113-278)
    def build_training(self, world_size=1, group=None):
        """models (PGRMs + CMM), frozen PSN, DistillModules, ImageLoss and the flat-bucket clip+Adam / all-reduce trainer."""
        from ..loss.image_loss import ImageLoss
        from ..model.distill_module import DistillModule
        from ..train.optim import Trainer
        models, psn = self.build_models()
        b1, b2 = self.args.stu_iter_b1, self.args.stu_iter_b2
        distill = [DistillModule().to(self.device) for _ in range(b1 + b2 - 2)]
        crit = ImageLoss(gradient=self.args.gradient, loss_weight=[1, 1])
        cfg = self.config.TRAIN
        trainer = Trainer(models + distill, lr=cfg.lr, beta1=cfg.beta1, max_norm=0.25, world_size=world_size, group=group)
        for m in models + distill:
            m.train()
            for p in m.parameters():
                p.requires_grad = True
        # the modules, parameters and buckets built above live as long as the run: out of the garbage collector's generations, or a
        # full collection walks them all in the middle of a step (measured: a 65-85 ms host stall every few dozen steps, a 23 ms step
        # took 44 ms when the launch queue ran dry)
        import gc
        gc.collect()
        gc.freeze()
        return models, psn, distill, crit, trainer

    @staticmethod
    def rotate_pair(images_lr, images_hr, max_deg):
        """Rotation augmentation (super_resolution.py:144-151 / 358-365): one random angle in [-max_deg, max_deg] and one
        aspect-ratio jitter per sample (numpy RNG, like the reference), applied to the LR and HR image alike."""
        import math
        import numpy as np
        from ..utils.util import torch_rotate_img
        bs = images_lr.shape[0]
        angle = np.random.rand(bs) * max_deg * 2 - max_deg
        arc = torch.tensor(angle / 180. * math.pi).float().to(images_lr.device)
        rand_offs = torch.tensor(np.random.rand(bs)).float().to(images_lr.device)
        return torch_rotate_img(images_lr, arc, rand_offs), torch_rotate_img(images_hr, arc, rand_offs)

    def psn_forward(self, psn, images_lr, label_vecs=None):
        """The frozen PSN's image (super_resolution.py:156-169): the arch decides the call signature."""
        with torch.no_grad():
            if self.args.arch in ('tsrn', 'tbsrn', 'tg'):
                return psn(images_lr)
            if self.args.arch == 'tpgsr':
                return psn(images_lr, label_vecs)
            return psn(images_lr, label_vecs)[0]

    def prefetch_psn(self, psn, images_lr, label_vecs=None):
        """The PSN image of the NEXT batch, computed on a lane stream of its own while the current step runs: the PSN is frozen
        (super_resolution.py:56-59), so its output does not depend on the optimisation steps in between, and the step it overlaps
        has single-stream phases (CMM forward / backward, optimizer) with idle CUs.  Returns a handle for train_step(psn_out=...)."""
        dev = images_lr.device
        # one lane per device for the whole process (like the branch and weight-gradient streams): HIP maps streams onto a few hardware
        # queues in creation order, and a second TextSR object's fresh lane landed on a branch stream's queue (bench.py's second
        # training leg in one process: 27.7 vs 27.4 ms)
        lane = _streams.pool(dev)["psn"]
        lane.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(lane):
            out = self.psn_forward(psn, images_lr, label_vecs)
            ev = torch.cuda.Event()
            ev.record(lane)
        return out, ev, images_lr.data_ptr()

    def train_step(self, models, psn, distill, crit, trainer, images_lr, images_hr, label_vecs=None, text_priors=None,
                   text_prior_fn=None, psn_out=None, prefetch=None):
        """One optimisation step = super_resolution.py:140-278 (loss sum / (b1+b2+1), per-model clip 0.25, Adam).
        psn_out: the handle prefetch_psn returned for THIS batch (else the PSN runs here); prefetch = (images_lr, label_vecs) of the
        next batch: its PSN image is started behind this step's branch forward and left in self.psn_prefetched."""
        b1, b2 = self.args.stu_iter_b1, self.args.stu_iter_b2
        share = self.args.sr_share
        trainer.zero_grad()
        if getattr(self.args, "rotate_train", 0):
            images_lr, images_hr = self.rotate_pair(images_lr, images_hr, self.args.rotate_train)
            psn_out = None       # the prefetched image belongs to the unrotated batch
        hr3 = images_hr[:, :3, :]
        if psn_out is not None and psn_out[2] == images_lr.data_ptr():
            images_lr_psn, ev, _ = psn_out
            cur = torch.cuda.current_stream(images_lr.device)
            cur.wait_event(ev)
            images_lr_psn.record_stream(cur)
        else:
            images_lr_psn = self.psn_forward(psn, images_lr, label_vecs)
        br1, br2, part = [], [], [[], []]      # part / dl: the 0-d loss terms of a branch, summed once by _CombineLosses

        def run_branch1():
            cascade = images_lr_psn
            for k in range(b1):
                x_q = text_priors[k] if text_priors is not None else text_prior_fn(cascade.detach(), k)
                sr = models[0 if share else k](x_q, cascade[:, :3, :], br1[:k])
                br1.append(sr)
                cascade = sr
                part[0].append(crit(sr, hr3))

        def run_branch2():
            cascade = images_lr_psn
            for k in range(b1, b1 + b2):
                with torch.no_grad():
                    x_q = ops.to_mask(cascade.detach())      # toMask is not differentiable (PIL round trip in the reference)
                sr = models[0 if share else k](x_q, cascade[:, :3, :], br2[:(k - b2)])
                br2.append(sr)
                cascade = sr
                part[1].append(crit(sr, hr3))

        dl = [[], []]

        def run_distill(branch):
            imgs, off = (br1, 0) if branch == 0 else (br2, b1 - 1)
            feat = imgs[-1]
            for k in range(len(imgs) - 1, 0, -1):
                ld, feat = distill[k - 1 + off](feat, imgs[k - 1])
                dl[branch].append(ld)

        # two HIP streams for the two branches (see refine()); autograd runs each node's backward on its forward's stream and joins
        # the streams at the end of backward().  The step's weight packs are refreshed on the main stream before the fork.
        forked = TRAIN_BRANCH_STREAMS and not share and images_lr.is_cuda and (GRAPH_MULTISTREAM or not torch.cuda.is_current_stream_capturing())
        if forked:
            from ..model import packing
            if packing.ACTIVE is not None:
                packing.ACTIVE.ensure_fresh()
            cur = torch.cuda.current_stream()
            s1, s2 = self._side_streams = side_streams(images_lr.device)
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            # the distillation chain of a branch (super_resolution.py:219-236) needs that branch's images only: it stays on the
            # branch's stream BEHIND the event the CMM waits for, so it -- and, since autograd runs a node's backward on its forward's
            # stream, its backward -- overlaps the CMM's single-stream forward / backward instead of preceding it
            with torch.cuda.stream(s1):
                run_branch1()
                e1 = torch.cuda.Event()
                e1.record(s1)
                if DISTILL_ON_BRANCH:
                    run_distill(0)
            with torch.cuda.stream(s2):
                run_branch2()
                e2 = torch.cuda.Event()
                e2.record(s2)
                if DISTILL_ON_BRANCH:
                    run_distill(1)
            cur.wait_event(e1)
            cur.wait_event(e2)
            for t_ in br1 + br2:
                t_.record_stream(cur)
        else:
            run_branch1()
            run_branch2()
        self.psn_prefetched = None
        if prefetch is not None and not getattr(self.args, "rotate_train", 0) and not torch.cuda.is_current_stream_capturing():
            self.psn_prefetched = self.prefetch_psn(psn, *prefetch)      # overlaps the CMM forward, the backward and the optimizer
        if not (forked and DISTILL_ON_BRANCH):
            run_distill(0)
            run_distill(1)
        sr = models[-1](br1[-1], br2[-1])
        lc = crit(sr, hr3)
        if forked and DISTILL_ON_BRANCH:
            cur.wait_stream(s1)
            cur.wait_stream(s2)
        terms = part[0] + part[1] + dl[0] + dl[1] + [lc]
        if forked:
            for t_ in terms[:-1]:
                t_.record_stream(cur)
        # loss = (sum_k 100 ImageLoss_k + sum_j 100 distill_j + 100 ImageLoss_cmm) / (b1 + b2 + 1)   (super_resolution.py:205-262)
        if os.environ.get("DPMN_COMBINE_LOSSES", "1") != "0":
            loss = _CombineLosses.apply(b1 + b2 + 1, *terms)
        else:
            loss = 0
            for t_ in terms:
                loss = loss + t_.mean() * 100
            loss = loss / (b1 + b2 + 1)
        if hasattr(trainer, "arm_early_step"):
            trainer.arm_early_step()      # trainer.step() follows: a model's clip + Adam may run as soon as its backward has finished
        try:
            loss.backward()
            if forked:
                # the PGRMs' backward kernels ran on the side streams and wrote the gradient arena directly (no AccumulateGrad node the
                # autograd engine would join on): the optimizer kernels on this stream must wait for them explicitly
                cur = torch.cuda.current_stream()
                cur.wait_stream(self._side_streams[0])
                cur.wait_stream(self._side_streams[1])
            trainer.step()
        finally:
            # the promise made by arm_early_step() ends with this call whatever happened: if step() was not reached the gradients a
            # backward left on a side stream are joined here, so a caller that inspects p.grad or retries sees complete buffers
            if hasattr(trainer, "disarm"):
                trainer.disarm()
        return loss.detach()

    def graphed_train_step(self, models, psn, distill, crit, trainer, images_lr, images_hr, label_vecs=None, text_priors=None,
                           warmup=3):
        """hipGraph capture of one whole training step (~2000 launches): returns run(images_lr, images_hr, label_vecs,
        text_priors) -> loss that copies the batch into the captured step's static inputs and replays the graph.
        Static shapes, precomputed text priors, single process (the RCCL exchange is left to the eager path)."""
        if torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise RuntimeError("dpmn_amd: graphed_train_step is single-process; multi-GPU runs use train_step")
        assert text_priors is not None, "graph capture needs the text priors as inputs"
        for m in models:
            dp = getattr(m, "drop_probs", None)
            if dp is not None and (dp[0] > 0 or dp[1] > 0 or max(dp[2]) > 0):
                raise RuntimeError("dpmn_amd: graphed_train_step would freeze the Dropout / DropPath seeds (kernel arguments) into "
                                   "the captured graph and replay the same masks every step; train with train_step, or zero rates")
        trainer.device_step_counter()
        st = dict(lr=images_lr.clone(), hr=images_hr.clone(), lv=None if label_vecs is None else label_vecs.clone(),
                  tp=[t.clone() for t in text_priors])
        call = lambda: self.train_step(models, psn, distill, crit, trainer, st["lr"], st["hr"], st["lv"], text_priors=st["tp"])
        # the warm-up steps and the capture pass (which only records) must not train: snapshot everything a step mutates
        # (parameters, Adam moments and step count, BatchNorm running statistics) and put it back afterwards
        snap = trainer.state_snapshot()
        bufs = [b for m in models + distill for b in m.buffers()]
        buf_snap = [b.clone() for b in bufs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):          # allocator pools, workspaces and kernel attributes settle before capture
                call()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            loss = call()
        trainer.state_restore(snap)
        for b, s_ in zip(bufs, buf_snap):
            b.copy_(s_)

        def run(images_lr, images_hr, label_vecs=None, text_priors=None):
            st["lr"].copy_(images_lr)
            st["hr"].copy_(images_hr)
            if label_vecs is not None:
                st["lv"].copy_(label_vecs)
            if text_priors is not None:
                for d, s_ in zip(st["tp"], text_priors):
                    d.copy_(s_)
            graph.replay()
            # the replayed optimizer kernels rewrote the parameters through raw pointers: eval-mode weight packs (CMM) and the
            # PGRMs' LayerNorm-folded attention weights cached before this step are stale now (trainer.step() in Python, which
            # drops them on the eager path, ran only while the graph was captured)
            trainer.invalidate_packs()
            return loss
        run.graph = graph
        return run

    def train(self, loader=None, steps=None, val_loader=None, rec=None, epochs=None, sampler=None):
        """Training loop (super_resolution.py:125-337) over (images_hr, images_lr, label_vecs) batches (synthetic ones, or the
        TextZoom reader of dataset/textzoom.py through main.py).
        loader: a callable `loader(epoch) -> iterable` (a fresh pass per epoch), a re-iterable (list, DataLoader-like: walked
        once per epoch for `epochs` / config.TRAIN.epochs epochs, `for epoch in range(cfg.epochs)` of the reference), or a
        one-shot iterator (a single epoch).  sampler: a DistributedSampler whose set_epoch is called per epoch.
        Bookkeeping as in the reference: iters = len(loader) * epoch + j + 1 counted globally, display every displayInterval,
        eval + best-model checkpoint every VAL.valInterval when a val_loader is given (best = recognition accuracy when `rec`
        computes one, else PSNR), checkpoint.pth every saveInterval and at the end, epoch written to checkpoints and log.csv.
        Rank 0 writes the files; under ZeRO-1 every save waits for the step's parameter all-gather first (trainer.sync_params:
        the gather runs asynchronously on RCCL's stream, a state_dict clone before it lands would tear the checkpoint)."""
        dist = torch.distributed
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        models, psn, distill, crit, trainer = self.build_training(world)
        if getattr(self, "rank_seed", None) is not None:      # models are built (and broadcast): now diverge the per-rank RNG
            from ..utils.util import set_seed
            set_seed(self.rank_seed)
        fn = self.default_text_prior()
        if loader is None:
            raise RuntimeError("dpmn_amd: pass a loader of (images_hr, images_lr, label_vecs) batches")
        cfg = self.config.TRAIN
        val_int = getattr(getattr(cfg, "VAL", None), "valInterval", None) or 80
        log_path = os.path.join(getattr(cfg, "ckpt_dir", None) or ".", "log.csv")
        best, best_info, converge = None, {}, []
        best_hist = {}         # best_history_acc[data_name] of the reference (super_resolution.py:299-313)
        # validation subsets: {data_name: loader | callable -> loader}; a bare loader / callable is one subset named "val"
        if val_loader is None:
            val_sets = []
        elif isinstance(val_loader, dict):
            val_sets = list(val_loader.items())
        else:
            val_sets = [("val", val_loader)]
        if callable(loader):
            passes = loader
            n_epochs = epochs if epochs is not None else int(getattr(cfg, "epochs", 1) or 1)
        elif hasattr(loader, "__next__"):      # a generator can be walked once
            passes = lambda _e: loader
            n_epochs = 1
        else:
            passes = lambda _e: loader
            n_epochs = epochs if epochs is not None else int(getattr(cfg, "epochs", 1) or 1)

        def save(epoch, it, is_best, metric="sum"):
            if rank == 0:
                self.save_checkpoint(models, epoch, it, dict(best_hist, score=best), best_info, is_best, converge, None, metric=metric, trainer=trainer)

        def log_row(row):
            if rank == 0:
                os.makedirs(os.path.dirname(log_path) or ".", exist_ok=True)
                with open(log_path, "a+", newline="") as out:
                    csv.writer(out).writerow(row)

        it, epoch, saved_at = 0, 0, -1
        for epoch in range(n_epochs):
            if sampler is not None and hasattr(sampler, "set_epoch"):
                sampler.set_epoch(epoch)
            def on_device(data):
                hr, lr = data[0].to(self.device), data[1].to(self.device)
                lv = data[2].to(self.device) if len(data) > 2 and data[2] is not None else None
                if lv is None and self.args.arch in ('tatt', 'tpgsr'):
                    lv = self.label_vecs_from_crnn(lr)
                return hr, lr, lv
            batches = iter(passes(epoch))
            nxt = next(batches, None)
            nxt = on_device(nxt) if nxt is not None else None
            handle = None
            while nxt is not None:
                hr, lr, lv = nxt
                # one batch of look-ahead (its frozen-PSN image is computed during this step) -- fetched before the step is issued only
                # when the prefetch can be used: not past the last requested step (a one-shot loader would lose the batch), not with
                # rotate_train (the PSN sees the rotated batch, which does not exist yet)
                ahead = (steps is None or it + 1 < steps) and not getattr(self.args, "rotate_train", 0)
                nxt = next(batches, None) if ahead else None
                nxt = on_device(nxt) if nxt is not None else None
                loss = self.train_step(models, psn, distill, crit, trainer, lr, hr, lv, text_prior_fn=fn, psn_out=handle,
                                       prefetch=None if nxt is None else (nxt[1], nxt[2]))
                handle = self.psn_prefetched
                if not ahead and not (steps is not None and it + 1 >= steps):      # no look-ahead: fetch the next batch after the step was issued
                    nxt = next(batches, None)
                    nxt = on_device(nxt) if nxt is not None else None
                it += 1
                if it % cfg.displayInterval == 0 and rank == 0:
                    print('Epoch: [%d] iter %d | Loss: %f' % (epoch, it, float(loss)))
                if val_sets and it % val_int == 0:
                    # super_resolution.py:293-330: every validation subset on its own -- a log.csv row and a best-so-far checkpoint
                    # per data_name -- and the best model overall by the SUM of the subsets' scores (score = recognition accuracy
                    # when `rec` computes one, as in the reference; PSNR otherwise: the recognisers are out of scope)
                    trainer.sync_params()
                    current, psnr_d, ssim_d = {}, {}, {}
                    for data_name, vl in val_sets:
                        md = self.eval(models, vl() if callable(vl) else vl, epoch, rec=rec, model_psn=psn, text_prior_fn=fn)
                        converge.append({'iterator': it, 'acc': md['accuracy'], 'psnr': md['psnr_avg'], 'ssim': md['ssim_avg']})
                        score = md['accuracy'] if md['accuracy'] is not None else md['psnr_avg']
                        current[data_name], psnr_d[data_name], ssim_d[data_name] = float(score), md['psnr_avg'], md['ssim_avg']
                        new_best = data_name not in best_hist or score > best_hist[data_name]
                        if new_best:
                            best_hist[data_name] = float(score)
                            best_hist['epoch'] = epoch
                            best_info = {'accuracy': md['accuracy'], 'psnr': md['psnr_avg'], 'ssim': md['ssim_avg']}
                            if len(val_sets) > 1:
                                save(epoch, it, True, data_name)
                        log_row([epoch, data_name, md['accuracy'], md['psnr_avg'], md['ssim_avg']] + (["best_%s" % data_name] if new_best else []))
                    for m_ in models + distill:
                        m_.train()
                    total = sum(current.values())
                    if best is None or total > best:
                        best = total
                        best_info = {'accuracy': dict(current, epoch=epoch), 'psnr': psnr_d, 'ssim': ssim_d}
                        save(epoch, it, True)
                        log_row([epoch, "", "", "", "", "", "best_sum"])
                if it % cfg.saveInterval == 0:
                    save(epoch, it, False)
                    saved_at = it
                if steps is not None and it >= steps:
                    break
            if steps is not None and it >= steps:
                break
        if saved_at != it:
            save(epoch, it, False)
        trainer.sync_params()
        self.trainer = trainer
        return models, distill

    def label_vecs_from_crnn(self, images_lr):
        """--arch tatt on real data (super_resolution.py:165-169): label_vecs = softmax of the frozen CRNN's logits on the LR
        image, reshaped (T, B, 37) -> (B, 37, 1, T).  The CRNN is loaded from <resume>/recognizer_best_crnn.pth (line 92)."""
        crnn = getattr(self, "_crnn_psn", None)
        if crnn is None:
            from ..model.crnn import load_crnn
            path = os.path.join(self.resume, "recognizer_best_crnn.pth") if self.resume and os.path.isdir(self.resume) else None
            crnn = self._crnn_psn = load_crnn(path, self.device)
        return crnn.label_vecs(images_lr[:, :3])

    def test(self, loader=None, rec=None):
        """super_resolution.py:515-775: PGRMs from model_best_{k}.pth, CMM from model_best_cmm.pth, PSN from model_{arch}.pth
        under --resume (all required: there is no evaluation of untrained weights)."""
        models, psn = self.build_models(testing=True)
        if loader is None:
            raise RuntimeError("dpmn_amd: pass a loader of (images_hr, images_lr, label_vecs) batches: "
                               "dataset.textzoom.sr_batches(self.get_test_data(dir)[1], device) or dpmn_amd.utils.synth.synth_batch")
        return self.eval(models, loader, 0, rec=rec, model_psn=psn)
