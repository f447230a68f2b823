#!/usr/bin/env python3
"""bench.py -- SR images/sec of the DPMN forward stack on MI355X (BASELINE.json metric).

A "step" = one pass of the hot path over one synthetic batch already resident in HBM:
frozen TATT PSN -> 3 text-prior PGRMs -> 3 mask-prior PGRMs (toMask on the GPU) -> CMM -> alpha blend,
fp32, per-GPU batch 48 (BASELINE.json configs[1]).  The K timed steps are K independent batches; by default two of them are in
flight (--pipeline 2, interfaces/super_resolution.py RefinePipeline: the single-stream PSN / CMM phases of one batch overlap the
next batch's work) -- `value` is the whole-job throughput, `one_batch_at_a_time` the same step with one batch in flight.  `--gpus N` runs N ranks, one per GPU, over RCCL: launched by the
driver under torch.distributed.run (WORLD_SIZE in the environment) or, when started as a plain `python bench.py --gpus N`,
by re-executing itself under torch.distributed.run.  Every rank runs the same step on its own batch shard (weak
scaling, no data-path collective in the forward path; `--mode train` adds the bucketed RCCL gradient exchange); the
timed region is bracketed by barrier + synchronize and the MAX over ranks is reported.

Extra objects on the JSON line:
  roofline     -- the kernel family that takes the most time in the step (picked from an all-families profile of the last
                  warm-up step, so it always names what the profile ranks first): algorithmic FLOPs (or bytes) of its
                  launches / their summed duration, both taken from HIP events the library records around EVERY launch of
                  that family inside the timed steps, on the stream it is launched on (include/dpmn_hip.h dpmn_profile_*).
                  `traffic` is static: HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), or null.
  kernels      -- the same measurement for the top families (3 extra, untimed steps after the timed region with every
                  family armed): launches/step, us/launch, TFLOP/s, GB/s, which roof bounds it and the fraction reached.
  cpu_baseline -- the CPU oracle (oracle/, kind "port") timed on this box's host cores: B = 48 forward at 32 / 64 / 128 threads
                  (best point = value); `train.cpu_baseline` = the oracle's optimisation step (autograd + clip + Adam) there.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: dense fp32 MFMA (v_mfma_f32_16x16x4_f32), no TF32 on gfx950
BF16X3_PEAK_TFLOPS = 2500.0 / 6   # "f32 via bf16x3": six v_mfma_f32_16x16x32_bf16 per fp32-class product at the ~2.5 PFLOP/s dense bf16 peak
HBM_PEAK_GBS = 8000.0             # HBM3E spec (6.3 TB/s is the measured streaming ceiling)
RIDGE = FP32_MFMA_PEAK_TFLOPS * 1e12 / (HBM_PEAK_GBS * 1e9)   # FLOP per byte above which the MFMA roof bounds a kernel


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--graph", action="store_true", help="train mode, 1 GPU: replay the step from one hipGraph capture")
    ap.add_argument("--workload", default="cfg1", choices=["cfg0", "cfg1", "cfg3", "cfg4"])
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--mode", default="fwd", choices=["fwd", "train"],
                    help="fwd: BASELINE.json configs[1] (headline metric); train: configs[2] step (loss, backward, clip+Adam, RCCL gradient exchange)")
    ap.add_argument("--drop", type=float, default=0.1,
                    help="train mode: Dropout = attn_drop = DropPath rate (default 0.1 = the reference's published recipe, "
                         "/root/reference/README.md:34; 0 = the deterministic step the parity tests run)")
    ap.add_argument("--zero1", type=int, default=None, help="train mode, N > 1: 1 = reduce-scatter + sharded clip/Adam + all-gather (default), 0 = all-reduce")
    ap.add_argument("--prior", default="synthetic", choices=["synthetic", "visionlan"],
                    help="branch-1 text priors: precomputed synthetic tensors (default) or the in-loop batched VisionLAN + glyph-atlas "
                         "pipeline inside the timed step (BASELINE.json configs[3]: 'VisionLAN text-prior branch enabled')")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16", "x3"],
                    help="arithmetic of the GEMM-shaped kernels: f32 (the reference's precision, the headline) or bf16 MFMA operands with fp32 "
                         "accumulation (BASELINE.json configs[2..4] name bf16; a separate line, never the headline); x3 = fp32 products as six "
                         "bf16 MFMAs of an exact three-term operand split in the kernels that have the variant (fp32-class results; "
                         "a separate line until its parity record is accepted)")
    ap.add_argument("--pipeline", type=int, default=2,
                    help="fwd mode: batches in flight (interfaces/super_resolution.py RefinePipeline: batch i runs on lane i %% N, so the "
                         "single-stream PSN / CMM phases of one batch overlap the next batch's work); 1 = one batch at a time")
    ap.add_argument("--no-psn-prefetch", action="store_true",
                    help="train mode: run the frozen PSN inside the step instead of prefetching the next batch's PSN image during it")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train", action="store_true", help="fwd mode: skip the `train` object (configs[2] step timed after the forward region)")
    ap.add_argument("--no-x3", action="store_true", help="default line: skip the `x3` object (the same forward / training legs in mode 2, f32 via bf16x3)")
    ap.add_argument("--train-steps", type=int, default=10, help="timed steps of the `train` object")
    ap.add_argument("--no-kernel-profile", action="store_true", help="skip the per-kernel event timing (roofline = null)")
    ap.add_argument("--cpu-sample", type=int, default=None, help="images in the CPU-baseline sample (default: the per-GPU batch)")
    return ap.parse_args()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


def kernel_row(r, steps):
    """dict from one dpmn_profile row: which roof bounds the family (arithmetic intensity vs the ridge) and how close it is."""
    s = r["total_ms"] * 1e-3
    tf = r["flops"] / s / 1e12 if s > 0 else 0.0
    gbs = r["bytes"] / s / 1e9 if s > 0 else 0.0
    mfma = r["bytes"] <= 0 or r["flops"] / max(r["bytes"], 1.0) >= RIDGE
    frac = tf / FP32_MFMA_PEAK_TFLOPS if mfma else gbs / HBM_PEAK_GBS
    return {"kernel": r["kernel"], "launches_per_step": round(r["launches"] / steps, 2), "us_per_launch": round(r["total_ms"] * 1e3 / r["launches"], 2),
            "ms_per_step": round(r["total_ms"] / steps, 4), "tflops": round(tf, 2), "gbs": round(gbs, 1), "bound": "mfma" if mfma else "hbm",
            "frac": round(frac, 4)}


def static_traffic(kernel, B, mode="fwd"):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 --pmc passes (tools/collect_profiles.py writes
    profiles/*_pmc_traffic.json; counters cannot be read from inside this process).  Only files collected over the SAME
    leg are used: `mode` "fwd" reads the forward collections, "train" the training-step ones (a family moves other bytes
    in the training forward -- the fused attention kernel also writes q / kv there)."""
    import glob
    best = (None, None)
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            rec = json.load(open(f))
            fmode = rec.get("mode") or ("train" if "TRAINING" in rec.get("workload", "") else "fwd")
            if fmode == mode and rec.get("per_gpu_batch") == B and kernel in rec.get("kernels", {}):
                best = (rec["kernels"][kernel]["hbm_bytes_per_launch"], os.path.basename(f))
        except Exception:
            pass
    return best


PGRM_ONLY_FAMILIES = ("k_gemm_pw", "k_gemm_wstat|rowreg", "k_gemm_wstat|rowreg<LN prologue>", "k_gemm_kloop", "k_dwconv_gelu",
                      "k_ln_qkv_window_attn", "k_sk_mlp_in", "k_sk_gate", "k_patch_embed_ln", "k_tail_conv2", "k_attn_fold")


def pgrm_mfma_util():
    """BASELINE.json's second metric, "PGRM MFMA-util %": MFMA-busy cycles / available SIMD cycles, time-weighted over every
    kernel of a PGRM forward at B = 48.  Counters cannot be read from inside this process: the number is STATIC, from the newest
    committed rocprofv3 --pmc pass -- profiles/*_pmc_pgrm_mfma_util.csv (tools/prof_pgrm.py: PGRM forwards only, TOTAL row) or,
    failing that, the PGRM-only families of the whole-forward pass profiles/*_pmc_mfma_util.csv."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_pgrm_mfma_util.csv")))
    try:
        if files:
            for row in csv.reader(open(files[-1])):
                if row and row[0].startswith("TOTAL"):
                    return {"value": round(float(row[6]), 2), "unit": "% MFMA-busy", "kind": "static: rocprofv3 --pmc over tools/prof_pgrm.py 48 (PGRM forwards only, time-weighted over all their kernels)",
                            "source": "profiles/" + os.path.basename(files[-1])}
        files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc_mfma_util.csv")) if "_train_" not in f)
        if files:
            busy = tot = 0.0
            for row in csv.DictReader(open(files[-1])):
                fam = row["kernel_family"]
                if fam in PGRM_ONLY_FAMILIES or fam.replace("|rowreg", "") in PGRM_ONLY_FAMILIES or fam == "k_gemm_rowreg":
                    t = float(row["launches"]) * float(row["avg_us"])
                    busy += t * float(row["mfma_busy_pct"]); tot += t
            if tot > 0:
                return {"value": round(busy / tot, 2), "unit": "% MFMA-busy", "kind": "static: PGRM-only kernel families of the whole-forward rocprofv3 --pmc pass, time-weighted",
                        "source": "profiles/" + os.path.basename(files[-1])}
    except Exception:
        pass
    return None


def cpu_baseline_worker(workload_name, n_img, threads, leg="fwd"):
    """Runs in a child process (no GPU context): the oracle on synthetic weights/inputs, prints JSON.
    leg "fwd": the eval forward, 2 warm-ups + median of 5 (first thread count) / 1 warm-up + median of 3 (further counts of
    the sweep).  leg "train": one optimisation step -- autograd through oracle/dpmn.py train_loss, per-model clip 0.25,
    Adam -- 1 warm-up + median of 3."""
    import torch
    from dpmn_amd.utils import synth
    from oracle import dpmn as odpmn
    from dpmn_amd.workload import cpu_state_dicts, cpu_priors, geom
    counts = [int(t) for t in str(threads).split(",")]
    arch, b1, b2, sd_psn, sds = cpu_state_dicts(workload_name)
    _, win, h, w = geom(workload_name)
    batch = synth.synth_batch(n_img, seed=2, h_lr=h // 2, w_lr=w // 2)
    priors = cpu_priors(workload_name, n_img)
    if leg == "train":
        from dpmn_amd.model.distill_module import DistillModule
        torch.manual_seed(2)
        sd_dist = [{k: v.clone() for k, v in DistillModule().state_dict().items()} for _ in range(b1 + b2 - 2)]
        leaf = lambda sd: {k: v.clone().requires_grad_(torch.is_floating_point(v) and "running" not in k and "index" not in k and "mask" not in k)
                           for k, v in sd.items()}
        models = [leaf(sd) for sd in sds] + [leaf(sd) for sd in sd_dist]
        params = [[v for v in m.values() if v.requires_grad] for m in models]
        opts = [torch.optim.Adam(ps, lr=1e-3, betas=(0.5, 0.999)) for ps in params]

        def run():
            for o in opts:
                o.zero_grad(set_to_none=True)
            loss = odpmn.train_loss(sd_psn, models[:b1 + b2], models[b1 + b2 + 1:], models[b1 + b2], arch, b1, b2, batch["images_lr"],
                                    batch["images_hr"], batch["label_vecs"], priors, windows=win)
            loss.backward()
            for ps, o in zip(params, opts):
                torch.nn.utils.clip_grad_norm_(ps, 0.25)
                o.step()
        plan = [(counts[0], 1, 3)]
    else:
        def run():
            with torch.no_grad():
                odpmn.refine(sd_psn, sds[:-1], sds[-1], arch, b1, b2, batch["images_lr"], batch["label_vecs"], priors, 0.5, windows=win)
        plan = [(counts[0], 2, 5)] + [(c, 1, 3) for c in counts[1:]]
    points = []
    for c, nw, nt in plan:
        torch.set_num_threads(c)
        for _ in range(nw):
            run()
        ts = []
        for _ in range(nt):
            t0 = time.perf_counter()
            run()
            ts.append(time.perf_counter() - t0)
        points.append({"threads": c, "seconds": sorted(ts)[len(ts) // 2], "all": ts})
        print(json.dumps({"points": points}), flush=True)      # a line per finished point: a timeout keeps what was measured


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_child(workload_name, n_img, counts, leg, budget_s):
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", workload_name, str(n_img), ",".join(map(str, counts)), leg]
    env = dict(os.environ, OMP_NUM_THREADS=str(max(counts)), HIP_VISIBLE_DEVICES="", CUDA_VISIBLE_DEVICES="")
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=budget_s, env=env, cwd=ROOT).stdout
    except subprocess.TimeoutExpired as e:
        out = e.stdout.decode() if isinstance(e.stdout, bytes) else (e.stdout or "")
    try:
        return json.loads(out.strip().splitlines()[-1])["points"]
    except Exception:
        return []


def cpu_baseline(workload_name, n_img, budget_s=240, train=False):
    """Oracle (CPU restatement, test infrastructure) timed on the host cores in child processes with hard time budgets: a
    reported baseline, not the target.  BASELINE.md section 4 asks for os.cpu_count() threads; torch's intra-op pool stops
    scaling far below the 256 logical cores of the GPU box on these tensors, so the forward is timed at 32 threads (2
    warm-ups + median of 5) and again at 64 and 128 (1 warm-up + median of 3): the best point is `value`, every point is
    kept in `thread_sweep`.  train=True adds the oracle's optimisation step (autograd, clip, Adam) at the best count."""
    import torch
    ncpu = os.cpu_count() or 1
    counts = [c for c in (32, 64, 128) if c <= ncpu] or [ncpu]
    points = _cpu_child(workload_name, n_img, counts, "fwd", budget_s)
    if points:
        best = min(points, key=lambda p: p["seconds"])
        value, cores = round(n_img / best["seconds"], 3), best["threads"]
        note = "best of the thread sweep (median of %d timed runs per point)" % len(best["all"])
    else:
        value, cores, note = None, counts[0], "failed within %ds budget" % budget_s
    rec = {"value": value, "unit": "images/s", "cores": cores, "kind": "port", "cpu": "%s (%d logical cores on the box)" % (cpu_model(), ncpu),
           "thread_sweep": [{"threads": p["threads"], "images_per_s": round(n_img / p["seconds"], 3)} for p in points],
           "sample": "%s forward on one batch of %d synthetic images, %s, torch %s CPU fp32 oracle" % (workload_name, n_img, note, torch.__version__)}
    if train:
        tp = _cpu_child(workload_name, n_img, [cores], "train", budget_s)
        rec["train_step"] = ({"value": round(n_img / tp[0]["seconds"], 3), "unit": "images/s", "seconds_per_step": round(tp[0]["seconds"], 3), "cores": cores,
                              "sample": "one optimisation step (oracle autograd + per-model clip 0.25 + Adam) on one batch of %d images, 1 warm-up, median of 3" % n_img}
                             if tp else {"value": None, "sample": "failed within %ds budget" % budget_s})
    return rec


PIPE_ON = [True]      # False while per-kernel durations are measured: batches in flight on several lanes overlap each other's kernels


def set_branch_streams(on):
    """The two refinement branches of a step run on two HIP streams (interfaces/super_resolution.py).  Per-kernel durations are only
    meaningful when kernels do not overlap, so the family ranking and the per-family table are measured with the branches on ONE
    stream; the timed region always runs the product configuration (two streams)."""
    from dpmn_amd.interfaces import super_resolution as sr_mod
    prev = (sr_mod.BRANCH_STREAMS, sr_mod.TRAIN_BRANCH_STREAMS)
    PIPE_ON[0] = bool(on)
    if not os.environ.get("DPMN_BRANCH_STREAMS") == "0":
        sr_mod.BRANCH_STREAMS = bool(on)
    if not os.environ.get("DPMN_TRAIN_BRANCH_STREAMS") == "0":
        sr_mod.TRAIN_BRANCH_STREAMS = bool(on)
    if not os.environ.get("DPMN_WGRAD_STREAM") == "0":      # the CMM weight gradients' side stream (train/cmm_train.py)
        from dpmn_amd.train import cmm_train
        cmm_train.WGRAD_STREAM = bool(on)
    return prev


# families whose launches all lie outside the two-stream section of a step (PSN before the fork, CMM after the join): their
# per-launch event durations inside the timed region are their own
UNFORKED_FAMILIES = ("k_conv_igemm<128,128>", "k_conv_splitk_reduce", "k_bigru", "k_mha32")
# training step: the CMM's data-gradient convs overlap its weight gradients (side stream), so only the PSN families qualify
UNFORKED_FAMILIES_TRAIN = ("k_bigru", "k_mha32")


def timed_leg(step, steps, warmup, profiling, torch, dist, _abi, post=3, arm_timed=True, unforked=UNFORKED_FAMILIES):
    """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + synchronize on both sides; MAX over ranks.
    Returns (seconds, rows of the dominant family event-timed inside the timed steps, per-family rows of `post` extra steps).
    Family ranking (last warm-up steps) and the per-family table (`post` steps after the region): branches on one stream."""
    dominant = None
    n_pick = min(2, warmup)                        # all families armed on the last warm-up steps: who is the dominant kernel?
    rank_in_warmup = profiling and arm_timed        # legs that do not arm the timed region rank the families in the post steps
    for i in range(warmup):                         # (two steps: the top two families of the forward are within 5 % of each other)
        if rank_in_warmup and i == warmup - n_pick:
            torch.cuda.synchronize()
            set_branch_streams(False)
            _abi.profile_begin(None)
        step()
        if rank_in_warmup and i == warmup - 1:
            torch.cuda.synchronize()
            rows = _abi.profile_end()
            set_branch_streams(True)
            if rows:
                dominant = max(rows, key=lambda r: r["total_ms"])["kernel"]
    if rank_in_warmup:
        step()                                      # one more warm-up in the product configuration (side streams, allocator pools)
    arm_timed = arm_timed and dominant in unforked      # else: measured in the joined-stream steps after the region
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    if profiling and dominant and arm_timed:
        _abi.profile_begin([dominant])     # only this family is bracketed by events inside the timed region
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    live = _abi.profile_end() if (profiling and dominant and arm_timed) else []
    if dist.is_initialized():
        tt = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    kernels = []
    if profiling and post:
        set_branch_streams(False)
        _abi.profile_begin(None)
        for _ in range(post):
            step()
        torch.cuda.synchronize()
        rows = _abi.profile_end()
        set_branch_streams(True)
        kernels = sorted((kernel_row(r, post) for r in rows), key=lambda k: -k["ms_per_step"])[:10]
        if not arm_timed and rows:
            if dominant is None:
                dominant = max(rows, key=lambda r: r["total_ms"])["kernel"]
            live = [r for r in rows if r["kernel"] == dominant]
            for r in live:
                r["_post_steps"] = post
    return elapsed, live, kernels


def roofline_of(live, steps, B, mode="fwd"):
    if not live:
        return None
    post = live[0].get("_post_steps")
    if post:
        steps = post
    r = kernel_row(live[0], steps)
    traffic, src = static_traffic(r["kernel"], B, mode)
    return {"kernel": r["kernel"], "bound": r["bound"],
            "achieved": r["tflops"] if r["bound"] == "mfma" else r["gbs"],
            "peak": FP32_MFMA_PEAK_TFLOPS if r["bound"] == "mfma" else HBM_PEAK_GBS,
            "unit": "TFLOP/s" if r["bound"] == "mfma" else "GB/s", "frac": r["frac"],
            "traffic": traffic, "traffic_kind": None if traffic is None else "static: per-launch mean from the rocprofv3 --pmc passes in profiles/%s" % src,
            "algorithmic_bytes_per_launch": round(live[0]["bytes"] / live[0]["launches"]),
            "flops_per_launch": round(live[0]["flops"] / live[0]["launches"]),
            "launches_timed": live[0]["launches"], "us_per_launch": r["us_per_launch"], "ms_per_step": r["ms_per_step"],
            "timing": ("HIP events around each of the %d launches of this family inside the timed steps, on the launch stream" % live[0]["launches"]) if not post else
                      ("HIP events around each of the %d launches of this family in %d untimed steps after the timed region, the two branch streams "
                       "joined into one (in the timed region this family's kernels overlap the other branch's: a launch's duration is not its own there)" % (live[0]["launches"], post))}


TRAIN_GFLOP_PER_IMAGE = 36.0      # 3 x (6 PGRM 6.81 + CMM 4.46) + PSN forward, config 1 / 2 stack


def build_train_step(args, workload, world, force_dist, dist, torch, drop):
    """BASELINE.json configs[2]: the training step of the workload's stack (own models: the forward leg's stay in eval)."""
    from dpmn_amd.loss.image_loss import ImageLoss
    from dpmn_amd.model.distill_module import DistillModule
    from dpmn_amd.train.optim import Trainer
    sr, models, psn, inp = workload.build(args.workload, batch=args.batch, drop=drop)
    spec = workload.describe(args.workload)
    b1, b2 = spec["b1"], spec["b2"]
    torch.manual_seed(2)       # every rank builds the same DistillModules (the Trainer broadcasts rank 0's anyway)
    distill = [DistillModule().to(sr.device) for _ in range(b1 + b2 - 2)]
    crit = ImageLoss(gradient=True, loss_weight=[1, 1])
    for m in models + distill:
        m.train()
        for p in m.parameters():
            p.requires_grad = True
    trainer = Trainer(models + distill, lr=1e-3, beta1=0.5, max_norm=0.25, world_size=world, zero1=args.zero1,
                      force_collectives=force_dist and dist.is_initialized())

    state = {"h": None}

    def step():
        # like TextSR.train: the frozen PSN's image of the NEXT batch (here: the same synthetic one) is computed on a lane stream of
        # its own during this step (--no-psn-prefetch: inside the step, as before round 4)
        if args.no_psn_prefetch:
            return sr.train_step(models, psn, distill, crit, trainer, inp["images_lr"], inp["images_hr"], inp.get("label_vecs"),
                                 text_priors=inp["text_priors"])
        loss = sr.train_step(models, psn, distill, crit, trainer, inp["images_lr"], inp["images_hr"], inp.get("label_vecs"),
                             text_priors=inp["text_priors"], psn_out=state["h"], prefetch=(inp["images_lr"], inp.get("label_vecs")))
        state["h"] = sr.psn_prefetched
        return loss
    if args.graph and world == 1:
        run = sr.graphed_train_step(models, psn, distill, crit, trainer, inp["images_lr"], inp["images_hr"],
                                    inp.get("label_vecs"), inp["text_priors"])

        def step():   # noqa: F811  (same batch every step, like the eager path of this bench)
            return run(inp["images_lr"], inp["images_hr"], inp.get("label_vecs"), inp["text_priors"])
    freeze_heap()
    return step, trainer, inp["images_lr"].shape[0]


def freeze_heap():
    """The models / buckets just built stay for the whole leg: moved out of the garbage collector's generations (as
    TextSR.build_training does), so a full collection inside the timed steps has nothing long-lived to walk.  Measured without it:
    one 65-85 ms host stall at a fixed step of the third in-process training leg -- the step took 44 ms instead of 23 whenever the
    launch queue ran dry."""
    import gc
    gc.collect()
    gc.freeze()


def train_object(args, workload, world, rank, force_dist, dist, torch, _abi):
    """The `train` object of the default line: the configs[2] training step timed in the same process after the forward
    region -- headline = the step the reference trains with (Dropout = attn_drop = DropPath = 0.1, /root/reference/README.md:34),
    `without_dropout` = the deterministic p = 0 step of the parity tests; both with the dominant family's roofline."""
    out = {}
    profiling = rank == 0 and not args.no_kernel_profile
    for drop in (0.1, 0.0):
        torch.cuda.synchronize()
        torch.cuda.empty_cache()            # the forward leg's cached blocks (three streams' pools) otherwise fragment this leg's arena
        step0, trainer, B = build_train_step(args, workload, world, force_dist, dist, torch, drop)
        steps, warmup = args.train_steps, max(8, args.warmup)      # three streams: the caching allocator needs a few steps to settle
        marks = []

        import gc
        n_alloc = lambda: torch.cuda.memory_stats().get("num_device_alloc", 0)
        n_gc = lambda: sum(g["collections"] for g in gc.get_stats())
        host = []

        def step():            # one event per step on the step's own stream (no synchronisation): the per-step times of the region
            t_ = time.perf_counter()
            r = step0()
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            marks.append(e)
            host.append((time.perf_counter() - t_, n_alloc(), n_gc()))
            return r
        elapsed, live, kernels = timed_leg(step, steps, warmup, profiling, torch, dist, _abi, post=3, arm_timed=False)
        ms = elapsed / steps * 1e3
        per_step = [round(marks[i - 1].elapsed_time(marks[i]), 2) for i in range(warmup, warmup + steps)]
        rec = {"ms_per_step": round(ms, 3), "images_per_s": round(world * B * steps / elapsed, 2), "steps": steps, "warmup": warmup,
               "timed_seconds": round(elapsed, 4), "dropout": drop, "per_step_ms": per_step,
               "host_issue_ms": [round(host[i][0] * 1e3, 2) for i in range(warmup, warmup + steps)],
               "device_allocs_in_timed_region": host[warmup + steps - 1][1] - host[warmup - 1][1],
               "gc_collections_in_timed_region": host[warmup + steps - 1][2] - host[warmup - 1][2],
               "whole_step_frac_of_fp32_mfma_peak": round(TRAIN_GFLOP_PER_IMAGE * 1e9 * B / (ms * 1e-3) / (FP32_MFMA_PEAK_TFLOPS * 1e12), 4),
               "roofline": roofline_of(live, steps, B, "train_x3" if _abi.lib.dpmn_get_compute_dtype() == 2 else "train")}
        if drop > 0.0:
            out.update(rec)
            out["psn_prefetch"] = not args.no_psn_prefetch
            out["what"] = ("config 2 step on the same stack with the reference's published rates (Dropout = attn_drop = DropPath = 0.1): forward, "
                           "ImageLoss + distill, backward, per-model clip 0.25, Adam%s" % (
                               "" if world == 1 else ", RCCL gradient exchange (%s)" % ("reduce-scatter + sharded clip/Adam + all-gather" if trainer.zero1 else "all-reduce")))
            out["algorithmic_gflop_per_image"] = TRAIN_GFLOP_PER_IMAGE
            out["kernels"] = kernels[:8]
        else:
            out["without_dropout"] = rec
        del step, step0, trainer, marks
        torch.cuda.empty_cache()
    return out


def x3_rows(kernels):
    """per-family rows of a mode-2 leg: the fraction of the fp32 pipe's peak stays (it can exceed 1: the products run on the bf16
    pipe), plus the fraction of the bf16x3 ceiling (2500 / 6 = 417 TFLOP/s of fp32-equivalent work) for the MFMA-bound families"""
    out = []
    for k in kernels:
        k = dict(k)
        if k["bound"] == "mfma":
            k["frac_of_bf16x3_peak"] = round(k["tflops"] / BF16X3_PEAK_TFLOPS, 4)
        out.append(k)
    return out


X3_WHAT = ("mode 2, 'f32 via bf16x3' (dpmn_set_compute_dtype(2)): fp32 tensors everywhere; in the kernel families that have the variant "
           "(implicit-GEMM conv 128 x 128, 3 x 3 halo conv on 8-row tiles, conv weight gradient, pointwise GEMM, k-loop GEMMs, Linear weight gradient, "
           "the whole-K token GEMMs with their rows in registers incl. the fused SKConv-select + LayerNorm + fc1 kernel) every "
           "operand is split exactly into three bf16 terms on the way into LDS and a product is six v_mfma_f32_16x16x32_bf16 with fp32 "
           "accumulation (dropped terms < 2^-21 |x y|: the rounding class of an fp32 multiply); every other kernel is the fp32 one.  The whole "
           "-m gpu parity suite runs in this mode too, same tolerances (tests/conftest.py).  `value` of the line stays the plain-fp32 number.")


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--cpu-baseline-worker":
        return cpu_baseline_worker(sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "fwd")
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d ranks; reporting n_gpus = %d" % (args.gpus, world, world), file=sys.stderr)
    # one process per GPU.  (Test hook: DPMN_DIST_BACKEND=gloo lets several ranks share one GPU to exercise the N > 1 code
    # path on a single-GPU box -- RCCL itself refuses two ranks on one device.)
    backend = os.environ.get("DPMN_DIST_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and world > ndev:
        raise SystemExit("bench.py: %d ranks need %d GPUs, this node has %d" % (world, world, ndev))
    local = local % max(1, ndev) if backend != "nccl" else local
    torch.cuda.set_device(local)
    force_dist = os.environ.get("DPMN_FORCE_DIST") == "1"      # exercise the RCCL code path on one GPU (world size 1)
    if world > 1 or (force_dist and "RANK" in os.environ):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    from dpmn_amd import workload, _abi
    _abi.check(_abi.lib.dpmn_set_compute_dtype({"f32": 0, "bf16": 1, "x3": 2}[args.dtype]))
    spec = workload.describe(args.workload)
    arch, b1, b2 = spec["arch"], spec["b1"], spec["b2"]
    if args.mode == "train":
        step, trainer, B = build_train_step(args, workload, world, force_dist, dist, torch, args.drop)
    else:
        sr, models, psn, inp = workload.build(args.workload, batch=args.batch)
        B = inp["images_lr"].shape[0]
        freeze_heap()
    if args.mode == "train":
        pass
    else:
        from dpmn_amd.interfaces.super_resolution import RefinePipeline
        depth = max(1, args.pipeline)
        pipe = RefinePipeline(sr, models, psn, depth) if depth > 1 else None
        kw = dict(text_prior_fn=workload.build_text_prior(sr, b1)) if args.prior == "visionlan" else dict(text_priors=inp["text_priors"])

        def step():     # every call is one more independent batch (the same synthetic one): `depth` of them are in flight
            if pipe is not None and PIPE_ON[0]:
                return pipe.submit(inp["images_lr"], inp.get("label_vecs"), **kw)
            return sr.refine(models, psn, inp["images_lr"], inp.get("label_vecs"), **kw)

    profiling = rank == 0 and not args.graph and not args.no_kernel_profile
    pipelined = args.mode == "fwd" and max(1, args.pipeline) > 1
    elapsed, live, kernels = timed_leg(step, args.steps, args.warmup, profiling, torch, dist, _abi,
                                       unforked=() if pipelined else (UNFORKED_FAMILIES if args.mode == "fwd" else UNFORKED_FAMILIES_TRAIN))
    latency_ms = None
    if pipelined:       # the same step one batch at a time (what `ms_per_step` meant before round 4's pipeline): reported next to the throughput
        PIPE_ON[0] = False
        n_lat = max(4, args.steps // 2)
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_lat):
            step()
        torch.cuda.synchronize()
        latency_ms = (time.perf_counter() - t0) / n_lat * 1e3
        PIPE_ON[0] = True
    # the same forward in mode 2 (f32 via bf16x3), same process, same models and inputs: the `x3` object of the default line
    want_x3 = (args.mode == "fwd" and args.workload == "cfg1" and args.prior == "synthetic" and not args.graph and args.dtype == "f32" and not args.no_x3)
    x3 = None
    if want_x3:
        _abi.check(_abi.lib.dpmn_set_compute_dtype(2))
        e3, live3, k3 = timed_leg(step, args.steps, max(2, args.warmup), profiling, torch, dist, _abi, unforked=() if pipelined else UNFORKED_FAMILIES)
        lat3 = None
        if pipelined:
            PIPE_ON[0] = False
            n_lat = max(4, args.steps // 2)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n_lat):
                step()
            torch.cuda.synchronize()
            lat3 = (time.perf_counter() - t0) / n_lat * 1e3
            PIPE_ON[0] = True
        _abi.check(_abi.lib.dpmn_set_compute_dtype(0))
        if rank == 0:
            roof3 = roofline_of(live3, args.steps, B, "fwd_x3")      # traffic: the mode-2 PMC passes (profiles/*_x3_pmc_traffic.json)
            if roof3 and roof3["bound"] == "mfma":
                roof3["frac_of_bf16x3_peak"] = round(roof3["achieved"] / BF16X3_PEAK_TFLOPS, 4)
            x3 = {"what": X3_WHAT, "dtype": "f32 (products as six bf16 MFMAs of a three-term operand split, fp32 accumulation)",
                  "value": round(world * B * args.steps / e3, 2), "unit": "images/s", "ms_per_step": round(e3 / args.steps * 1e3, 3),
                  "steps": args.steps, "batches_in_flight": max(1, args.pipeline),
                  "one_batch_at_a_time": None if lat3 is None else {"ms_per_step": round(lat3, 3), "images_per_s": round(world * B / lat3 * 1e3, 2)},
                  "peaks_tflops": {"fp32_mfma": FP32_MFMA_PEAK_TFLOPS, "bf16x3": round(BF16X3_PEAK_TFLOPS, 1)},
                  "roofline": roof3, "kernels": x3_rows(k3)}
    if rank == 0:
        what = "forward" if args.mode == "fwd" else "training step"
        line = {
            "metric": "SR images/sec (%s, bs=%d per GPU, fp32 %s)" % (spec["shape"], B, what),
            "value": round(world * B * args.steps / elapsed, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "bf16": "bf16 MFMA operands in the implicit-GEMM convs / pointwise GEMM, fp32 accumulation, storage and statistics",
                      "x3": "f32 via bf16x3 (implicit-GEMM / halo convs, conv weight gradients, pointwise and k-loop GEMMs: six bf16 MFMAs of an exact three-term operand split, fp32-class products; every other kernel plain f32)"}[args.dtype],
            "data": "synthetic",
            "config": {"workload": "%s: %s, %s" % (
                args.workload, spec["text"],
                ("forward-only" + (", in-loop VisionLAN + glyph-atlas text priors" if args.prior == "visionlan" else "")) if args.mode == "fwd" else "training step: ImageLoss+distill, backward, per-model clip 0.25, Adam"
                + (", dropout/attn_drop/drop_path %g" % args.drop if args.drop else "")),
                "per_gpu_batch": B, "global_batch": B * world, "ranks": world,
                "batches_in_flight": (max(1, args.pipeline) if args.mode == "fwd" else 1), "backend": (backend + (" (RCCL)" if backend == "nccl" else "") + (", collectives forced at world size 1 (DPMN_FORCE_DIST)" if world == 1 else "")) if dist.is_initialized() else None,
                "parallelism": ("dp%d (independent batch shards, no forward collective)" % world) if args.mode == "fwd" else
                               ("dp%d (coalesced gradient groups, RCCL %s overlapped with backward)" % (
                                   world, "reduce-scatter + sharded clip/Adam + all-gather" if trainer.zero1 else "all-reduce"))},
        }
        roof = roofline_of(live, args.steps, B, (args.mode if args.mode == "fwd" else "train") + ("_x3" if args.dtype == "x3" else "") if args.dtype in ("f32", "x3") else args.mode)
        if latency_ms is not None:
            line["one_batch_at_a_time"] = {"ms_per_step": round(latency_ms, 3), "images_per_s": round(world * B / latency_ms * 1e3, 2),
                                           "what": "the same step with ONE batch in flight (--pipeline 1), timed after the region on rank 0"}
        line["roofline"] = roof
        line["kernels"] = kernels
        if args.mode == "fwd" and args.workload == "cfg1":
            line["pgrm_mfma_util"] = pgrm_mfma_util()
    # the configs[2] training step, timed after the forward region in the same process (every rank takes part: the step holds
    # the RCCL gradient exchange when N > 1); headline `value` stays the forward
    train = None
    if args.mode == "fwd" and args.workload == "cfg1" and args.prior == "synthetic" and not args.no_train and not args.graph and args.dtype == "f32":
        del step
        train = train_object(args, workload, world, rank, force_dist, dist, torch, _abi)
        if want_x3:
            _abi.check(_abi.lib.dpmn_set_compute_dtype(2))
            t3 = train_object(args, workload, world, rank, force_dist, dist, torch, _abi)
            _abi.check(_abi.lib.dpmn_set_compute_dtype(0))
            if rank == 0 and x3 is not None:
                t3["kernels"] = x3_rows(t3.get("kernels", []))
                x3["train"] = t3
    if rank == 0:
        line["train"] = train
        line["x3"] = x3
        line["cpu_baseline"] = None if (args.no_cpu_baseline or world > 1 or args.mode != "fwd") else cpu_baseline(args.workload, args.cpu_sample or B, train=train is not None) if args.prior == "synthetic" else None   # N=1 only
        if train is not None and line["cpu_baseline"] and "train_step" in line["cpu_baseline"]:
            train["cpu_baseline"] = line["cpu_baseline"].pop("train_step")
        print(json.dumps(line))
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
