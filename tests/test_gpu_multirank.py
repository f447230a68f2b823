"""N > 1 code paths on ONE MI355X (RCCL refuses two ranks on a device; DPMN_DIST_BACKEND=gloo lets two ranks share it -- the
launcher, rank bookkeeping, sharding, exchange logic and file I/O are backend-independent):
(a) bench.py itself under `python -m torch.distributed.run --nproc-per-node 2` exactly as the driver launches it (forward line and
    training line, ZeRO-1 on and off): one JSON line from rank 0, n_gpus / ranks = 2, whole-job value = 2 x per-rank batch;
(b) TextSR.train (interfaces/super_resolution.py, reference loop super_resolution.py:125-337) with world size 2 over 2 epochs of a
    DistributedSampler-sharded dataset: set_epoch reshuffles, eval every valInterval, best-model and periodic checkpoints, log.csv,
    identical parameters on both ranks at the end, and a checkpoint written by rank 0 reloads bitwise-equal on both ranks."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _bench(extra, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4",
           "--no-cpu-baseline", "--no-kernel-profile"] + extra
    env = dict(os.environ, DPMN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, "exactly one JSON line (rank 0): %r" % r.stdout[-1000:]
    return json.loads(lines[0])


def test_bench_forward_two_ranks_one_line():
    d = _bench(["--no-train"])
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and d["config"]["per_gpu_batch"] == 4 and d["config"]["global_batch"] == 8
    assert d["config"]["backend"].startswith("gloo") and d["scaling"] == "weak" and d["steps"] == 2 and d["warmup"] == 1
    assert abs(d["value"] - 2 * 4 * 2 / (d["ms_per_step"] * 2 * 1e-3)) < 0.02 * d["value"]      # whole-job images/s over both ranks
    assert d["cpu_baseline"] is None        # N = 1 only


def test_bench_default_line_two_ranks_carries_train_and_x3_objects():
    """The driver's N > 1 command has no flags beyond --gpus / --steps / --warmup: every rank runs the forward leg, the two training legs
    (gradient exchange over the process group) and the same legs in mode 2; rank 0 prints one line with the `train` and `x3` objects."""
    d = _bench(["--train-steps", "2"], timeout=900)
    assert d["n_gpus"] == 2 and d["dtype"] == "f32" and d["value"] > 0
    assert d["train"]["ms_per_step"] > 0 and d["train"]["without_dropout"]["ms_per_step"] > 0
    assert len(d["train"]["per_step_ms"]) == 2
    x3 = d["x3"]
    assert x3["value"] > 0 and x3["train"]["ms_per_step"] > 0 and x3["train"]["without_dropout"]["ms_per_step"] > 0


@pytest.mark.parametrize("zero1", [1, 0])
def test_bench_training_step_two_ranks(zero1):
    d = _bench(["--mode", "train", "--zero1", str(zero1)])
    assert d["n_gpus"] == 2 and d["config"]["ranks"] == 2 and d["config"]["backend"].startswith("gloo")
    par = d["config"]["parallelism"]
    assert par.startswith("dp2") and (("reduce-scatter" in par) == bool(zero1)) and (("all-reduce" in par) == (not zero1)), par
    assert d["value"] > 0 and d["ms_per_step"] > 0


# ----------------------------------------------------------------------------------------------------------------------- (b)
class _Pairs(torch.utils.data.Dataset):
    """(images_hr, images_lr) samples with the sample index in pixel (0, 0, 0) of the HR image (to see which rank drew what)."""

    def __init__(self, n):
        from dpmn_amd.utils import synth
        b = synth.synth_batch(n, seed=77)
        self.hr, self.lr = b["images_hr"].clone(), b["images_lr"].clone()
        for i in range(n):
            self.hr[i, 0, 0, 0] = i / 256.0

    def __len__(self):
        return self.hr.shape[0]

    def __getitem__(self, i):
        return self.hr[i], self.lr[i]


def _train_worker(rank, world, port, out_dir, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from dpmn_amd import workload
        from dpmn_amd.interfaces.super_resolution import TextSR
        from dpmn_amd.utils import synth
        from dpmn_amd.utils.util import set_seed
        B, b1, b2 = 2, 1, 2
        cfg = workload.make_config(B)
        cfg.TRAIN.epochs, cfg.TRAIN.saveInterval, cfg.TRAIN.displayInterval = 2, 3, 100
        cfg.TRAIN.ckpt_dir = out_dir
        cfg.TRAIN.VAL = SimpleNamespace(valInterval=4)
        args = workload.make_args("tsrn", b1, b2, B)
        args.vis_dir = "run"
        set_seed(cfg.TRAIN.manualSeed)                  # as main.py: same seed while the replicas are built
        sr = TextSR(cfg, args)
        sr.vis_dir = out_dir
        sr.rank_seed = cfg.TRAIN.manualSeed + rank
        ds = _Pairs(16)
        sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=True, drop_last=True)
        dl = torch.utils.data.DataLoader(ds, batch_size=B, sampler=sampler, drop_last=True)
        seen = []

        def passes(epoch):
            for hr, lr in dl:
                seen.append((epoch, [int(round(float(v) * 256)) for v in hr[:, 0, 0, 0]]))
                yield hr, lr
        vb = synth.synth_batch(4, seed=5)
        val = [(vb["images_hr"][:2], vb["images_lr"][:2]), (vb["images_hr"][2:], vb["images_lr"][2:])]
        models, distill = sr.train(passes, val_loader={"easy": lambda: val[:1], "hard": lambda: val[1:]}, sampler=sampler)
        trainer = sr.trainer
        torch.cuda.synchronize()
        p1 = trainer.flat_p.clone()
        others = [torch.zeros_like(p1) for _ in range(world)]
        dist.all_gather(others, p1)
        same = all(torch.equal(others[0], o) for o in others)
        if rank == 0:
            sr.save_checkpoint(models, 1, 8, {}, {}, True, [], metric="final", trainer=trainer)
        dist.barrier()
        reload_ok = True
        fresh, _ = sr.build_models()
        for i, (m_new, m_old) in enumerate(zip(fresh, models)):
            ck = torch.load(os.path.join(out_dir, "ckpt", "model_best_final_1_%d.pth" % i), map_location=dev)
            m_new.load_state_dict(ck["state_dict_G"])
            # parameters on every rank; buffers (BatchNorm running statistics are per-rank, as under nn.DataParallel only replica
            # 0's persist) on the rank that wrote the file
            pnames = {k for k, _ in m_old.named_parameters()}
            for (k, a), (_, b_) in zip(m_new.state_dict().items(), m_old.state_dict().items()):
                if k in pnames or rank == 0:
                    reload_ok = reload_ok and torch.equal(a, b_)
        q.put((rank, same, reload_ok, seen))
        dist.barrier()
    except Exception:
        import traceback
        traceback.print_exc()
        q.put((rank, False, False, []))
    finally:
        dist.destroy_process_group()


def test_train_loop_world2_two_epochs_eval_checkpoint_reload(tmp_path):
    out = str(tmp_path)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_train_worker, args=(r, 2, port, out, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = {r[0]: r for r in (q.get(timeout=600) for _ in procs)}
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank in (0, 1):
        _, same, reload_ok, seen = res[rank]
        assert same, "ranks hold different parameters after 2 epochs (rank %d)" % rank
        assert reload_ok, "rank %d: the checkpoint rank 0 wrote does not reload bitwise-equal" % rank
        assert len(seen) == 8 and [e for e, _ in seen] == [0] * 4 + [1] * 4, "2 epochs x 4 steps per rank"
    # the two ranks' shards of an epoch are disjoint and cover the dataset; set_epoch reshuffled the second epoch
    for epoch in (0, 1):
        idx = [sorted(i for e, b in res[r][3] if e == epoch for i in b) for r in (0, 1)]
        assert not set(idx[0]) & set(idx[1]) and sorted(idx[0] + idx[1]) == list(range(16)), idx
    assert [b for e, b in res[0][3] if e == 0] != [b for e, b in res[0][3] if e == 1], "DistributedSampler.set_epoch was not applied"
    ck = os.listdir(os.path.join(out, "ckpt"))
    assert "checkpoint.pth" in ck and any(f.startswith("model_best_sum_") for f in ck), ck
    # super_resolution.py:293-330: every validation subset is evaluated, logged and check-pointed on its own (evals at iterations
    # 4 and 8, written by rank 0 only), the best model overall goes by the sum over the subsets
    assert any(f.startswith("model_best_easy_") for f in ck) and any(f.startswith("model_best_hard_") for f in ck), ck
    rows = open(os.path.join(out, "log.csv")).read().strip().splitlines()
    assert sum(",easy," in r_ for r_ in rows) == 2 and sum(",hard," in r_ for r_ in rows) == 2, rows
    assert any(r_.endswith("best_easy") for r_ in rows) and any(r_.endswith("best_hard") for r_ in rows) and any(r_.endswith("best_sum") for r_ in rows), rows
    assert len(rows) <= 6, rows
