#!/usr/bin/env python3
"""Command-line entry with the reference's flags (main.py:37-67) and config file (config/super_resolution.yaml, main.py:69).

`python main.py --arch tatt --mask --gradient --stu_iter_b1 3 --stu_iter_b2 3 ...` builds the same TextSR mission on the
HIP-backed modules.  The TextZoom LMDB reader and the recognisers are outside this repo's scope, so the
loop is fed synthetic (images_hr, images_lr, label_vecs) batches of the real shapes -- `--synthetic_steps` of them --
and the text priors come from `TextSR.synthetic_text_prior()`; everything between the loader and the optimizer step is
the real path.  Multi-GPU: launch with `python -m torch.distributed.run --nproc-per-node N main.py ...` (one process
per GPU, RCCL gradient all-reduce; replaces nn.DataParallel)."""
import argparse
import csv
import os

import yaml


class AttrDict(dict):
    """Tiny stand-in for easydict.EasyDict (not installed here): attribute access, recursive."""

    def __init__(self, d=None):
        super().__init__()
        for k, v in (d or {}).items():
            self[k] = AttrDict(v) if isinstance(v, dict) else v

    __getattr__ = dict.get
    __setattr__ = dict.__setitem__


def synthetic_loader(batch_size, steps, seed):
    from dpmn_amd.utils import synth
    for i in range(steps):
        b = synth.synth_batch(batch_size, seed=seed + i)
        yield b["images_hr"], b["images_lr"], b["label_vecs"]


def main(config, args):
    import torch
    import torch.distributed as dist
    from dpmn_amd.interfaces.super_resolution import TextSR
    from dpmn_amd.utils.util import set_seed
    if "RANK" in os.environ and not dist.is_initialized():
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl")
    rank = dist.get_rank() if dist.is_initialized() else 0
    # same seed on every rank while the models are built (replicas must start identical, like nn.DataParallel's replicate of
    # ONE model, base.py:160-162; the Trainer additionally broadcasts rank 0's parameters and buffers); the per-rank stream
    # (data order, augmentation, dropout seeds) is re-seeded in TextSR.train after construction
    set_seed(config.TRAIN.manualSeed)
    mission = TextSR(config, args)
    mission.rank_seed = config.TRAIN.manualSeed + rank
    bs = args.batch_size or config.TRAIN.batch_size
    os.makedirs(config.TRAIN.ckpt_dir, exist_ok=True)
    if args.test:
        result_path = os.path.join(config.TRAIN.ckpt_dir, "test_result.csv")
        if rank == 0 and not os.path.exists(result_path):
            with open(result_path, "w+") as out:
                csv.writer(out).writerow(["recognizer", "subset", "accuracy", "psnr", "ssim"])
        if args.test_data_dir and os.path.isdir(args.test_data_dir):       # a TextZoom LMDB directory (needs the lmdb package)
            from dpmn_amd.dataset.textzoom import sr_batches
            loader = sr_batches(mission.get_test_data(args.test_data_dir)[1], mission.device, mission.mask)
        else:
            loader = synthetic_loader(bs, args.synthetic_steps, 1000 + rank)
        res = mission.test(loader)
        if rank == 0:
            with open(result_path, "a") as out:
                csv.writer(out).writerow([args.rec, "synthetic", res["accuracy"], res["psnr_avg"], res["ssim_avg"]])
            print("psnr %.4f ssim %.4f over %d synthetic batches" % (res["psnr_avg"], res["ssim_avg"], args.synthetic_steps))
    else:
        log_path = os.path.join(config.TRAIN.ckpt_dir, "log.csv")
        if rank == 0 and not os.path.exists(log_path):
            with open(log_path, "w+") as out:
                csv.writer(out).writerow(["epoch", "dataset", "accuracy", "psnr_avg", "ssim_avg", "best", "best_sum"])
        dirs = config.TRAIN.train_data_dir or []
        if dirs and all(os.path.isdir(d) for d in dirs):                   # TextZoom LMDBs from the config, like base.py:85-103
            from dpmn_amd.dataset.textzoom import sr_batches
            world = dist.get_world_size() if dist.is_initialized() else 1
            if bs % world != 0 or bs // world < 2:
                raise SystemExit("main.py: batch_size %d does not shard over %d ranks in per-rank batches of >= 2 images (the global batch "
                                 "stays batch_size: nn.DataParallel's scatter, base.py:160-162)" % (bs, world))
            dl = mission.get_train_data()[1]           # per-rank shard of a per-epoch permutation (DistributedSampler)
            val_dirs = (config.TRAIN.VAL or {}).get("val_data_dir") or []
            val_dls = mission.get_val_data()[1] if val_dirs and all(os.path.isdir(d) for d in val_dirs) else []
            # eval every VAL.valInterval over every validation subset + best-model checkpoints (super_resolution.py:283-337)
            # one entry per validation subset (easy / medium / hard): evaluated, logged and check-pointed separately, best model by the sum
            val_loader = {name: (lambda v=vdl: sr_batches(v, mission.device, mission.mask))
                          for name, vdl in zip(subset_names(val_dirs), val_dls)} if val_dls else None
            mission.train(lambda epoch: sr_batches(dl, mission.device, mission.mask), epochs=config.TRAIN.epochs,
                          sampler=getattr(mission, "train_sampler", None), val_loader=val_loader)
        else:
            mission.train(synthetic_loader(bs, args.synthetic_steps, 2000 + rank), steps=args.synthetic_steps)


def subset_names(val_dirs):
    """One key per validation directory: its leaf name (the reference's data_name, super_resolution.py:286), made unique -- two
    directories with the same leaf get their parent in front, and a leaf that equals one of train()'s bookkeeping keys
    ('epoch', 'score') gets the parent too -- so that no subset silently replaces another in the per-subset tables."""
    norm = [os.path.normpath(d) for d in val_dirs]
    leaf = [os.path.basename(d) for d in norm]
    names = []
    for d, name in zip(norm, leaf):
        if leaf.count(name) > 1 or name in ("epoch", "score", ""):
            name = (os.path.basename(os.path.dirname(d)) + "_" + name).strip("_") or d
        while name in names or name in ("epoch", "score"):
            name += "_"
        names.append(name)
    return names


if __name__ == '__main__':
    parser = argparse.ArgumentParser(description='DPMN scene-text SR on MI355X (reference CLI, main.py:37-67)')
    parser.add_argument('--arch', default='tsrn', choices=['tsrn', 'tbsrn', 'tg', 'tpgsr', 'tatt'])
    parser.add_argument('--test', action='store_true', default=False)
    parser.add_argument('--test_data_dir', type=str, default='/root/data/TextZoom/test/easy')
    parser.add_argument('--batch_size', type=int, default=None)
    parser.add_argument('--resume', type=str, default=None)
    parser.add_argument('--vis_dir', type=str, default=None)
    parser.add_argument('--rec', default='aster', choices=['aster', 'moran', 'crnn'])
    parser.add_argument('--mask', action='store_true', default=False)
    parser.add_argument('--gradient', action='store_true', default=False)
    parser.add_argument('--hd_u', type=int, default=32)
    parser.add_argument('--srb', type=int, default=5)
    parser.add_argument('--STN', action='store_true', default=False)
    parser.add_argument('--patch_size', type=str, default="4,", help='1, 2, 4, 8, 16')
    parser.add_argument('--embed_dim', type=str, default="96,")
    parser.add_argument('--window_size', type=str, default="2,")
    parser.add_argument('--depths', type=str, default="1,")
    parser.add_argument('--num_heads', type=str, default="6,")
    parser.add_argument('--mlp_ratio', type=str, default="4,")
    parser.add_argument('--drop_rate', type=str, default="0,")
    parser.add_argument('--attn_drop_rate', type=str, default="0,")
    parser.add_argument('--drop_path_rate', type=str, default="0.1,")
    parser.add_argument('--rotate_train', type=float, default=0.)
    parser.add_argument('--rotate_test', type=float, default=0.)
    parser.add_argument('--stu_iter_b1', type=int, default=1)
    parser.add_argument('--stu_iter_b2', type=int, default=1)
    parser.add_argument('--tpg', default='visionlan', type=str, choices=['aster', 'moran', 'crnn', 'visionlan', None])
    parser.add_argument('--rec_path', type=str, default=None)
    parser.add_argument('--font_path', type=str, default=None)
    parser.add_argument('--sr_share', action='store_true', default=False)
    parser.add_argument('--alpha', type=float, default=0.5)
    parser.add_argument('--window_num', type=int, default=3)
    parser.add_argument('--synthetic_prior', action='store_true', default=False,
                        help='seeded noise text priors instead of the VisionLAN + glyph-atlas prior of --tpg visionlan')
    parser.add_argument('--synthetic_steps', type=int, default=20, help='number of synthetic batches to run (no dataset reader here)')
    args = parser.parse_args()
    config_path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'config', 'super_resolution.yaml')
    config = AttrDict(yaml.load(open(config_path, 'r'), Loader=yaml.Loader))
    main(config, args)
