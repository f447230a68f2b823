"""Mirror of ``interfaces/base.py::TextBase`` restricted to the SR hot path.

Keeps the attribute names, the per-PGRM hyper-parameter string parsing (base.py:64-82, with a safe
parser instead of eval()), ``generator_init`` (base.py:127-198) and the checkpoint format
(base.py:328-373).  Out of scope here: LMDB datasets, recognisers, pygame renderer.
"""
import os

import torch

from ..model import pgrm, cmm, tsrn, tatt, tbsrn
from ..utils import ssim_psnr


def parse_list(s):
    """'2,4,8,' -> [2, 4, 8]; replaces eval() of base.py:64-82."""
    out = []
    for tok in str(s).split(','):
        tok = tok.strip()
        if tok:
            out.append(float(tok) if ('.' in tok or 'e' in tok.lower()) else int(tok))
    return out


class TextBase(object):
    def __init__(self, config, args, opt_TPG=None):
        self.config = config
        self.args = args
        self.scale_factor = self.config.TRAIN.down_sample_scale
        self.rec_path = getattr(args, "rec_path", None)
        self.resume = args.resume if getattr(args, "resume", None) is not None else config.TRAIN.resume
        self.batch_size = args.batch_size if args.batch_size is not None else self.config.TRAIN.batch_size
        val = getattr(config.TRAIN, "VAL", None)     # base.py:56
        self.vis_dir = args.vis_dir if getattr(args, "vis_dir", None) is not None else (getattr(val, "vis_dir", None) or "./vis")
        self.voc_type = getattr(config.TRAIN, "voc_type", None)
        if not torch.cuda.is_available():
            raise RuntimeError("dpmn_amd: the DPMN hot path needs a MI355X (no CPU fallback)")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.cal_psnr = ssim_psnr.calculate_psnr
        self.cal_ssim = ssim_psnr.SSIM()
        self.mask = self.args.mask
        self.depths = parse_list(self.args.depths)
        self.patch_size = parse_list(self.args.patch_size)
        self.embed_dim = parse_list(self.args.embed_dim)
        ws = parse_list(self.args.window_size)
        nh = parse_list(self.args.num_heads)
        self.window_size, self.num_heads = [], []
        pre = 0
        for _ in self.depths:
            self.window_size.append(ws[pre:pre + self.args.window_num])
            pre += self.args.window_num
        pre = 0
        for layer_num in self.depths:
            self.num_heads.append(nh[pre:pre + layer_num])
            pre += layer_num
        self.mlp_ratio = parse_list(self.args.mlp_ratio)
        self.drop_rate = parse_list(self.args.drop_rate)
        self.attn_drop_rate = parse_list(self.args.attn_drop_rate)
        self.drop_path_rate = parse_list(self.args.drop_path_rate)

    # ------------------------------------------------------------------ data (base.py:85-125)
    def _loader(self, dirs, test, shuffle, drop_last, shard=False, gpu_finish=True):
        """shard=True (training under torch.distributed): every rank walks its own 1/world of a per-epoch permutation
        (DistributedSampler, call `self.train_sampler.set_epoch(epoch)`) with batch_size // world samples per step, so the
        GLOBAL batch stays config.TRAIN.batch_size at the configured learning rate -- nn.DataParallel's scatter of one batch
        over the GPUs (base.py:160-162), not world x batch_size."""
        from ..dataset import textzoom as tz
        cfg = self.config.TRAIN
        sets = [tz.lmdbDataset_real(root=d, voc_type=cfg.voc_type, max_len=cfg.max_len, test=test) for d in dirs]
        ds = torch.utils.data.ConcatDataset(sets)
        dist = torch.distributed
        world = dist.get_world_size() if (shard and dist.is_initialized()) else 1
        sampler, bs = None, self.batch_size
        if world > 1:
            sampler = torch.utils.data.distributed.DistributedSampler(ds, num_replicas=world, rank=dist.get_rank(), shuffle=shuffle,
                                                                      drop_last=drop_last)
            bs = max(self.batch_size // world, 2)       # B_local = 1 changes SKConv's squeeze() semantics (quirk Q3)
        loader = torch.utils.data.DataLoader(
            ds, batch_size=bs, shuffle=shuffle and sampler is None, sampler=sampler, num_workers=int(cfg.workers), pin_memory=True,
            drop_last=drop_last,
            collate_fn=tz.alignCollate_realWTLAMask(imgH=cfg.height, imgW=cfg.width, down_sample_scale=cfg.down_sample_scale, mask=self.mask,
                                                    gpu_finish=gpu_finish))      # True: ToTensor + mask channel on the GPU by sr_batches
        # (dataset/textzoom.py; uint8 pixels in the batch); False: the reference's float (B, 3 + mask, H, W) tensors for consumers that
        # iterate the loader themselves
        if shard:
            self.train_sampler = sampler
        return ds, loader

    def get_train_data(self):
        cfg = self.config.TRAIN
        if not isinstance(cfg.train_data_dir, list):
            raise TypeError('check trainRoot')
        return self._loader(cfg.train_data_dir, False, True, True, shard=True)

    def get_val_data(self):
        pairs = [self.get_test_data(d) for d in self.config.TRAIN.VAL.val_data_dir]
        return [p[0] for p in pairs], [p[1] for p in pairs]

    def get_test_data(self, dir_):
        return self._loader([dir_], True, True, False)

    def generator_init(self, iter=0, mode=True, psn=False, hidden_size=64, testing=False):
        cfg = self.config.TRAIN
        kw = dict(scale_factor=self.scale_factor, width=cfg.width, height=cfg.height, STN=self.args.STN, mask=self.mask,
                  srb_nums=self.args.srb, hidden_units=self.args.hd_u)
        if psn and self.args.arch in ('tsrn', 'tg'):
            model = tsrn.TSRN(**kw)
        elif psn and self.args.arch == 'tatt':
            model = tatt.TSRN_TL_TRANS(**kw)
        elif psn and self.args.arch == 'tbsrn':
            model = tbsrn.TBSRN(**kw)
        elif psn and self.args.arch == 'tpgsr':       # base.py:141-144
            model = tsrn.TSRN_TL(**kw)
        elif psn:
            raise NotImplementedError("dpmn_amd: PSN arch %r is not built (built: tsrn, tg, tatt, tbsrn, tpgsr)" % self.args.arch)
        else:
            # base.py:151 never passes img_size, so the reference PGRM is hard-wired to 32x128 outputs (weight_list_i is
            # (1, hidden, 32, 128), pgrm.py:497, quirk Q7).  Here the size follows the config (height x width), which is the
            # identical call for the 32x128 recipe and lets the 64x256 stress configuration run at all.
            model = pgrm.PGRM(img_size=[cfg.height, cfg.width],
                              patch_size=self.patch_size, embed_dim=self.embed_dim, depths=self.depths,
                              num_heads=self.num_heads, window_size=self.window_size, mlp_ratio=self.mlp_ratio,
                              drop_rate=self.drop_rate, attn_drop_rate=self.attn_drop_rate,
                              drop_path_rate=self.drop_path_rate, iter=iter, mode=mode, hidden_size=hidden_size)
        from ..loss.image_loss import ImageLoss
        image_crit = ImageLoss(gradient=self.args.gradient, loss_weight=[1, 1])
        model = model.to(self.device)
        if self.resume and (psn or testing):
            if os.path.isdir(self.resume):
                name = "model_{}.pth".format(self.args.arch) if psn else "model_best_" + str(iter) + ".pth"
                path = os.path.join(self.resume, name)
            else:
                path = self.resume
            print('loading pre-trained model from %s ' % path)
            model.load_state_dict(torch.load(path, map_location=self.device)['state_dict_G'])
        return {'model': model, 'crit': image_crit}

    def save_checkpoint(self, netG_list, epoch, iters, best_acc_dict, best_model_info, is_best, converge_list,
                        recognizer=None, metric="sum", trainer=None):
        """Same files and dict keys as base.py:328-358: model_best_{metric}_{epoch}_{i}.pth when is_best, otherwise every
        model overwrites checkpoint.pth (the reference's behaviour, line 358).  Recogniser files (base.py:360-373) are written
        when a recogniser list is passed.  trainer (ours): the train/optim.py Trainer that owns the parameters -- its pending
        ZeRO-1 parameter all-gathers are awaited before any state_dict is cloned (they write the flat parameter arena from
        RCCL's stream; a clone racing them would store step t-1 values for the shards other ranks own)."""
        if trainer is not None:
            trainer.sync_params()
        ckpt_path = os.path.join(self.vis_dir, 'ckpt')
        os.makedirs(ckpt_path, exist_ok=True)
        for i, netG in enumerate(netG_list):
            # parameters may be views into the trainer's flat buckets: store compact, storage-independent copies
            save_dict = {'state_dict_G': {k: v.detach().clone() for k, v in netG.state_dict().items()},
                         'info': {'arch': self.args.arch, 'iters': iters, 'epochs': epoch, 'batch_size': self.batch_size,
                                  'voc_type': getattr(self, "voc_type", None), 'up_scale_factor': self.scale_factor},
                         'best_history_res': best_acc_dict, 'best_model_info': best_model_info,
                         'param_num': sum(p.numel() for p in netG.parameters()), 'converge': converge_list}
            fname = 'model_best_{}_{}_{}.pth'.format(metric, epoch, i) if is_best else 'checkpoint.pth'
            torch.save(save_dict, os.path.join(ckpt_path, fname))
        if recognizer is not None:
            recs = recognizer if isinstance(recognizer, list) else [recognizer]
            for i, r in enumerate(recs):
                if isinstance(recognizer, list):
                    fname = ('recognizer_best_{}_{}_{}.pth' if is_best else 'recognizer_{}_{}_{}.pth').format(metric, epoch, i)
                else:
                    fname = 'recognizer_best.pth' if is_best else 'recognizer.pth'
                torch.save(r.state_dict(), os.path.join(ckpt_path, fname))
        return ckpt_path
