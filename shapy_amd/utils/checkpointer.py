"""Checkpointer (reference: regressor/human_shape/utils/checkpointer.py:11-124).

Same on-disk format (``torch.save({'model': state_dict, ...})``, a ``latest_checkpoint`` text
pointer) and the same load order: ``<save_dir>/best_checkpoint`` ->
``<pretrained>/checkpoints/latest_checkpoint`` -> nothing.  ``strict=False`` like the
reference; the HIP engine re-packs its weights after ``load_state_dict`` (post-hooks)."""
import logging
import os
import os.path as osp

import torch

logger = logging.getLogger('shapy_amd')


class Checkpointer(object):
    def __init__(self, model, optimizer=None, scheduler=None, adv_optimizer=None, pretrained='',
                 distributed=False, rank=0, save_dir='/tmp/exp'):
        self.rank = rank
        self.distributed = distributed
        self.model = model
        self.optimizer, self.scheduler, self.adv_optimizer = optimizer, scheduler, adv_optimizer
        self.save_dir = save_dir
        if self.rank == 0:
            os.makedirs(self.save_dir, exist_ok=True)
        self.pretrained = pretrained

    def save_checkpoint(self, name, **kwargs):
        if self.rank > 0:
            return
        ckpt_data = {'model': self.model.state_dict()}
        for key in ('optimizer', 'scheduler', 'adv_optimizer'):
            obj = getattr(self, key)
            if obj is not None:
                ckpt_data[key] = obj.state_dict()
        ckpt_data.update(kwargs)
        fn = osp.join(self.save_dir, name)
        torch.save(ckpt_data, fn)
        with open(osp.join(self.save_dir, 'latest_checkpoint'), 'w') as f:
            f.write(fn)

    def load_checkpoint(self):
        save_fn = osp.join(self.save_dir, 'best_checkpoint')
        load_pretrained = False
        if not osp.exists(save_fn):
            if len(self.pretrained) > 1:
                self.pretrained = osp.expandvars(self.pretrained)
                load_pretrained = True
                save_fn = osp.join(self.pretrained, 'checkpoints', 'latest_checkpoint')
            if not osp.exists(save_fn):
                logger.warning('No checkpoint found in %s!', self.save_dir)
                return {}
        map_location = torch.device('cpu')
        try:
            latest = save_fn
            ckpt_data = torch.load(latest, map_location=map_location, weights_only=False)
        except Exception:
            with open(save_fn, 'r') as f:
                latest = f.read().strip()
            ckpt_data = torch.load(latest, map_location=map_location, weights_only=False)
        logger.warning('Loading checkpoint from %s!', latest)
        missing, unexpected = self.model.load_state_dict(ckpt_data['model'], strict=False)
        if missing:
            logger.warning('The following keys were not found: %s', missing)
        if unexpected:
            logger.warning('The following keys were not expected: %s', unexpected)
        if not load_pretrained:
            for key in ('optimizer', 'scheduler', 'adv_optimizer'):
                obj = getattr(self, key)
                if obj is not None and key in ckpt_data:
                    obj.load_state_dict(ckpt_data[key])
        else:
            ckpt_data['iteration'] = 0
            ckpt_data['epoch_number'] = 0
        return ckpt_data
