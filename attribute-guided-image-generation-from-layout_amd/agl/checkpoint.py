"""Checkpoint I/O with the reference's file conventions (utils/model_saver_iter.py:6-87).

`save_model` / `load_model` keep the reference's behaviour: files `iter-<n>[_<appendix>].pkl` holding a plain
`state_dict()` (the key layout of the agl modules equals the reference's, so checkpoints are interchangeable in
both directions), pruning to the last `save_num` multiples of `save_step`, `iter='l'` = latest, `'s'` = scratch.
Parameters live in flat arenas (agl.flat); `load_state_dict` copies in place so the views stay valid.

The reference does not checkpoint its optimisers; `save_optimizer` / `load_optimizer` add that for the Trainer's
Adam arenas (moments + step count) under `iter-<n>_optim.pkl`.
"""
from __future__ import annotations

import os
import re
from typing import Optional

import torch

_ITER = re.compile(r"iter-(\d+)")


def _iter_of(name: str) -> Optional[int]:
    m = _ITER.search(name)
    return int(m.group(1)) if m else None


def _device_of(model) -> torch.device:
    for t in model.parameters():
        return t.device
    return torch.device("cpu")


def load_model(model, model_dir=None, appendix=None, iter='l'):
    """Returns the iteration loaded (0 = train from scratch)."""
    if iter == 's' or not os.path.isdir(model_dir) or len(os.listdir(model_dir)) == 0:
        if not os.path.isdir(model_dir):
            print('models dir not exist')
        elif len(os.listdir(model_dir)) == 0:
            print('models dir is empty')
        print('train from scratch.')
        return 0
    # network checkpoints only: `iter-<n>_optim.pkl` (save_optimizer below) is not a state_dict and must never be a candidate
    files = [f for f in os.listdir(model_dir) if f.endswith('.pkl') and not f.endswith('_optim.pkl') and _iter_of(f) is not None]
    if iter == 'l':
        cands = [f for f in files if appendix is None or appendix in f]
        if not cands:
            raise FileNotFoundError(f"no checkpoint matching {appendix!r} in {model_dir}")
        best = max(cands, key=_iter_of)
        print('load from iter: %d' % _iter_of(best))
        model.load_state_dict(torch.load(os.path.join(model_dir, best), map_location=_device_of(model)))
        return _iter_of(best)
    want = int(iter)
    for f in files:
        if _iter_of(f) == want and (appendix is None or appendix in f):
            model.load_state_dict(torch.load(os.path.join(model_dir, f), map_location=_device_of(model)))
            print('load from iter: %d' % want)
            return want
    print('there is not saved models of iter %d' % want)
    print('train from scratch.')
    return 0


def save_model(model, model_dir=None, appendix=None, iter=1, save_num=5, save_step=1000):
    keep = set(range(iter, iter - save_num * save_step, -save_step))
    os.makedirs(model_dir, exist_ok=True)
    for f in os.listdir(model_dir):
        if f.endswith('.pkl'):
            it = _iter_of(f)
            if it is not None and it not in keep:
                os.remove(os.path.join(model_dir, f))
    name = 'iter-%d_%s.pkl' % (iter, appendix) if appendix else 'iter-%d.pkl' % iter
    path = os.path.join(model_dir, name)
    torch.save(model.state_dict(), path)
    return path


def save_optimizer(trainer, model_dir, iter):
    trainer.finish()
    os.makedirs(model_dir, exist_ok=True)
    path = os.path.join(model_dir, 'iter-%d_optim.pkl' % iter)
    torch.save(trainer.optimizer_state(), path)
    return path


def load_optimizer(trainer, model_dir, iter) -> bool:
    path = os.path.join(model_dir, 'iter-%d_optim.pkl' % int(iter))
    if not os.path.isfile(path):
        return False
    trainer.load_optimizer_state(torch.load(path, map_location=trainer.dev))
    return True
