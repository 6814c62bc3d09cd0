"""TrainingEpochLoop's per-epoch reset of the validation loop's ``best_model_to_save`` flag (reference
passl/engine/loops/loop.py:200-202, ``reset_state`` at the top of every epoch) and checkpoint pruning by file name — the
round-3 advisor findings: with ``eval_interval: 2`` the epoch after a new best used to re-write ``best.*`` with
unevaluated weights; a 'latest' / 'best' anywhere in the output PATH used to switch pruning off."""
import os
import pickle
import time
import types

from passl_amd.engine.loops.loop import TrainingEpochLoop


class _Model(object):
    def __init__(self):
        self.saved = []

    def train(self):
        pass

    def save(self, prefix, rank=0):
        self.saved.append(os.path.basename(prefix))
        os.makedirs(os.path.dirname(prefix), exist_ok=True)
        with open(prefix + '.pdparams', 'wb') as f:
            pickle.dump({}, f)


class _Opt(object):
    def state_dict(self):
        return {}

    def get_lr(self):
        return 0.1


class _Val(object):
    """Improves at every evaluation."""

    def __init__(self):
        self.best_model_to_save = False
        self.best_model_metric = None
        self.latest_model_metric = None
        self.runs = 0

    def run(self):
        self.runs += 1
        self.best_model_to_save = True
        self.best_model_metric = {'metric': float(self.runs)}
        self.latest_model_metric = {'metric': float(self.runs)}


class _Loop(TrainingEpochLoop):
    def train_one_epoch(self):
        return False


def _trainer(tmp, cfg):
    t = types.SimpleNamespace()
    t.mode, t.training, t.validating = 'train', True, False
    t.train_dataloader = [0, 1]
    t.lr_decay_unit, t.lr_scheduler = 'step', None
    t.save_interval = 1
    t.config = {'Global': dict(cfg)}
    t.output_dir, t.model_name = str(tmp), 'm'
    t.model, t.optimizer = _Model(), _Opt()
    t.print_batch_step = 10
    return t


def test_best_flag_is_reset_every_epoch(tmp_path):
    tr = _trainer(tmp_path, {'eval_during_train': True, 'eval_interval': 2, 'eval_unit': 'epoch'})
    val = _Val()
    loop = _Loop(tr, epochs=4, val_loop=val)
    loop.run()
    assert val.runs == 2                                           # epochs 2 and 4
    per_epoch = [tr.model.saved[i:i + 3] for i in range(0, len(tr.model.saved), 3)]
    flat = tr.model.saved
    # `best` is written exactly by the evaluated epochs (2 and 4), never by epoch 3 with epoch 2's stale flag
    assert flat.count('best') == 2, flat
    idx3 = flat.index('epoch_3')
    assert 'best' not in flat[idx3:flat.index('epoch_4')], (flat, per_epoch)


def test_pruning_goes_by_file_name_and_epoch(tmp_path):
    out = tmp_path / 'latest_best_run'                           # 'latest' and 'best' in the PATH must not matter
    tr = _trainer(out, {'max_num_latest_checkpoint': 2})
    loop = _Loop(tr, epochs=5, val_loop=None)
    loop.run()                                                   # all five checkpoints within the same second
    names = sorted(os.listdir(os.path.join(str(out), 'm')))
    kept = sorted({n.split('.')[0] for n in names})
    assert kept == ['epoch_4', 'epoch_5', 'latest'], kept
