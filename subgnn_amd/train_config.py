"""Build-owned counterpart of the reference driver SubGNN/train_config.py.

The reference driver cannot travel (it imports optuna, commentjson and pytorch-lightning 0.7.1 at
module level, train_config.py:8,21-29); this file replays the same call sequence against the same
``config.json`` schema (SubGNN/config_files/README.md:5-116) with nothing but the standard library
and torch, so the drop-in module can be exercised end to end on the GPU box:

  read json (// comments allowed)                                  train_config.py:44-52
  -> dataset paths from data.task + hyperparams_fix.embedding_type  :213-232
  -> fixed + "suggested" hyper-parameters merged into one dict      :60-86
  -> seed torch / numpy                                             :96-101
  -> SubGNN(hparams, 7 paths)                                       :104-106
  -> hyperparams.json                                               :174-183
  -> fit: prepare_data, configure_optimizers, per batch training_step -> model.backward ->
     clip grad-norm to grad_clip -> optimizer.step/zero_grad; per epoch validation_step* ->
     validation_epoch_end; keep the best monitored metric           (PL 0.7.x hook order)
  -> final_metric_scores.json from model.metric_scores[-1]          :189-193
  -> return the monitored metric                                    :196-200

There is no hyper-parameter search here: a ``FixedTrial`` answers every ``suggest_*`` call with the
first categorical choice / the lower bound (or a value supplied by the caller).
"""
import argparse
import json
import random
import re
from collections import OrderedDict
from pathlib import Path

import numpy as np
import torch

from . import config
from .SubGNN import SubGNN, dataset_paths
from .optim import ClipAdam, accelerate


def read_json(fname):
    """commentjson.load(..., object_hook=OrderedDict): strips // and /* */ comments."""
    txt = Path(fname).read_text()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    txt = re.sub(r'(^|[\s,{\[])//[^\n]*', r'\1', txt)
    txt = re.sub(r',(\s*[}\]])', r'\1', txt)
    return json.loads(txt, object_pairs_hook=OrderedDict)


class FixedTrial:
    """Deterministic stand-in for optuna.Trial: first categorical choice, lower bound of ranges."""

    def __init__(self, values=None):
        self.values, self.params = dict(values or {}), {}

    def _pick(self, name, default):
        v = self.values.get(name, default)
        self.params[name] = v
        return v

    def suggest_categorical(self, name, choices):
        return self._pick(name, choices[0])

    def suggest_float(self, name, low, high, **kw):
        return self._pick(name, float(low))

    def suggest_int(self, name, low, high, **kw):
        return self._pick(name, int(low))

    suggest_uniform = suggest_loguniform = suggest_discrete_uniform = suggest_float

    def report(self, *a, **k):
        pass

    def should_prune(self):
        return False


def get_hyperparams(run_config, trial):
    hp = dict(run_config['hyperparams_fix'])
    for name, spec in run_config.get('hyperparams_optuna', {}).items():
        hp[name] = getattr(trial, spec['type'])(name, *spec.get('args', []), **spec.get('kwargs', {}))
    return hp


def build_model(run_config, trial=None):
    trial = trial or FixedTrial()
    hp = get_hyperparams(run_config, trial)
    if 'seed' in hp:
        torch.manual_seed(hp['seed'])
        np.random.seed(hp['seed'])
        random.seed(hp['seed'])
        if torch.cuda.is_available():
            torch.cuda.manual_seed_all(hp['seed'])
    paths = dataset_paths(run_config['data']['task'], hp.get('embedding_type', 'gin'))
    return SubGNN(hp, **paths), hp


class Trainer:
    """The slice of pl.Trainer the reference uses (train_config.py:121-156): max_epochs,
    gradient clipping, validation every epoch, best-by-monitor bookkeeping."""

    def __init__(self, max_epochs, gradient_clip_val=0.0, monitor='val_micro_f1', mode='max', log=print,
                 hip_graph_step=True):
        self.max_epochs, self.clip, self.monitor, self.mode, self.log = max_epochs, gradient_clip_val, monitor, mode, log
        self.best, self.history = None, []
        # Full batches replay a recorded step (graph_step.CapturedTrainStep) unless hparams['hip_graph_step'] is False: at the
        # reference's batch sizes the eager step is bound by the host's ~250 launches (3.5-11.5 ms against 1.3-4.6 ms replayed,
        # profiles/r03_bench_standin_*.json).  A short last batch runs eagerly; a model whose step cannot be recorded (an
        # operation that needs the host inside training_step) is reported and trained eagerly.  With
        # hparams['resample_anchor_patches'] the prepared tensors change at every epoch end, so the step is recorded again
        # once per epoch (a device synchronisation + one capture: ~the cost of 3-4 eager steps per epoch).
        self.hip_graph_step = bool(hip_graph_step) and torch.cuda.is_available()
        # measurement aid (standins.bench_config's ``epoch`` object): a list makes ``fit`` append one dict of wall-clock phase
        # times per epoch (each phase then ends with a device synchronisation -- not for a production run)
        self.phase_times = None

    def _phase(self, rec, name, t0):
        if rec is None:
            return t0
        import time
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        rec[name] = rec.get(name, 0.0) + (t1 - t0)
        return t1

    def _eager_step(self, model, opt, batch, bi):
        out = model.training_step(batch, bi)
        opt.zero_grad(set_to_none=True)
        model.backward(self, out['loss'], opt, 0)
        if self.clip and self.clip > 0 and not isinstance(opt, ClipAdam):      # (ClipAdam clips inside its step)
            torch.nn.utils.clip_grad_norm_(model.parameters(), self.clip)
        opt.step()
        return out['loss'].detach()

    def _validation_outputs(self, model):
        """[validation_step(batch) for batch in val_dataloader] (train_config.py:156-186 via PL's validation loop).  With
        ``hip_graph_step`` the forward of every batch is a replay of ONE recording (graph_step.CapturedEvalStep; a short last
        batch padded with index 0, its padded rows dropped), the logits and labels of the whole epoch come to the host in one
        transfer and the per-batch loss / accuracy / F1 are evaluated there by the same functions -- an eager validation step is
        ~100 launches + three read-backs (2-6 ms at a batch of 64: as much as the epoch's training steps on the small configs)."""
        loader = model.val_dataloader()
        if not self.hip_graph_step or len(loader) == 0:
            return [model.validation_step(b, i) for i, b in enumerate(loader)]
        from .graph_step import CapturedEvalStep, StepNotRecordable
        cap = self.__dict__.get('_captured_eval')
        if cap is None or cap.model is not model or cap.stale() or cap.B != min(loader.bs, loader.n):
            cap = self.__dict__['_captured_eval'] = CapturedEvalStep(model, min(loader.bs, loader.n), 'val', warmup=1)
        kept, sizes = [], []
        try:
            for idx in loader.index_batches():
                n = idx.numel()
                if n < cap.B:
                    idx = torch.cat([idx, idx.new_zeros(cap.B - n)])
                logits, labels = cap.replay(idx)
                kept.append((logits[:n].clone(), labels[:n].clone()))
                sizes.append(n)
        except StepNotRecordable as ex:
            self.log('hip_graph_step: the validation forward could not be recorded (%s); validating eagerly' % (ex,))
            self.__dict__['_captured_eval'] = None
            return [model.validation_step(b, i) for i, b in enumerate(loader)]
        all_logits = torch.cat([a for a, _ in kept], 0).cpu()              # one transfer (and the one wait of the epoch's validation)
        all_labels = torch.cat([b for _, b in kept], 0).cpu()
        from . import ops
        ops.poll_index_errors(block=True)
        outs, lo = [], 0
        for n in sizes:
            outs.append(model.val_test_outputs('val', all_logits[lo:lo + n], all_labels[lo:lo + n].squeeze(-1)))
            lo += n
        return outs

    def fit(self, model, prepared=False):
        """``prepared``: the caller has already run prepare_data (or hotpath.prepare_sparse for graphs whose dense structures
        cannot exist)."""
        if not prepared:
            model.prepare_data()
        if model.hparams.get('gc_freeze', True):
            # everything prepare_data left behind (the loaded dataset: subgraph lists, the graph's containers) lives as long as the
            # run: moved out of the cyclic collector's sight, so that a full collection does not walk a few million long-lived
            # objects in the middle of an epoch
            import gc
            gc.collect()
            gc.freeze()
        # plain Adam over CUDA parameters (what configure_optimizers returns) becomes optim.ClipAdam: same update and clipping
        # rule, the embedding table in one HIP pass, the clip coefficient a device scalar (optim.accelerate)
        opt = accelerate(model.configure_optimizers(), self.clip, capturable=self.hip_graph_step)
        captured = None
        if self.hip_graph_step:
            from .graph_step import CapturedTrainStep, StepNotRecordable, make_capturable, make_eager
            if not isinstance(opt, ClipAdam):
                make_capturable(opt)
        import time
        for epoch in range(self.max_epochs):
            rec = None
            if self.phase_times is not None:
                rec = {'replayed_steps': 0, 'eager_steps': 0, 'recordings': 0}
                self.phase_times.append(rec)
                torch.cuda.synchronize()
            t_ph = time.perf_counter()
            model.train()
            losses = []
            loader = model.train_dataloader()
            if self.hip_graph_step:
                # full batches replay the recorded step (graph_step.py); a ragged last batch, or
                # anchors resampled at the end of the previous epoch, fall back / record again
                if captured is None or captured.stale():
                    captured = CapturedTrainStep(model, opt, loader.bs, 0.0 if isinstance(opt, ClipAdam) else self.clip,
                                                 warmup=3 if captured is None else 0)
                    if rec is not None:
                        rec['recordings'] += 1
                for bi, idx in enumerate(loader.index_batches()):
                    if idx.numel() == loader.bs and self.hip_graph_step:
                        try:
                            losses.append(captured.replay(idx)[0].clone())
                            if rec is not None:
                                rec['replayed_steps' if captured.graph is not None and captured._warm_left == 0 else 'eager_steps'] += 1
                            continue
                        except StepNotRecordable as ex:
                            # only a failure of the RECORDING falls back (an error of the eager warm-up steps or of a replay is
                            # the step's own and propagates); the optimizer returns to its eager form: the steps that follow
                            # are the ones hip_graph_step=False would have run
                            self.log('hip_graph_step: the training step could not be recorded (%s); training eagerly' % (ex,))
                            self.hip_graph_step = False
                            opt.make_eager() if isinstance(opt, ClipAdam) else make_eager(opt)
                            torch.cuda.synchronize()
                    losses.append(self._eager_step(model, opt, model.make_batch('train', idx), bi))
                    if rec is not None:
                        rec['eager_steps'] += 1
            else:
                for bi, batch in enumerate(loader):
                    losses.append(self._eager_step(model, opt, batch, bi))
                    if rec is not None:
                        rec['eager_steps'] += 1
            t_ph = self._phase(rec, 'train_steps_s', t_ph)
            model.eval()
            with torch.no_grad():
                outs = self._validation_outputs(model)
                t_ph = self._phase(rec, 'validation_steps_s', t_ph)
                if rec is not None:
                    rec['validation_batches'] = len(outs)
                res = model.validation_epoch_end(outs)
                t_ph = self._phase(rec, 'validation_epoch_end_s', t_ph)
            val = float(res['log'][self.monitor])
            if self.best is None or (val > self.best if self.mode == 'max' else val < self.best):
                self.best = val
            tl = float(torch.stack(losses).mean()) if losses else float('nan')
            self.history.append({'epoch': epoch, 'train_loss': tl, 'val_loss': float(res['avg_val_loss']), self.monitor: val})
            self.log('epoch %d  train_loss %.4f  val_loss %.4f  %s %.4f' % (epoch, tl, float(res['avg_val_loss']), self.monitor, val))
        return self

    def test(self, model):
        model.eval()
        with torch.no_grad():
            outs = [model.test_step(b, i) for i, b in enumerate(model.test_dataloader())]
            return model.test_epoch_end(outs)


def train_model(run_config, trial=None, results_dir=None, log=print):
    model, hp = build_model(run_config, trial)
    opt_cfg = run_config.get('optuna', {})
    monitor = opt_cfg.get('monitor_metric', 'val_micro_f1')
    mode = 'max' if opt_cfg.get('opt_direction', 'maximize') == 'maximize' else 'min'
    trainer = Trainer(hp['max_epochs'], hp.get('grad_clip', 0.0), monitor, mode, log,
                      hip_graph_step=bool(hp.get('hip_graph_step', True)))
    if results_dir is not None:
        Path(results_dir).mkdir(parents=True, exist_ok=True)
        with open(Path(results_dir) / 'hyperparams.json', 'w') as f:
            json.dump({k: v for k, v in hp.items()}, f, indent=2, default=str)
    trainer.fit(model)
    scores = {k: (float(v) if hasattr(v, '__float__') else v) for k, v in model.metric_scores[-1].items()}
    if results_dir is not None:
        with open(Path(results_dir) / 'final_metric_scores.json', 'w') as f:
            json.dump(scores, f, indent=2)
    return trainer.best, model, trainer


def main(argv=None):
    ap = argparse.ArgumentParser(description='Train SubGNN on MI355X from a reference-format config.json')
    ap.add_argument('-config_path', type=str, required=True)
    ap.add_argument('-project_root', type=str, default=None, help='overrides subgnn_amd.config.PROJECT_ROOT')
    ap.add_argument('-results_dir', type=str, default=None)
    args = ap.parse_args(argv)
    if args.project_root:
        config.PROJECT_ROOT = Path(args.project_root)
    run_config = read_json(args.config_path)
    best, model, trainer = train_model(run_config, results_dir=args.results_dir)
    print('best %s: %.4f' % (trainer.monitor, best))
    return best


if __name__ == '__main__':
    main()
