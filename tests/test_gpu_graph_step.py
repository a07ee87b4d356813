"""-m gpu: the training step replayed from a hipGraph (subgnn_amd/graph_step.py) takes the model to
the same place as the eager step -- same batches, same losses, same parameters -- including across an
anchor resample, which invalidates the recording.  The backward pass has no float atomics any more
(sorted segmented sums, per-row partials), so what is left between the two runs is the batch trimming
(the recorded step keeps the split's padded widths: PAD columns add zeros but regroup the sums) and Adam's
capturable form: tolerances of 1e-5."""
import json

import pytest
import torch

from helpers import write_dataset_from_golden
from test_gpu_train_driver import CONFIG

pytestmark = pytest.mark.gpu


def _fit(tiny, root, graph, resample, epochs=7, extra=None):
    from subgnn_amd import config, train_config
    write_dataset_from_golden(tiny, root, 'ds')
    fix = dict(tiny.hp)
    for k in ('batch_size', 'learning_rate', 'n_layers'):
        fix.pop(k, None)
    fix.update({'max_epochs': epochs, 'seed': 3, 'lin_dropout': 0.0, 'compute_similarities': True,
                'resample_anchor_patches': resample, 'hip_graph_step': graph})
    fix.update(extra or {})
    (root / 'config.json').write_text(CONFIG % json.dumps(fix))
    config.PROJECT_ROOT = root
    rc = train_config.read_json(root / 'config.json')
    trial = train_config.FixedTrial({'batch_size': 4, 'n_layers': 2})
    torch.manual_seed(11)
    best, model, trainer = train_config.train_model(rc, trial=trial, log=lambda *a: None)
    return model, trainer


@pytest.mark.parametrize('resample', [False, True])
def test_captured_step_matches_eager(tiny, tmp_path, resample):
    (tmp_path / 'a').mkdir()
    (tmp_path / 'b').mkdir()
    m0, t0 = _fit(tiny, tmp_path / 'a', False, resample)
    m1, t1 = _fit(tiny, tmp_path / 'b', True, resample)
    assert t1.hip_graph_step and not t0.hip_graph_step
    l0 = torch.tensor([e['train_loss'] for e in t0.history])
    l1 = torch.tensor([e['train_loss'] for e in t1.history])
    assert torch.allclose(l0, l1, rtol=1e-5, atol=1e-6), (l0, l1)
    assert l1[-1] < l1[0]
    sd0, sd1 = m0.state_dict(), m1.state_dict()
    assert sd0.keys() == sd1.keys()
    for k in sd0:
        a, b = sd0[k].float(), sd1[k].float()
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-5), (k, (a - b).abs().max())
    # the validation epochs: the recorded forward (graph_step.CapturedEvalStep: every batch one replay, the short last batch padded,
    # one transfer per epoch, metrics on the host copies) reports what the eager validation steps report
    assert t1.__dict__.get('_captured_eval') is not None and t1._captured_eval.graph is not None
    assert len(m0.metric_scores) == len(m1.metric_scores) == 7
    for e0, e1 in zip(m0.metric_scores, m1.metric_scores):
        assert e0.keys() == e1.keys()
        for k in e0:
            a, b = float(e0[k]), float(e1[k])
            assert (a != a and b != b) or abs(a - b) <= 2e-5 * max(1.0, abs(a)), (k, a, b)


def test_captured_step_rejects_other_batch_size(tiny, tmp_path):
    from subgnn_amd.graph_step import CapturedTrainStep
    (tmp_path / 'a').mkdir()
    m, t = _fit(tiny, tmp_path / 'a', False, False, epochs=1)
    opt = m.configure_optimizers()
    cap = CapturedTrainStep(m, opt, 4, 1.0)
    with pytest.raises(ValueError):
        cap.replay(torch.arange(3))
    m.train()
    for _ in range(5):                                      # 3 eager warm-ups, 1 recording, replays
        loss, acc = cap.replay(torch.arange(4))
    assert cap.graph is not None and torch.isfinite(loss) and 0.0 <= float(acc) <= 1.0


def test_captured_step_with_fp16_table_keeps_validation_fresh(tiny, tmp_path):
    """embedding_dtype='fp16' + hip_graph_step: a replayed Adam step changes the master table without
    moving its host-side version counter.  The half copy the eager validation reads must follow it --
    after every epoch the copy equals the rounded master, and the run matches the eager fp16 run."""
    (tmp_path / 'a').mkdir()
    (tmp_path / 'b').mkdir()
    m0, t0 = _fit(tiny, tmp_path / 'a', False, False, epochs=4, extra={'embedding_dtype': 'fp16'})
    m1, t1 = _fit(tiny, tmp_path / 'b', True, False, epochs=4, extra={'embedding_dtype': 'fp16'})
    assert t1.hip_graph_step
    with torch.no_grad():                     # a read that does not refresh by itself: what the last validation saw
        for m in (m0, m1):
            assert torch.equal(m._half_table().float(), m.node_embeddings.weight.detach().half().float())
    v0 = torch.tensor([float(e['val_loss']) for e in m0.metric_scores])
    v1 = torch.tensor([float(e['val_loss']) for e in m1.metric_scores])
    assert len(v0) == 4 and torch.allclose(v0, v1, rtol=1e-4, atol=1e-5), (v0, v1)
    # the table moved during training, so a stale copy would have shown
    assert float((m1.node_embeddings.weight.detach() - torch.from_numpy(tiny['embeddings']).to(m1.device)).abs().max()) > 0


def test_recording_goes_stale_when_prepared_tensors_are_replaced(tiny, tmp_path):
    from subgnn_amd.graph_step import CapturedTrainStep
    (tmp_path / 'a').mkdir()
    m, t = _fit(tiny, tmp_path / 'a', False, False, epochs=1)
    cap = CapturedTrainStep(m, m.configure_optimizers(), 4, 1.0)
    assert not cap.stale()
    m._build_sim_cols()                      # nothing replaced: same patches
    assert not cap.stale()
    m.prepare_data()                         # every prepared tensor is a new object now
    assert cap.stale()


def test_training_step_is_bit_reproducible(tiny, tmp_path):
    """Two models built and trained identically end with identical bits in every parameter: nothing in the
    step (forward, backward, clipping, Adam) depends on the order in which workgroups happen to run."""
    (tmp_path / 'a').mkdir()
    (tmp_path / 'b').mkdir()
    m0, _ = _fit(tiny, tmp_path / 'a', False, False, epochs=3)
    m1, _ = _fit(tiny, tmp_path / 'b', False, False, epochs=3)
    sd0, sd1 = m0.state_dict(), m1.state_dict()
    for k in sd0:
        assert torch.equal(sd0[k], sd1[k]), k


def test_step_that_cannot_be_recorded_is_trained_eagerly(tiny, tmp_path, monkeypatch):
    """hip_graph_step is the trainer's default: a training_step with a host round trip in it (here: the loss read back on the
    host) cannot be recorded -- the trainer says so, switches the recording off and trains eagerly; the run still equals the
    eager run of the same model."""
    from subgnn_amd.SubGNN import SubGNN
    (tmp_path / 'a').mkdir()
    (tmp_path / 'b').mkdir()
    real = SubGNN.training_step

    def syncing_step(self, batch, batch_idx):
        out = real(self, batch, batch_idx)
        float(out['loss'].detach())                          # a host synchronisation: illegal while a stream is capturing
        return out
    monkeypatch.setattr(SubGNN, 'training_step', syncing_step)
    m0, t0 = _fit(tiny, tmp_path / 'a', False, False, epochs=4)
    m1, t1 = _fit(tiny, tmp_path / 'b', True, False, epochs=4)
    assert not t0.hip_graph_step and not t1.hip_graph_step    # asked for, could not be recorded, reported and switched off
    l0 = torch.tensor([e['train_loss'] for e in t0.history])
    l1 = torch.tensor([e['train_loss'] for e in t1.history])
    assert torch.allclose(l0, l1, rtol=1e-5, atol=1e-6), (l0, l1)
    x = torch.ones(8, device=m1.device)
    assert float((x * 2).sum()) == 16.0                       # the device is healthy after the failed capture


def test_an_error_of_the_eager_warm_up_steps_is_not_taken_for_a_failed_recording(tiny, tmp_path, monkeypatch):
    """only a failure inside the RECORDING falls back to eager training; a genuine error of the step itself (here:
    raised by the second call, one of the three eager warm-up steps of CapturedTrainStep) reaches the caller as it is."""
    from subgnn_amd.SubGNN import SubGNN
    (tmp_path / 'a').mkdir()
    real = SubGNN.training_step
    calls = {'n': 0}

    def failing_step(self, batch, batch_idx):
        calls['n'] += 1
        if calls['n'] == 2:
            raise RuntimeError('boom: the step itself is broken')
        return real(self, batch, batch_idx)
    monkeypatch.setattr(SubGNN, 'training_step', failing_step)
    with pytest.raises(RuntimeError, match='boom'):
        _fit(tiny, tmp_path / 'a', True, False, epochs=2)


def test_fallback_returns_the_optimizer_to_its_eager_form(tiny, tmp_path, monkeypatch):
    from subgnn_amd import graph_step
    opt = torch.optim.Adam([torch.nn.Parameter(torch.zeros(4, device='cuda'))], lr=0.1)
    graph_step.make_capturable(opt)
    assert all(g['capturable'] for g in opt.param_groups)
    graph_step.make_eager(opt)
    assert not any(g['capturable'] for g in opt.param_groups)


def test_trainer_steps_with_clipadam_and_matches_plain_adam(tiny, tmp_path, monkeypatch):
    """The trainer swaps configure_optimizers' plain Adam for optim.ClipAdam (optim.accelerate: the table in one sgnn_adam_step
    launch, clip coefficient on the device, no separate clip_grad_norm_): same losses and parameters as the plain path
    (clip_grad_norm_ + torch's Adam), eager and replayed."""
    from subgnn_amd import optim, train_config
    for d in 'abc':
        (tmp_path / d).mkdir()
    seen = []
    real = optim.accelerate

    def spy(opt, *a, **k):
        out = real(opt, *a, **k)
        seen.append(type(out).__name__)
        return out
    monkeypatch.setattr(train_config, 'accelerate', spy)
    m1, t1 = _fit(tiny, tmp_path / 'a', False, False, epochs=4)
    m2, t2 = _fit(tiny, tmp_path / 'b', True, False, epochs=4)
    assert seen == ['ClipAdam', 'ClipAdam']
    monkeypatch.setattr(train_config, 'accelerate', lambda opt, *a, **k: opt)
    m0, t0 = _fit(tiny, tmp_path / 'c', False, False, epochs=4)
    l0 = torch.tensor([e['train_loss'] for e in t0.history])
    for t in (t1, t2):
        l = torch.tensor([e['train_loss'] for e in t.history])
        assert torch.allclose(l0, l, rtol=1e-5, atol=1e-6), (l0, l)
    sd0 = m0.state_dict()
    for m in (m1, m2):
        sd = m.state_dict()
        for k in sd0:
            assert torch.allclose(sd0[k].float(), sd[k].float(), rtol=1e-5, atol=1e-5), k
