"""One SubGNN training step captured in a hipGraph.

With the reference's hyper-parameters (batch of 64 subgraphs, D = 128) a training step is ~250 small
kernel launches: on MI355X the GPU finishes each of them before the host has issued the next one, so
the step time is the host's launch time.  The step itself is static once the batch indices are data
rather than control flow -- gathers from the split-resident tensors, the three channels' message
passing, the read-out, the loss, backward, gradient clipping and Adam never look at a value on the
host (SubGNN.make_batch(trim=False), SubGNN.initialize_cc_embeddings and subgraph_utils.calc_accuracy
were written for that) -- so it is recorded once and replayed with a new index vector per batch.

The recorded sequence is exactly Trainer.fit's body (training_step -> zero_grad -> model.backward ->
clip_grad_norm_ -> optimizer.step, train_config.py: PL 0.7.x hook order).  What changes from the
eager step: batches are not trimmed to their widest row (PAD columns add zeros, S.py:1098-1110 is a
memory optimisation) and Adam runs with ``capturable=True`` (its step counter lives on the device).

Anything that replaces tensors the graph reads -- SubGNN._prepare_anchors_only after
``resample_anchor_patches``, a new prepare_data -- invalidates the recording; ``stale()`` reports it
and the trainer records again.
"""
import torch


def make_capturable(optimizer):
    """Adam keeps ``step`` on the host unless told otherwise; a host counter cannot be replayed."""
    for g in optimizer.param_groups:
        if 'capturable' in g:
            g['capturable'] = True
    for st in optimizer.state.values():
        if 'step' in st and torch.is_tensor(st['step']) and not st['step'].is_cuda:
            p = next(v for v in st.values() if torch.is_tensor(v) and v.is_cuda)
            st['step'] = st['step'].to(p.device)
    return optimizer


def make_eager(optimizer):
    """Undo make_capturable for an optimizer that goes back to eager steps (the trainer's fallback): with ``capturable``
    left on, torch's Adam keeps the device-side step arithmetic -- not the code path of a run that never asked for a
    recorded step.  The step counters stay tensors (host ones for the unfused optimizer, as it creates them)."""
    fused = any(g.get('fused') for g in optimizer.param_groups)
    for g in optimizer.param_groups:
        if 'capturable' in g:
            g['capturable'] = False
    if not fused:                                   # (torch's fused Adam keeps its step counters on the device in either mode)
        for st in optimizer.state.values():
            if 'step' in st and torch.is_tensor(st['step']) and st['step'].is_cuda:
                st['step'] = st['step'].cpu()
    return optimizer


def abandon_capture(model, optimizer):
    """After a capture that raised: the Python side of the step body ran although none of its kernels did.  Gradient buffers it
    took or handed back belong to the dead capture's memory pool and were never zeroed (``ops.take_zeroed`` popped the table's
    pre-zeroed buffer, ``ops.release_zeroed`` may have hung an un-executed one on a parameter), ``p.grad`` may point into that
    pool, and the layer bodies it queued are still waiting.  Drop all of it: the eager steps that follow start from fresh fills."""
    from . import ops
    params = list(getattr(optimizer, 'all', None) or getattr(optimizer, 'big', None) or [])
    seen = {id(p) for p in params}
    params += [p for p in model.parameters() if id(p) not in seen]
    for p in params:
        ops.drop_zeroed(p)
        p.grad = None
    ops.drop_lazy_mpn()
    model.__dict__['_tapped_table'] = None
    model.__dict__['_fwd_cache'] = None


class StepNotRecordable(RuntimeError):
    """The training step could not be RECORDED (an operation inside it needs the host while the stream is capturing).
    Raised by CapturedTrainStep._record only: an error of the eager warm-up steps, or of a replay, is the step's own
    failure and propagates as it is."""


class CapturedTrainStep:
    """Record ``step(idx)`` for batches of exactly ``batch_size`` subgraphs of ``split``.

    ``replay(idx)`` copies the indices into the graph's input and launches it; the returned loss and
    accuracy are the graph's static outputs (clone them to keep them past the next replay)."""

    def __init__(self, model, optimizer, batch_size, clip=0.0, split='train', warmup=3):
        if not torch.cuda.is_available():
            raise RuntimeError('CapturedTrainStep needs the GPU: there is no CPU path')
        self.model, self.opt, self.B, self.clip, self.split = model, optimizer, int(batch_size), clip, split
        self.idx = torch.zeros(self.B, dtype=torch.int64, device=model.device)
        self._token = self._anchor_token()
        if hasattr(optimizer, 'param_groups'):
            make_capturable(optimizer)
        elif not getattr(optimizer, 'capturable', False):          # optim.ClipAdam: its step counts must live on the device
            raise ValueError('CapturedTrainStep needs an optimizer whose step count is device-resident (ClipAdam(capturable=True))')
        self.graph, self.loss, self.acc = None, None, None
        self._warm_left = warmup

    # -- what the recording depends on -----------------------------------------------------
    def _anchor_token(self):
        """The model's preparation generation: every code path that replaces a tensor the recording reads
        (prepare_data / prepare_test_data, hotpath.prepare_sparse, the anchor resample, the structure
        column lists) bumps it (SubGNN._bump_generation).  Object ids would miss the similarity tensors
        and can be reused after garbage collection."""
        return self.model.__dict__.get('_prep_generation', 0)

    def stale(self):
        return self._anchor_token() != self._token

    # -- the step, identical in eager and recorded form -------------------------------------
    def _body(self):
        m = self.model
        batch = m.make_batch(self.split, self.idx, trim=False)
        out = m.training_step(batch, 0)
        self.opt.zero_grad(set_to_none=True)
        m.backward(None, out['loss'], self.opt, 0)
        if self.clip and self.clip > 0:
            torch.nn.utils.clip_grad_norm_(m.parameters(), self.clip)
        self.opt.step()
        return out['loss'].detach(), out['log']['train_acc'].detach()

    def _record(self):
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        try:
            with torch.cuda.graph(g):
                self.loss, self.acc = self._body()
        except RuntimeError as ex:
            self.loss = self.acc = None
            abandon_capture(self.model, self.opt)
            raise StepNotRecordable(str(ex)) from ex
        self.graph = g

    def replay(self, idx):
        """One training step on subgraphs ``idx`` (length must be ``batch_size``)."""
        idx = torch.as_tensor(idx)
        if idx.numel() != self.B:
            raise ValueError('captured step takes %d indices, got %d' % (self.B, idx.numel()))
        self.idx.copy_(idx.view(-1), non_blocking=True)
        if self.graph is None:
            if self._warm_left > 0:                       # lazy initialisations (handles, optimizer state,
                self._warm_left -= 1                      # autograd buffers) must not land in the recording
                return self._body()
            self._record()
        self.graph.replay()
        # the replayed Adam step changed the master table without moving its host-side version counter
        self.model.invalidate_half_table()
        return self.loss, self.acc


class CapturedEvalStep:
    """The forward pass of a validation / test batch (SubGNN.val_test_step's device half: make_batch + forward, eval mode, no
    gradients) recorded once for batches of exactly ``batch_size`` subgraphs of ``split`` and replayed with a new index vector:
    an eager validation step is ~100 launches the host takes 2-6 ms to queue, a replay one.  A short last batch is padded with
    index 0 by the caller (every row of a forward pass depends on its own subgraph only -- BatchNorm runs on its running
    statistics in eval mode -- so the padded rows change nothing and are dropped).  ``replay`` returns the recording's static
    logits and labels (clone to keep them past the next replay)."""

    def __init__(self, model, batch_size, split='val', warmup=1):
        if not torch.cuda.is_available():
            raise RuntimeError('CapturedEvalStep needs the GPU: there is no CPU path')
        self.model, self.B, self.split = model, int(batch_size), split
        self.idx = torch.zeros(self.B, dtype=torch.int64, device=model.device)
        self._token = model.__dict__.get('_prep_generation', 0)
        self.graph = self.logits = self.labels = None
        self._warm_left = warmup

    def stale(self):
        return self.model.__dict__.get('_prep_generation', 0) != self._token

    def _body(self):
        m = self.model
        batch = m.make_batch(self.split, self.idx, trim=False)
        return m._forward_batch(self.split, batch), batch['label']

    def replay(self, idx):
        idx = torch.as_tensor(idx)
        if idx.numel() != self.B:
            raise ValueError('captured step takes %d indices, got %d' % (self.B, idx.numel()))
        if self.model.training or torch.is_grad_enabled():
            raise RuntimeError('CapturedEvalStep replays an eval-mode, no-grad forward: call model.eval() under torch.no_grad()')
        self.idx.copy_(idx.view(-1), non_blocking=True)
        if self.graph is None:
            if self._warm_left > 0:
                self._warm_left -= 1
                return self._body()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    self.logits, self.labels = self._body()
            except RuntimeError as ex:
                self.logits = self.labels = None
                abandon_capture(self.model, None)
                raise StepNotRecordable(str(ex)) from ex
            self.graph = g
        self.graph.replay()
        return self.logits, self.labels
