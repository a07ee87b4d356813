"""Gradient clipping + Adam as the reference's caller runs them per step (train_config.py: Trainer(gradient_clip_val)
-> clip_grad_norm_, then SubGNN.configure_optimizers' torch.optim.Adam, SubGNN/SubGNN.py:1156-1161), with the one large
parameter -- the (N+1, D) embedding table -- updated by one HIP pass (sgnn_adam_step) instead of a multiply by the clip
coefficient, four chunked multi-tensor launches and a zero fill of the gradient buffer on the next pass.

The small parameters stay with torch's fused Adam (one launch for all of them).  Same update rule, same clipping rule
(coefficient = min(1, max_norm / (total_norm + 1e-6)) over ALL parameters); the table's clip coefficient is a device
scalar read by the kernel, so the step has no host round trip."""
import copy

import torch

from . import ops


class ClipAdam:
    """``step()`` = clip_grad_norm_(params, max_norm) followed by Adam(params, lr).step();  ``zero_grad()`` as usual.
    After ``step()`` the gradient of a large parameter is gone (``p.grad is None``): its buffer, zeroed by the update
    kernel, hangs on the parameter (``ops.release_zeroed``) and becomes the next backward's accumulator -- a caller that
    kept a reference to ``p.grad`` across ``step()`` holds that recycled buffer, not the old gradient.  ``release()``
    drops the kept buffers."""

    def __init__(self, params, lr, max_norm=None, betas=(0.9, 0.999), eps=1e-8, big_bytes=16 << 20, capturable=False, fuse_tail=True,
                 skip_untouched_rows=True):
        params = [p for p in params if p.requires_grad]
        self.betas, self.eps, self.max_norm = (float(betas[0]), float(betas[1])), float(eps), max_norm
        # one parameter group, torch-shaped: a learning-rate scheduler (or a caller) that writes param_groups[0]['lr'] is
        # honoured by the next step (``lr`` below reads it); 'params' lists every parameter this optimizer updates
        self.param_groups = [{'params': list(params), 'lr': float(lr), 'betas': self.betas, 'eps': self.eps, 'weight_decay': 0,
                              'amsgrad': False, 'maximize': False, 'max_norm': max_norm}]
        self.big = [p for p in params if p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
                    and p.numel() * 4 >= big_bytes and p.data_ptr() % 16 == 0]
        ids = {id(p) for p in self.big}
        self.small = [p for p in params if id(p) not in ids]
        # capturable: every step count lives on the device, so that a step recorded into a hipGraph (hotpath.CapturedTraining)
        # replays with the right bias corrections; same arithmetic either way
        self.capturable = bool(capturable)
        self.small_opt = torch.optim.Adam(self.small, lr=float(lr), betas=betas, eps=eps, capturable=self.capturable,
                                          fused=all(p.is_cuda for p in self.small)) if self.small else None
        self.state = {id(p): {'step': 0, 'exp_avg': torch.zeros_like(p), 'exp_avg_sq': torch.zeros_like(p),
                              'step_dev': torch.zeros(1, dtype=torch.int64, device=p.device) if self.capturable else None}
                      for p in self.big}
        # every parameter float32, contiguous, on one GPU (the model's case): the whole tail -- norm of all gradients, clip
        # coefficient, Adam on all of them -- is two launches (ops.OptimTail) instead of torch's multi-tensor norm, multiply and
        # fused Adam (two 40-50 us launches: pow() in double per thread) around the table's pass: ~155 -> ~30 us per step
        self.tail = None
        devs = {p.device for p in params}
        if fuse_tail and params and len(devs) == 1 and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() for p in params):
            self.all = self.big + self.small                    # the table first: its workgroups start first
            ids = {id(p) for p in self.big}
            self.small_opt = None
            for p in self.small:
                self.state[id(p)] = {'step': 0, 'exp_avg': torch.zeros_like(p), 'exp_avg_sq': torch.zeros_like(p), 'step_dev': None}
            self.tail = ops.OptimTail(self.all, [self.state[id(p)]['exp_avg'] for p in self.all],
                                      [self.state[id(p)]['exp_avg_sq'] for p in self.all], [id(p) in ids for p in self.all],
                                      row_skip=range(len(self.big)) if skip_untouched_rows else ())
            self.counters = torch.zeros(len(self.all), dtype=torch.int64, device=params[0].device) if self.capturable else None
            self.last_clip = None                               # (2,) device tensor [coefficient, total norm] of the last step

    @property
    def lr(self):
        return float(self.param_groups[0]['lr'])

    @lr.setter
    def lr(self, value):
        self.param_groups[0]['lr'] = float(value)

    def _step_fused(self):
        which, grads, takes = [], [], []
        for i, p in enumerate(self.all):
            g = p.grad
            if g is None:
                continue
            take = g.is_contiguous() and g.dtype == torch.float32
            which.append(i)
            grads.append(g if take else g.contiguous().float())
            takes.append(take)
        if not which:
            return
        steps = None
        if self.counters is None:
            steps = []
            for i in which:
                st = self.state[id(self.all[i])]
                st['step'] += 1
                steps.append(st['step'])
        self.last_clip = self.tail.step(which, grads, self.lr, self.betas, self.eps, self.max_norm, steps=steps,
                                        step_counters=self.counters)
        for i, g, take in zip(which, grads, takes):
            p = self.all[i]
            if self.tail.zero[i] and take:                      # (the kernel zeroed the gradient it consumed)
                ops.release_zeroed(p, g)
                p.grad = None
            elif self.tail.zero[i]:
                p.grad = None

    def step(self):
        if self.tail is not None:
            return self._step_fused()
        if self.small_opt is not None:
            for g in self.small_opt.param_groups:               # (a scheduler writes THIS optimizer's group)
                g['lr'] = self.lr
        small_grads = [p.grad for p in self.small if p.grad is not None]
        big = [p for p in self.big if p.grad is not None]
        scale = None
        if self.max_norm is not None and (small_grads or big):
            fast = [p.grad for p in big if p.grad.is_contiguous() and p.grad.dtype == torch.float32 and p.grad.data_ptr() % 16 == 0]
            slow = [p.grad for p in big if not (p.grad.is_contiguous() and p.grad.dtype == torch.float32 and p.grad.data_ptr() % 16 == 0)]
            scale = ops.clip_coefficient(fast, small_grads + slow, self.max_norm)
            if small_grads:
                torch._foreach_mul_(small_grads, scale[0])
        if self.small_opt is not None:
            self.small_opt.step()
        for p in big:
            st = self.state[id(p)]
            st['step'] += 1
            g = p.grad
            take = g.is_contiguous() and g.dtype == torch.float32 and g.data_ptr() % 16 == 0
            if not take:
                g = g.contiguous().float()
            # the kernel zeroes the gradient it has just consumed: the buffer goes back to the fused ops'
            # table-gradient accumulator as it is (ops.take_zeroed) instead of a 256 MB fill per pass
            ops.adam_step(p.data, g, st['exp_avg'], st['exp_avg_sq'], self.lr, self.betas, self.eps, st['step'],
                          grad_scale=scale, zero_grad=take, step_counter=st['step_dev'])
            if take:
                ops.release_zeroed(p, g)
                p.grad = None

    def release(self):
        """Drop the zeroed gradient buffers kept on the large parameters (256 MB for the benchmark's table)."""
        for p in self.big:
            ops.drop_zeroed(p)

    def zero_grad(self, set_to_none=True):
        if self.small_opt is not None:
            self.small_opt.zero_grad(set_to_none=set_to_none)
        for p in (self.big if self.tail is None else self.all):
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()


    # -- checkpointing (torch.optim.Optimizer's surface: what a Lightning-style caller saves and restores) ----------------------
    def _order(self):
        return self.all if self.tail is not None else self.big

    def state_dict(self):
        """Moments, step counts and the row-skip bytes of every parameter this optimizer updates itself, by position in
        ``param_groups[0]['params']`` (as torch keys optimizer state), + torch's own state dict of the small parameters when
        they are stepped by torch's fused Adam.  Device step counters are read back (one host round trip)."""
        pos = {id(p): i for i, p in enumerate(self.param_groups[0]['params'])}
        counts = self.counters.tolist() if (self.tail is not None and self.counters is not None) else None
        state = {}
        for k, p in enumerate(self._order()):
            st = self.state[id(p)]
            step = counts[k] if counts is not None else (int(st['step_dev'].item()) if st.get('step_dev') is not None else st['step'])
            ent = {'step': int(step), 'exp_avg': st['exp_avg'].detach().clone(), 'exp_avg_sq': st['exp_avg_sq'].detach().clone()}
            if self.tail is not None and k in self.tail.seen:
                ent['rows_seen'] = self.tail.seen[k].detach().clone()
            state[pos[id(p)]] = ent
        group = {k: v for k, v in self.param_groups[0].items() if k != 'params'}
        group['params'] = list(range(len(self.param_groups[0]['params'])))
        return {'state': state, 'param_groups': [group],
                'small': copy.deepcopy(self.small_opt.state_dict()) if self.small_opt is not None else None}

    def load_state_dict(self, sd):
        """Inverse of ``state_dict``.  A checkpoint without the row-skip bytes (moments restored from elsewhere) marks every row
        whose first moment is non-zero as seen -- a row with m = v = 0 is exactly the row Adam leaves alone, so skipping only
        those stays bit-identical to the full update."""
        params = self.param_groups[0]['params']
        for k, v in sd['param_groups'][0].items():
            if k != 'params':
                self.param_groups[0][k] = v
        self.max_norm = self.param_groups[0].get('max_norm', self.max_norm)
        order = {id(p): k for k, p in enumerate(self._order())}
        for i, ent in sd['state'].items():
            p = params[int(i)]
            st = self.state[id(p)]
            st['exp_avg'].copy_(ent['exp_avg'])
            st['exp_avg_sq'].copy_(ent['exp_avg_sq'])
            st['step'] = int(ent['step'])
            if st.get('step_dev') is not None:
                st['step_dev'].fill_(int(ent['step']))
            k = order[id(p)]
            if self.tail is not None and self.counters is not None:
                self.counters[k] = int(ent['step'])
            if self.tail is not None and k in self.tail.seen:
                seen = ent.get('rows_seen')
                if seen is None:
                    seen = ((st['exp_avg'] != 0) | (st['exp_avg_sq'] != 0)).reshape(p.shape[0], -1).any(1).to(torch.uint8)
                self.tail.seen[k].copy_(seen)
        if self.small_opt is not None and sd.get('small') is not None:
            self.small_opt.load_state_dict(copy.deepcopy(sd['small']))

    def make_eager(self):
        """Back to host-side step counts (the trainer's fallback when a step cannot be recorded): same arithmetic."""
        self.capturable = False
        if self.small_opt is not None:
            from .graph_step import make_eager
            make_eager(self.small_opt)
        if self.tail is not None and self.counters is not None:
            done = self.counters.tolist()
            for p, n in zip(self.all, done):
                self.state[id(p)]['step'] = int(n)
            self.counters = None
        for st in self.state.values():
            st['step_dev'] = None
        return self


TRAINER_BIG_BYTES = 6 << 20        # the embedding table of the stand-ins (7.5-30 MB), not their per-split component embeddings (4.6 MB each: torch's fused multi-tensor Adam takes those together)


def accelerate(optimizer, max_norm=None, capturable=False, big_bytes=TRAINER_BIG_BYTES):
    """What ``train_config.Trainer`` steps with: the optimizer ``configure_optimizers`` returned when it is anything but a plain
    ``torch.optim.Adam`` over CUDA parameters -- else a ClipAdam with the same learning rate, betas and eps that also applies
    the trainer's ``gradient_clip_val`` (so the caller must NOT clip again): the embedding table (and any other parameter of at
    least ``big_bytes``) is updated by one ``sgnn_adam_step`` launch and the clip coefficient is a device scalar -- at a batch
    of 64 torch's chunked multi-tensor Adam over a 9 MB table and the ten small launches of ``clip_grad_norm_`` were ~140 us of
    a 1.5 ms step (PPI-BP stand-in).  Same update rule (tests/test_gpu_float.py::test_clip_adam_matches_torch)."""
    if isinstance(optimizer, ClipAdam):
        return optimizer
    if type(optimizer) is not torch.optim.Adam or len(optimizer.param_groups) != 1 or len(optimizer.state) != 0:
        return optimizer
    g = optimizer.param_groups[0]
    if g.get('weight_decay', 0) or g.get('amsgrad', False) or g.get('maximize', False) or g.get('differentiable', False):
        return optimizer
    params = [p for p in g['params'] if p.requires_grad]
    if not params or not all(p.is_cuda for p in params) or torch.is_tensor(g['lr']):
        return optimizer
    return ClipAdam(params, g['lr'], max_norm=(max_norm if max_norm and max_norm > 0 else None), betas=g['betas'], eps=g['eps'],
                    big_bytes=big_bytes, capturable=capturable)
