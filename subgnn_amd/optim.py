"""Gradient clipping + Adam as the reference's caller runs them per step (train_config.py: Trainer(gradient_clip_val)
-> clip_grad_norm_, then SubGNN.configure_optimizers' torch.optim.Adam, SubGNN/SubGNN.py:1156-1161), with the one large
parameter -- the (N+1, D) embedding table -- updated by one HIP pass (sgnn_adam_step) instead of a multiply by the clip
coefficient, four chunked multi-tensor launches and a zero fill of the gradient buffer on the next pass.

The small parameters stay with torch's fused Adam (one launch for all of them).  Same update rule, same clipping rule
(coefficient = min(1, max_norm / (total_norm + 1e-6)) over ALL parameters); the table's clip coefficient is a device
scalar read by the kernel, so the step has no host round trip."""
import torch

from . import ops


class ClipAdam:
    """``step()`` = clip_grad_norm_(params, max_norm) followed by Adam(params, lr).step();  ``zero_grad()`` as usual.
    After ``step()`` the gradient of a large parameter is gone (``p.grad is None``): its buffer, zeroed by the update
    kernel, hangs on the parameter (``ops.release_zeroed``) and becomes the next backward's accumulator -- a caller that
    kept a reference to ``p.grad`` across ``step()`` holds that recycled buffer, not the old gradient.  ``release()``
    drops the kept buffers."""

    def __init__(self, params, lr, max_norm=None, betas=(0.9, 0.999), eps=1e-8, big_bytes=16 << 20, capturable=False):
        params = [p for p in params if p.requires_grad]
        self.lr, self.betas, self.eps, self.max_norm = float(lr), (float(betas[0]), float(betas[1])), float(eps), max_norm
        self.big = [p for p in params if p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
                    and p.numel() * 4 >= big_bytes and p.data_ptr() % 16 == 0]
        ids = {id(p) for p in self.big}
        self.small = [p for p in params if id(p) not in ids]
        # capturable: every step count lives on the device, so that a step recorded into a hipGraph (hotpath.CapturedTraining)
        # replays with the right bias corrections; same arithmetic either way
        self.capturable = bool(capturable)
        self.small_opt = torch.optim.Adam(self.small, lr=lr, betas=betas, eps=eps, capturable=self.capturable,
                                          fused=all(p.is_cuda for p in self.small)) if self.small else None
        self.state = {id(p): {'step': 0, 'exp_avg': torch.zeros_like(p), 'exp_avg_sq': torch.zeros_like(p),
                              'step_dev': torch.zeros(1, dtype=torch.int64, device=p.device) if self.capturable else None}
                      for p in self.big}

    def step(self):
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
        for p in self.big:
            if p.grad is not None:
                if set_to_none:
                    p.grad = None
                else:
                    p.grad.zero_()
