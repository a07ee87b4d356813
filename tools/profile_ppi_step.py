"""cProfile of the batch-64 training step on the PPI-BP stand-in (host-side overhead hunt)."""
import cProfile, pstats, io, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from subgnn_amd import standins as B
root = tempfile.mkdtemp(prefix='ppi_bp_')
model, d, _ = B.build_model(root, 'ppi_bp')
opt = model.configure_optimizers()
model.train()
loader = model.train_dataloader()
import itertools
def gen():
    while True:
        for b in model.train_dataloader():
            yield b
it = gen()
batches = None
def step(batch):
    out = model.training_step(batch, 0)
    opt.zero_grad(set_to_none=True)
    model.backward(None, out['loss'], opt, 0)
    torch.nn.utils.clip_grad_norm_(model.parameters(), B.H2['grad_clip'])
    opt.step()
for _ in range(5): step(next(it))
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
t = time.perf_counter()
for _ in range(30): step(next(it))
torch.cuda.synchronize()
el = time.perf_counter() - t
pr.disable()
print('ms/step', el / 30 * 1e3)
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue()[:6000])
