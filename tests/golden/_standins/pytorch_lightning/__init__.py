import torch


class LightningModule(torch.nn.Module):
    """PL 0.7.1-era base: plain attributes ``device`` / ``hparams`` are assignable."""
    pass


# import-only names of SubGNN/train_config.py (the trainer itself is never built by the golden harness)
import sys as _sys
import types as _types

for _name, _attrs in (('loggers', ('TensorBoardLogger',)), ('callbacks', ('ModelCheckpoint',)), ('profiler', ('AdvancedProfiler',))):
    _m = _types.ModuleType('pytorch_lightning.' + _name)
    for _a in _attrs:
        setattr(_m, _a, object)
    _sys.modules['pytorch_lightning.' + _name] = _m
    globals()[_name] = _m
Trainer = object
