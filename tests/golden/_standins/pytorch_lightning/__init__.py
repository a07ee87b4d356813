import torch


class LightningModule(torch.nn.Module):
    """PL 0.7.1-era base: plain attributes ``device`` / ``hparams`` are assignable."""
    pass
