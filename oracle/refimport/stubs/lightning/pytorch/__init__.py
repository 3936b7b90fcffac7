"""Scaffolding stub for lightning.pytorch (import-only)."""
import torch.nn as nn


class LightningModule(nn.Module):
    global_step = 0

    def save_hyperparameters(self, *a, **k):
        pass

    def log(self, *a, **k):
        pass


class LightningDataModule:
    pass


class Callback:
    pass
