"""The two helpers of core/setup.py the training step depends on; the rest of that
file (result dirs, logging, checkpoint restore) is control plane and out of scope."""
import random

import numpy as np
import torch


def seed_setup(seed: int = 0):
    """core/setup.py:12-19"""
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def weights_init(m):
    """core/setup.py:63-77: xavier_uniform(gain=sqrt(2)) on every Conv/Linear weight, zero bias
    (the BatchNorm branch of the reference is dead: the model has none)."""
    classname = m.__class__.__name__
    if classname.find('Conv') != -1 or classname.find('Linear') != -1:
        gain = torch.nn.init.calculate_gain('relu')
        torch.nn.init.xavier_uniform_(m.weight, gain)
        if m.bias is not None:
            torch.nn.init.constant_(m.bias, 0)
    elif classname.find('BatchNorm') != -1:
        torch.nn.init.constant_(m.weight, 1)
        torch.nn.init.constant_(m.bias, 0)
