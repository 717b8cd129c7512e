import torch


class Normal(object):
    """paddle.nn.initializer.Normal(mean=0.0, std=1.0)"""

    def __init__(self, mean=0.0, std=1.0):
        self.mean, self.std = mean, std

    def __call__(self, tensor):
        torch.nn.init.normal_(tensor, self.mean, self.std)
