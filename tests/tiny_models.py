"""Small seeded classifiers shared by the tests (the same construction tests/golden/make_golden.py used)."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Args:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class TinyNet(nn.Module):
    """conv3x3(8) -> ReLU -> avgpool2 -> linear; ReLU gives exact-zero gradients so sign(0) = 0 is exercised."""

    def __init__(self, cin, hw, ncls, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w1 = nn.Parameter(torch.randn(8, cin, 3, 3, generator=g) * 0.5)
        self.w2 = nn.Parameter(torch.randn(ncls, 8 * (hw // 2) * (hw // 2), generator=g) * 0.2)

    def forward(self, x):
        h = F.relu(F.conv2d(x, self.w1, padding=1))
        h = F.avg_pool2d(h, 2)
        return F.linear(h.flatten(1), self.w2)


class TinyBNNet(nn.Module):
    """TinyNet with a BatchNorm between the convolution and the ReLU (the free-AT fixtures: the loop stays in train mode)."""

    def __init__(self, cin, hw, ncls, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w1 = nn.Parameter(torch.randn(8, cin, 3, 3, generator=g) * 0.5)
        self.bn = nn.BatchNorm2d(8)
        self.w2 = nn.Parameter(torch.randn(ncls, 8 * (hw // 2) * (hw // 2), generator=g) * 0.2)

    def forward(self, x):
        h = F.relu(self.bn(F.conv2d(x, self.w1, padding=1)))
        h = F.avg_pool2d(h, 2)
        return F.linear(h.flatten(1), self.w2)


class TinySegNet(nn.Module):
    """Three segments with the interface eeadv.trainer._GraphedUpdate cuts at (segment_fns / grad_segments, as models.ResNet has them):
    conv -> [pair of tensors over one value] -> conv + identity -> linear; plus a parameter that no forward uses (its .grad must stay
    None under eeadv.ddp.FlatGradSync, as under DistributedDataParallel)."""

    def __init__(self, cin, hw, ncls, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.w1 = nn.Parameter(torch.randn(6, cin, 3, 3, generator=g) * 0.5)
        self.w2 = nn.Parameter(torch.randn(6, 6, 3, 3, generator=g) * 0.3)
        self.w3 = nn.Parameter(torch.randn(ncls, 6 * (hw // 2) * (hw // 2), generator=g) * 0.2)
        self.unused = nn.Parameter(torch.randn(5, generator=g))

    def segment_fns(self):
        def first(x):
            h = F.relu(F.conv2d(x, self.w1, padding=1))
            return h, h  # a forked output: one tensor for the main branch, one for the identity branch
        def middle(x):
            xm, xs = x
            return F.relu(F.conv2d(xm, self.w2, padding=1) + xs)
        def last(x):
            return F.linear(F.avg_pool2d(x, 2).flatten(1), self.w3)
        return [first, middle, last]

    def grad_segments(self):
        return [[self.w3, self.unused], [self.w2], [self.w1]]

    def forward(self, x):
        for fn in self.segment_fns():
            x = fn(x)
        return x


class TinyModuleNet(nn.Module):
    """conv (with bias) -> BatchNorm -> ReLU -> avgpool2 -> linear, built from nn modules so that the state_dict keys read
    `conv.weight`, `bn.weight`, `fc.weight` ...: AWP (utils_awp.py:8-18) perturbs the entries named '*weight*' with more than one dimension."""

    def __init__(self, cin, hw, ncls, seed):
        super().__init__()
        torch.manual_seed(seed)
        self.conv = nn.Conv2d(cin, 8, 3, padding=1)
        self.bn = nn.BatchNorm2d(8)
        self.fc = nn.Linear(8 * (hw // 2) * (hw // 2), ncls)

    def forward(self, x):
        h = F.relu(self.bn(self.conv(x)))
        return self.fc(F.avg_pool2d(h, 2).flatten(1))
