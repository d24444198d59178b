"""Parameter containers for the SWEM encoders / decoder.

These classes exist so that ``SWEM.state_dict()`` has exactly the reference's keys
(reference methods/basic_modules/networks.py:12-216,
mod_resnet.py:45-152, attentions.py:22-84) and reference checkpoints load with
``strict=True``.  They hold parameters only: none of them has a torch ``forward``.
The arithmetic lives in ``swem_amd/engine.py``, which walks these containers and
launches the HIP kernels of ``libswem_hip.so`` on packed copies of the weights.
"""
import torch
from torch import nn


def _conv(cin, cout, k, stride=1, bias=True):
    return nn.Conv2d(cin, cout, kernel_size=k, stride=stride, padding=k // 2, bias=bias)


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover - containers are never called
        raise RuntimeError('parameter container: the HIP engine runs this block (swem_amd/engine.py)')


class BasicBlock(_Holder):
    """mod_resnet.py:45-74 (bias=True) / torchvision BasicBlock (bias=False)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride, bias):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 3, stride, bias)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, 1, bias)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(_conv(inplanes, planes, 1, stride, bias), nn.BatchNorm2d(planes))
        self.stride = stride


class Bottleneck(_Holder):
    """mod_resnet.py:77-113 / torchvision v1.5 Bottleneck: the stride sits on the 3x3."""
    expansion = 4

    def __init__(self, inplanes, planes, stride, bias):
        super().__init__()
        self.conv1 = _conv(inplanes, planes, 1, 1, bias)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _conv(planes, planes, 3, stride, bias)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _conv(planes, planes * 4, 1, 1, bias)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.downsample = None
        if stride != 1 or inplanes != planes * 4:
            self.downsample = nn.Sequential(_conv(inplanes, planes * 4, 1, stride, bias),
                                            nn.BatchNorm2d(planes * 4))
        self.stride = stride


def _make_stage(block, inplanes, planes, nblocks, stride, bias):
    layers = [block(inplanes, planes, stride, bias)]
    inplanes = planes * block.expansion
    for _ in range(1, nblocks):
        layers.append(block(inplanes, planes, 1, bias))
    return nn.Sequential(*layers), inplanes


BACKBONES = {'resnet18': (BasicBlock, (2, 2, 2)), 'resnet50': (Bottleneck, (3, 4, 6))}


def _imagenet_stats(mod):
    mod.register_buffer('mean', torch.FloatTensor([0.485, 0.456, 0.406]).view(1, 3, 1, 1))
    mod.register_buffer('std', torch.FloatTensor([0.229, 0.224, 0.225]).view(1, 3, 1, 1))


class KeyEncoder(_Holder):
    """networks.py:132-170.  The trunk is torchvision's ResNet (bias-free convs)."""

    def __init__(self, backbone_name='resnet50'):
        super().__init__()
        if backbone_name not in BACKBONES:
            raise KeyError('The backbone {} is not supported yet.'.format(backbone_name))
        block, nb = BACKBONES[backbone_name]
        self.backbone_name = backbone_name
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.res2, c = _make_stage(block, 64, 64, nb[0], 1, False)
        self.layer2, c = _make_stage(block, c, 128, nb[1], 2, False)
        self.layer3, c = _make_stage(block, c, 256, nb[2], 2, False)
        self.num_features = [c, c // 2, c // 4]
        _imagenet_stats(self)


class ResBlock(_Holder):
    """networks.py:12-32."""

    def __init__(self, indim, outdim=None):
        super().__init__()
        outdim = indim if outdim is None else outdim
        self.downsample = None if indim == outdim else _conv(indim, outdim, 3)
        self.conv1 = _conv(indim, outdim, 3)
        self.conv2 = _conv(outdim, outdim, 3)


class _Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


class ChannelGate(_Holder):
    """attentions.py:22-50; keys ``mlp.1`` and ``mlp.3``."""

    def __init__(self, gate_channels, reduction_ratio=16):
        super().__init__()
        self.mlp = nn.Sequential(_Flatten(), nn.Linear(gate_channels, gate_channels // reduction_ratio),
                                 nn.ReLU(), nn.Linear(gate_channels // reduction_ratio, gate_channels))


class _BasicConv(_Holder):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(2, 1, kernel_size=7, stride=1, padding=3)


class SpatialGate(_Holder):
    """attentions.py:58-69; key ``spatial.conv``."""

    def __init__(self):
        super().__init__()
        self.spatial = _BasicConv()


class CBAM(_Holder):
    """attentions.py:72-84."""

    def __init__(self, gate_channels):
        super().__init__()
        self.ChannelGate = ChannelGate(gate_channels)
        self.SpatialGate = SpatialGate()


class FeatureFusionBlock(_Holder):
    """networks.py:35-50."""

    def __init__(self, indim, outdim):
        super().__init__()
        self.block1 = ResBlock(indim, outdim)
        self.attention = CBAM(outdim)
        self.block2 = ResBlock(outdim, outdim)


class ValueEncoder(_Holder):
    """networks.py:94-129 (extra_chan=2) and :56-90 (single object, extra_chan=1).
    The trunk is the reference's own mod_resnet.resnet18: every conv has a bias."""

    def __init__(self, in_dim=1024, extra_chan=2):
        super().__init__()
        self.conv1 = nn.Conv2d(3 + extra_chan, 64, kernel_size=7, stride=2, padding=3)
        self.bn1 = nn.BatchNorm2d(64)
        self.layer1, c = _make_stage(BasicBlock, 64, 64, 2, 1, True)
        self.layer2, c = _make_stage(BasicBlock, c, 128, 2, 2, True)
        self.layer3, c = _make_stage(BasicBlock, c, 256, 2, 2, True)
        self.fuser = FeatureFusionBlock(in_dim + 256, 512)
        _imagenet_stats(self)


class ValueEncoderSO(ValueEncoder):
    def __init__(self, in_dim=1024):
        super().__init__(in_dim, extra_chan=1)


class KeyProjection(_Holder):
    """networks.py:173-182."""

    def __init__(self, indim, keydim):
        super().__init__()
        self.key_proj = _conv(indim, keydim, 3)
        nn.init.orthogonal_(self.key_proj.weight.data)
        nn.init.zeros_(self.key_proj.bias.data)


class UpsampleBlock(_Holder):
    """networks.py:186-196."""

    def __init__(self, skip_c, up_c, out_c):
        super().__init__()
        self.skip_conv = _conv(skip_c, up_c, 3)
        self.out_conv = ResBlock(up_c, out_c)


class Decoder(_Holder):
    """networks.py:199-216."""

    def __init__(self, inplanes, mdim=256):
        super().__init__()
        self.compress = ResBlock(inplanes[0], 512)
        self.up_16_8 = UpsampleBlock(inplanes[1], 512, mdim)
        self.up_8_4 = UpsampleBlock(inplanes[2], 256, mdim)
        self.pred = _conv(mdim, 1, 3)
