"""ResNet-18 image encoders of SHERF (`ResNet18Classifier`, triplane.py:320-343): the 512-d global code that replaces z
(`encoder_2d`, triplane.py:75) and the 64-channel half-resolution feature map the pixel-aligned taps read
(`encoder_2d_feature`, extract_feature=True).  SURVEY.md section 8(f) rank 2: a per-frame PRODUCER, plain library convolutions
(MIOpen through PyTorch-ROCm), no custom kernel.

The network is the standard ResNet-18 (He et al. 2016) with torchvision's module names, so that a torchvision / SHERF checkpoint
loads with strict=True: `backbone.{conv1, bn1, layer1..4.{0,1}.{conv1, bn1, conv2, bn2, downsample.{0,1}}, fc}`.  torchvision is
not installed here: the ImageNet weights the reference downloads (`resnet18(pretrained=True)`) have to come from the SHERF
checkpoint; a fresh module is randomly initialised."""
import torch
from torch import nn


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = None
        if stride != 1 or inplanes != planes:
            self.downsample = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        return self.relu(self.bn2(self.conv2(y)) + idt)


class ResNet18(nn.Module):
    def __init__(self, num_classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = nn.Sequential(BasicBlock(64, 64), BasicBlock(64, 64))
        self.layer2 = nn.Sequential(BasicBlock(64, 128, 2), BasicBlock(128, 128))
        self.layer3 = nn.Sequential(BasicBlock(128, 256, 2), BasicBlock(256, 256))
        self.layer4 = nn.Sequential(BasicBlock(256, 512, 2), BasicBlock(512, 512))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, num_classes)
        for m in self.modules():                                   # torchvision's initialisation
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


class ResNet18Classifier(nn.Module):
    """triplane.py:320-343.  forward(x) -> [N, 512] pooled code (the classifier head `fc` is never applied);
    forward(x, extract_feature=True) -> layer1's map [N, 64, H/2, W/2] with the max-pool skipped."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        self.backbone = ResNet18()

    def forward(self, x, extract_feature=False):
        b = self.backbone
        x = b.relu(b.bn1(b.conv1(x)))
        if not extract_feature:
            x = b.maxpool(x)
        x = b.layer1(x)
        if extract_feature:
            return x
        x = b.layer4(b.layer3(b.layer2(x)))
        return torch.flatten(b.avgpool(x), 1)
