"""UNet depth regressors (mirror of the reference's model/unet.py:15-118 `Unet`, :121-186 `UNetMini`).

This is the stage BEFORE the hot path in BASELINE config 5.  It is not a
hand-kernel target of the north star: the layers are stock PyTorch-ROCm ops (MIOpen Conv2d,
BatchNorm2d, bilinear upsample).  Parameter names and shapes match the reference so `unet.*`
checkpoint entries load (trainer/trainer_scene_net.py:204-212).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _UNetBase(nn.Module):
    # encoder widths as multiples of num_filters; decoder: (name, in mult, out mult or None=channels_out, bn name)
    ENC = ()
    ENC_BN = ()
    DEC = ()

    def __init__(self, num_filters=32, channels_in=3, channels_out=3):
        super().__init__()
        nf = num_filters
        c_prev = channels_in
        for i, mult in enumerate(self.ENC, start=1):
            setattr(self, f"conv{i}", nn.Conv2d(c_prev, nf * mult, 4, 2, 1))
            c_prev = nf * mult
        self.up = nn.Upsample(scale_factor=2, mode="bilinear")
        for name, cin, cout, _ in self.DEC:
            setattr(self, name, nn.Conv2d(nf * cin, channels_out if cout is None else nf * cout, 3, 1, 1))
        for name, mult in self.BN:
            setattr(self, name, nn.BatchNorm2d(nf * mult))
        self.leaky_relu = nn.LeakyReLU(0.2)
        self.relu = nn.ReLU()

    def forward(self, input):
        skips = []
        x = input
        for i in range(1, len(self.ENC) + 1):
            if i > 1:
                x = self.leaky_relu(x)
            x = getattr(self, f"conv{i}")(x)
            bn = self.ENC_BN[i - 1]
            if bn is not None:
                x = getattr(self, bn)(x)
            skips.append(x)
        d = skips.pop()
        for name, _, _, bn in self.DEC:
            d = getattr(self, name)(self.up(self.relu(d)))
            if bn is not None:
                d = torch.cat((getattr(self, bn)(d), skips.pop()), 1)
        return d


class Unet(_UNetBase):
    """8-down / 8-up regressor for 256x256 (or multiples) inputs."""
    ENC = (1, 2, 4, 8, 8, 8, 8, 8)
    ENC_BN = (None, "batch_norm2_0", "batch_norm4_0", "batch_norm8_0", "batch_norm8_1", "batch_norm8_2", "batch_norm8_3", None)
    DEC = (("dconv1", 8, 8, "batch_norm8_4"), ("dconv2", 16, 8, "batch_norm8_5"), ("dconv3", 16, 8, "batch_norm8_6"),
           ("dconv4", 16, 8, "batch_norm8_7"), ("dconv5", 16, 4, "batch_norm4_1"), ("dconv6", 8, 2, "batch_norm2_1"),
           ("dconv7", 4, 1, "batch_norm"), ("dconv8", 2, None, None))
    BN = (("batch_norm", 1), ("batch_norm2_0", 2), ("batch_norm2_1", 2), ("batch_norm4_0", 4), ("batch_norm4_1", 4),
          ("batch_norm8_0", 8), ("batch_norm8_1", 8), ("batch_norm8_2", 8), ("batch_norm8_3", 8), ("batch_norm8_4", 8),
          ("batch_norm8_5", 8), ("batch_norm8_6", 8), ("batch_norm8_7", 8))


class UNetMini(_UNetBase):
    """4-down / 4-up variant for un-resized 240x320 inputs."""
    ENC = (1, 2, 4, 8)
    ENC_BN = (None, "batch_norm2_0", "batch_norm4_0", None)
    DEC = (("dconv5", 8, 4, "batch_norm4_1"), ("dconv6", 8, 2, "batch_norm2_1"), ("dconv7", 4, 1, "batch_norm"),
           ("dconv8", 2, None, None))
    BN = (("batch_norm", 1), ("batch_norm2_0", 2), ("batch_norm2_1", 2), ("batch_norm4_0", 4), ("batch_norm4_1", 4))
