"""`UNet` -- model1 of the Semi-Mamba-UNet configuration (BASELINE.json configs[3]).

The reference pairs a plain convolutional U-Net (code/networks/unet.py:304-321, chosen at
train_Semi_Mamba_UNet.py:302-303) with MambaUnet for cross pseudo supervision.  It is a caller beside the SS2D hot
path, not part of it: convolutions, BatchNorm (per-replica statistics, as under the reference's DataParallel),
LeakyReLU, Dropout, max-pooling and bilinear upsampling run on MIOpen / torch kernels.  What this file keeps is
the contract: constructor `UNet(in_chns, class_num)`, widths 16-32-64-128-256, dropout 0.05-0.5 in the encoder and 0
in the decoder, 1x1 convolution + bilinear upsampling in the decoder, and the state-dict keys of the reference
(`encoder.in_conv.conv_conv.0.weight`, `encoder.down1.maxpool_conv.1.conv_conv.*`, `decoder.up1.conv1x1.weight`,
`decoder.up1.conv.conv_conv.*`, `decoder.out_conv.*`), so its checkpoints load unchanged."""
from __future__ import annotations

import torch
import torch.nn as nn

FEATURES = (16, 32, 64, 128, 256)
ENCODER_DROPOUT = (0.05, 0.1, 0.2, 0.3, 0.5)


def _double_conv(cin: int, cout: int, p: float) -> nn.Sequential:
    """conv3x3-BN-LeakyReLU-Dropout-conv3x3-BN-LeakyReLU (unet.py:31-47); indices 0,1,4,5 carry parameters."""
    return nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.LeakyReLU(), nn.Dropout(p),
                         nn.Conv2d(cout, cout, 3, padding=1), nn.BatchNorm2d(cout), nn.LeakyReLU())


class _Block(nn.Module):
    def __init__(self, cin, cout, p):
        super().__init__()
        self.conv_conv = _double_conv(cin, cout, p)

    def forward(self, x):
        return self.conv_conv(x)


class _Down(nn.Module):
    def __init__(self, cin, cout, p):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), _Block(cin, cout, p))

    def forward(self, x):
        return self.maxpool_conv(x)


class _Up(nn.Module):
    def __init__(self, cin, cskip, cout):
        super().__init__()
        # the reference's Decoder never forwards its `bilinear: False` parameter, so UpBlock's default applies (unet.py:68-76,
        # :129-136): 1x1 convolution to the skip width, then bilinear x2 upsampling with align_corners
        self.conv1x1 = nn.Conv2d(cin, cskip, kernel_size=1)
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.conv = _Block(2 * cskip, cout, 0.0)

    def forward(self, x, skip):
        return self.conv(torch.cat([skip, self.up(self.conv1x1(x))], dim=1))          # skip first (unet.py:84)


class _Encoder(nn.Module):
    def __init__(self, in_chns):
        super().__init__()
        f, p = FEATURES, ENCODER_DROPOUT
        self.in_conv = _Block(in_chns, f[0], p[0])
        for i in range(1, 5):
            setattr(self, f"down{i}", _Down(f[i - 1], f[i], p[i]))

    def forward(self, x):
        feats = [self.in_conv(x)]
        for i in range(1, 5):
            feats.append(getattr(self, f"down{i}")(feats[-1]))
        return feats


class _Decoder(nn.Module):
    def __init__(self, class_num):
        super().__init__()
        f = FEATURES
        for i in range(1, 5):
            setattr(self, f"up{i}", _Up(f[5 - i], f[4 - i], f[4 - i]))
        self.out_conv = nn.Conv2d(f[0], class_num, kernel_size=3, padding=1)

    def forward(self, feats):
        x = feats[4]
        for i in range(1, 5):
            x = getattr(self, f"up{i}")(x, feats[4 - i])
        return self.out_conv(x)


class UNet(nn.Module):
    def __init__(self, in_chns: int, class_num: int):
        super().__init__()
        self.encoder = _Encoder(in_chns)
        self.decoder = _Decoder(class_num)

    # On the GPU the network runs channel-last (NHWC): MIOpen's bf16 / fp32 implicit-GEMM convolutions are NHWC kernels, and
    # with NCHW tensors every convolution is wrapped in a pair of transposing copies (Semi-Mamba-UNet step at batch 24:
    # 31.5 -> 25.0 ms).  Parameters keep their identity, shapes and state-dict keys (only their strides change, once, at the
    # first GPU forward); the logits are returned NCHW-contiguous as the reference's are.
    channels_last = True

    def native_layout(self):
        """Convert the parameters to channel-last now (call it before wrapping the model in DDP, whose buckets record the
        gradients' strides); forward() does it at the first GPU batch otherwise.  Returns self."""
        if self.channels_last and not getattr(self, "_nhwc", False):
            self.to(memory_format=torch.channels_last)
            self._nhwc = True
        return self

    def forward(self, x):
        if x.is_cuda and self.channels_last:
            self.native_layout()
            x = x.contiguous(memory_format=torch.channels_last)
            return self.decoder(self.encoder(x)).contiguous()
        return self.decoder(self.encoder(x))
