"""Ray-drop refinement U-Net (reference model/unet.py:139-170, used by
model/runner.py:413-416 on the stacked ``[raydrop, intensity, depth]`` range image).

Dense 2-D convolutions on a 64 x 2048 image: this is library work, so it runs on MIOpen / rocBLAS through
PyTorch-ROCm as the scope table prescribes ("custom kernels only if it shows up in the profile" -- it is
< 2 % of a frame's render time).  Module names are the reference's so that checkpoints load
(`inc.conv`, `down{1..4}.conv.double_conv.{0,3,4,7}`, `attn.{norm,proj_qkv,proj}`, `up{1..4}.conv.double_conv`,
`outc.conv.{0,2}`).  Input and output stay channels-first like the reference; internally tensors are kept in
``channels_last`` memory format, which is what MIOpen's NHWC kernels want on gfx950.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


def _pre_act_pair(c_in, c_out, c_mid=None, p_drop=0.1):
    """(BatchNorm -> ReLU -> Dropout2d -> 3x3 conv) twice; indices 0..7 match unet.py:22-31."""
    c_mid = c_mid or c_out
    layers = []
    for a, b in ((c_in, c_mid), (c_mid, c_out)):
        layers += [nn.BatchNorm2d(a), nn.ReLU(inplace=True), nn.Dropout2d(p_drop),
                   nn.Conv2d(a, b, kernel_size=3, padding=1, bias=False)]
    return nn.Sequential(*layers)


class _Block(nn.Module):
    def __init__(self, c_in, c_out, c_mid=None):
        super().__init__()
        self.double_conv = _pre_act_pair(c_in, c_out, c_mid)

    def forward(self, x):
        return self.double_conv(x)


class _Encode(nn.Module):
    """2x2 max-pool, then the conv pair (unet.py:37-50)."""

    def __init__(self, c_in, c_out):
        super().__init__()
        self.down = nn.MaxPool2d(2)
        self.conv = _Block(c_in, c_out)

    def forward(self, x):
        return self.conv(self.down(x))


class _Decode(nn.Module):
    """Bilinear x2 (align_corners), centre-pad to the skip's size, concat [skip, up], conv pair (unet.py:53-73)."""

    def __init__(self, c_in, c_out):
        super().__init__()
        self.up = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True)
        self.conv = _Block(c_in, c_out, c_in)

    def forward(self, low, skip):
        low = self.up(low)
        dh, dw = skip.shape[2] - low.shape[2], skip.shape[3] - low.shape[3]
        if dh or dw:
            low = F.pad(low, [dw // 2, dw - dw // 2, dh // 2, dh - dh // 2])
        return self.conv(torch.cat([skip, low], dim=1))


class _Attention(nn.Module):
    """8-head self-attention over the bottleneck pixels (unet.py:76-111).  The reference re-interprets the
    attention result ``[B, heads, H*W, C/heads]`` as ``[B, H, W, C]`` by a plain view (not a head transpose);
    a checkpoint is trained with exactly that wiring, so it is kept."""

    def __init__(self, channels, num_head=8, dropout=0.1):
        super().__init__()
        self.proj_qkv = nn.Conv2d(channels, 3 * channels, 1, bias=False)
        self.proj = nn.Conv2d(channels, channels, 1, bias=False)
        self.norm = nn.BatchNorm2d(channels)
        self.dropout = dropout
        self.num_head = num_head

    def forward(self, x):
        B, C, H, W = x.shape
        q, k, v = self.proj_qkv(self.norm(x)).contiguous().chunk(3, dim=1)
        heads, n = self.num_head, H * W
        q = q.reshape(B, heads, C // heads, n).transpose(2, 3)
        k = k.reshape(B, heads, C // heads, n)
        v = v.reshape(B, heads, C // heads, n).transpose(2, 3)
        logits = torch.matmul(q, k) * (int(C // heads) ** -0.5)
        if self.training:  # attention-logit dropout, unet.py:101-103
            logits = logits + torch.bernoulli(torch.full_like(logits, self.dropout)) * -1e12
        mixed = torch.matmul(torch.softmax(logits, dim=-1), v)
        mixed = mixed.contiguous().view(B, H, W, C).permute(0, 3, 1, 2)
        return x + self.proj(mixed)


class _Stem(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.conv = nn.Conv2d(c_in, c_out, kernel_size=1)

    def forward(self, x):
        return self.conv(x)


class _Head(nn.Module):
    def __init__(self, c_in, c_out):
        super().__init__()
        self.conv = nn.Sequential(nn.BatchNorm2d(c_in), nn.ReLU(inplace=True), nn.Conv2d(c_in, c_out, kernel_size=1))

    def forward(self, x):
        return self.conv(x)


class UNet(nn.Module):
    """``UNet(in_channels=3, channels=32, out_channels=1)``: ``[B, 3, H, W] -> sigmoid ray-drop [B, 1, H, W]``."""

    def __init__(self, in_channels, channels=32, out_channels=1):
        super().__init__()
        c = channels
        self.inc = _Stem(in_channels, c)
        self.down1, self.down2 = _Encode(c, 2 * c), _Encode(2 * c, 4 * c)
        self.down3, self.down4 = _Encode(4 * c, 8 * c), _Encode(8 * c, 8 * c)
        self.attn = _Attention(8 * c)
        self.up1, self.up2 = _Decode(16 * c, 4 * c), _Decode(8 * c, 2 * c)
        self.up3, self.up4 = _Decode(4 * c, c), _Decode(2 * c, c)
        self.outc = _Head(c, out_channels)
        self.sigmoid = nn.Sigmoid()

    def forward(self, input):
        x0 = self.inc(input.contiguous(memory_format=torch.channels_last) if input.is_cuda else input)
        x1 = self.down1(x0)
        x2 = self.down2(x1)
        x3 = self.down3(x2)
        x4 = self.attn(self.down4(x3))
        y = self.up1(x4, x3)
        y = self.up2(y, x2)
        y = self.up3(y, x1)
        y = self.up4(y, x0)
        return self.sigmoid(self.outc(y)).contiguous()
