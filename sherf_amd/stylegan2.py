"""StyleGAN2 generator (mapping + synthesis) -- the tri-plane PRODUCER of SHERF (`TriPlaneGenerator.backbone`, triplane.py:58) --
as a drop-in for `training.networks_stylegan2.Generator` (networks_stylegan2.py:538-561): same constructor arguments, the same
parameter / buffer names and shapes (checkpoint contract: `copy_params_and_buffers(require_all=True)`, training_loop.py:207-208)
and the same arithmetic, on PyTorch-ROCm's library convolutions plus the two HIP operators of libsherf_hip_ops.so
(`sherf_amd.bias_act`, `sherf_amd.upfirdn2d`).  SURVEY.md section 8(f) rank 2; verified against the unmodified reference on the CPU and on the MI355X
(tests/test_gpu_producers.py).

Structure (reference lines in the docstrings): a resolution pyramid 4 -> img_resolution of SynthesisBlocks, each two modulated 3x3
convolutions (the first one x2 up-sampling: transposed convolution, then the [1,3,3,1] FIR with gain 4) and a 1x1 ToRGB whose
outputs are summed over the pyramid ('skip').  Modulated convolution: in training the activations are scaled before and after a
plain convolution; at inference the per-sample weights are built once and the batch runs as one grouped convolution
(`fused_modconv_default='inference_only'`, train.py:312)."""
import math

import torch

from . import bias_act as _ba
from . import upfirdn2d as _uf

OPS_IMPL = 'cuda'       # implementation name handed to bias_act / upfirdn2d ('cuda' = the HIP kernels; tests may select 'ref' explicitly)


def _bias_act(x, b=None, **kw):
    return _ba.bias_act(x, b, impl=OPS_IMPL, **kw)


def normalize_2nd_moment(x, dim=1, eps=1e-8):
    """networks_stylegan2.py:28-29"""
    return x * torch.rsqrt(x.square().mean(dim=dim, keepdim=True) + eps)


def conv2d_resample(x, w, f=None, up=1, down=1, padding=0, groups=1, flip_weight=True, flip_filter=False):
    """Convolution with optional x`up` up-sampling before and x`down` down-sampling after, both low-pass filtered by `f`
    (torch_utils/ops/conv2d_resample.py:45-143).  `padding` is relative to the up-sampled image.  flip_weight=True = correlation
    (torch's conv2d), False = true convolution.  Every branch evaluates a window of the same linear operator
    FIR_f o conv_w o zero_stuff; the factorisation is chosen for cost: up-sampling as a stride-`up` transposed convolution (skips
    the stuffed zeros) followed by the FIR, down-sampling as the FIR followed by a strided convolution."""
    out_ch, in_ch_g, kh, kw = w.shape
    fw, fh = _uf._filter_size(f)
    px0, px1, py0, py1 = _uf._padding(padding)
    if up > 1:                                      # conv2d_resample.py:80-84
        px0 += (fw + up - 1) // 2; px1 += (fw - up) // 2; py0 += (fh + up - 1) // 2; py1 += (fh - up) // 2
    if down > 1:                                    # :85-89
        px0 += (fw - down + 1) // 2; px1 += (fw - down) // 2; py0 += (fh - down + 1) // 2; py1 += (fh - down) // 2

    def conv(t, weight, stride=1, pad=(0, 0), transpose=False, flip=True):
        # torch's conv2d correlates and its conv_transpose2d convolves: flip the kernel when the other one is asked for
        if (not flip) != transpose and (kh > 1 or kw > 1):
            weight = weight.flip([2, 3])
        fn = torch.nn.functional.conv_transpose2d if transpose else torch.nn.functional.conv2d
        return fn(t, weight, stride=stride, padding=pad, groups=groups)

    if kh == 1 and kw == 1 and down > 1 and up == 1:                      # 1x1: filter + decimate first, then mix channels
        x = _uf.upfirdn2d(x, f, down=down, padding=[px0, px1, py0, py1], flip_filter=flip_filter, impl=OPS_IMPL)
        return conv(x, w, flip=flip_weight)
    if kh == 1 and kw == 1 and up > 1 and down == 1:                      # 1x1: mix channels at low resolution, then up-sample
        x = conv(x, w, flip=flip_weight)
        return _uf.upfirdn2d(x, f, up=up, padding=[px0, px1, py0, py1], gain=up ** 2, flip_filter=flip_filter, impl=OPS_IMPL)
    if down > 1 and up == 1:                                              # blur, then strided convolution
        x = _uf.upfirdn2d(x, f, padding=[px0, px1, py0, py1], flip_filter=flip_filter, impl=OPS_IMPL)
        return conv(x, w, stride=down, flip=flip_weight)
    if up > 1:                                                            # transposed convolution, then the FIR (gain up^2)
        if groups == 1:
            wt = w.transpose(0, 1)
        else:
            wt = w.reshape(groups, out_ch // groups, in_ch_g, kh, kw).transpose(1, 2).reshape(groups * in_ch_g, out_ch // groups, kh, kw)
        px0 -= kw - 1; px1 -= kw - up; py0 -= kh - 1; py1 -= kh - up      # what the full transposed convolution already grew
        pxt, pyt = max(min(-px0, -px1), 0), max(min(-py0, -py1), 0)
        x = conv(x, wt, stride=up, pad=(pyt, pxt), transpose=True, flip=flip_weight)
        x = _uf.upfirdn2d(x, f, padding=[px0 + pxt, px1 + pxt, py0 + pyt, py1 + pyt], gain=up ** 2, flip_filter=flip_filter, impl=OPS_IMPL)
        if down > 1:
            x = _uf.upfirdn2d(x, f, down=down, flip_filter=flip_filter, impl=OPS_IMPL)
        return x
    if px0 == px1 and py0 == py1 and px0 >= 0 and py0 >= 0:               # plain convolution
        return conv(x, w, pad=(py0, px0), flip=flip_weight)
    x = _uf.upfirdn2d(x, None, padding=[px0, px1, py0, py1], impl=OPS_IMPL)  # asymmetric padding / cropping, then convolution
    return conv(x, w, flip=flip_weight)


def modulated_conv2d(x, weight, styles, noise=None, up=1, down=1, padding=0, resample_filter=None, demodulate=True, flip_weight=True,
                     fused_modconv=True):
    """Per-sample modulated (and demodulated) convolution, networks_stylegan2.py:34-92.  x [N,I,H,W], weight [O,I,k,k], styles [N,I]."""
    N = x.shape[0]
    O, I, kh, kw = weight.shape
    if x.dtype == torch.float16 and demodulate:     # pre-normalise against fp16 overflow (:57-60)
        weight = weight * (1 / math.sqrt(I * kh * kw) / weight.norm(float('inf'), dim=[1, 2, 3], keepdim=True))
        styles = styles / styles.norm(float('inf'), dim=1, keepdim=True)
    wmod = dcoefs = None
    if demodulate or fused_modconv:
        wmod = weight.unsqueeze(0) * styles.reshape(N, 1, I, 1, 1)                       # [N,O,I,k,k]
    if demodulate:
        dcoefs = torch.rsqrt(wmod.square().sum(dim=[2, 3, 4]) + 1e-8)                    # [N,O]
    if not fused_modconv:                            # scale activations around ONE shared-weight convolution (:73-83)
        x = x * styles.to(x.dtype).reshape(N, I, 1, 1)
        x = conv2d_resample(x, weight.to(x.dtype), f=resample_filter, up=up, down=down, padding=padding, flip_weight=flip_weight)
        if demodulate:
            x = x * dcoefs.to(x.dtype).reshape(N, O, 1, 1)
        if noise is not None:
            x = x + noise.to(x.dtype)
        return x
    if demodulate:
        wmod = wmod * dcoefs.reshape(N, O, 1, 1, 1)
    x = conv2d_resample(x.reshape(1, N * I, *x.shape[2:]), wmod.reshape(N * O, I, kh, kw).to(x.dtype), f=resample_filter, up=up, down=down,
                        padding=padding, groups=N, flip_weight=flip_weight)
    x = x.reshape(N, O, *x.shape[2:])
    return x if noise is None else x + noise


class FullyConnectedLayer(torch.nn.Module):
    """y = act(x W^T * (lr_mul / sqrt(in)) + b * lr_mul)   (networks_stylegan2.py:96-131)"""

    def __init__(self, in_features, out_features, bias=True, activation='linear', lr_multiplier=1, bias_init=0):
        super().__init__()
        self.in_features, self.out_features, self.activation = in_features, out_features, activation
        self.weight = torch.nn.Parameter(torch.randn([out_features, in_features]) / lr_multiplier)
        self.bias = torch.nn.Parameter(torch.full([out_features], float(bias_init))) if bias else None
        self.weight_gain = lr_multiplier / math.sqrt(in_features)
        self.bias_gain = lr_multiplier

    def forward(self, x):
        w = self.weight.to(x.dtype) * self.weight_gain
        b = None if self.bias is None else self.bias.to(x.dtype) * self.bias_gain
        if self.activation == 'linear' and b is not None:
            return torch.addmm(b.unsqueeze(0), x, w.t())
        return _bias_act(x.matmul(w.t()), b, act=self.activation)

    def extra_repr(self):
        return f'in_features={self.in_features:d}, out_features={self.out_features:d}, activation={self.activation:s}'


class Conv2dLayer(torch.nn.Module):
    """Plain (un-modulated) convolution with resampling, bias and activation (networks_stylegan2.py:136-190); the 'resnet' skip."""

    def __init__(self, in_channels, out_channels, kernel_size, bias=True, activation='linear', up=1, down=1, resample_filter=[1, 3, 3, 1],
                 conv_clamp=None, channels_last=False, trainable=True):
        super().__init__()
        self.in_channels, self.out_channels, self.activation, self.up, self.down, self.conv_clamp = in_channels, out_channels, activation, up, down, conv_clamp
        self.register_buffer('resample_filter', _uf.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.weight_gain = 1 / math.sqrt(in_channels * kernel_size ** 2)
        self.act_gain = _ba.activation_funcs[activation]['def_gain']
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        weight = torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt)
        b = torch.zeros([out_channels]) if bias else None
        if trainable:
            self.weight = torch.nn.Parameter(weight)
            self.bias = torch.nn.Parameter(b) if b is not None else None
        else:
            self.register_buffer('weight', weight)
            if b is not None:
                self.register_buffer('bias', b)
            else:
                self.bias = None

    def forward(self, x, gain=1):
        b = None if self.bias is None else self.bias.to(x.dtype)
        x = conv2d_resample(x, (self.weight * self.weight_gain).to(x.dtype), f=self.resample_filter, up=self.up, down=self.down, padding=self.padding,
                            flip_weight=self.up == 1)
        return _bias_act(x, b, act=self.activation, gain=self.act_gain * gain, clamp=None if self.conv_clamp is None else self.conv_clamp * gain)


class MappingNetwork(torch.nn.Module):
    """z (and the embedded label c) -> w, broadcast to num_ws, optional truncation towards the tracked mean (networks_stylegan2.py:195-264)."""

    def __init__(self, z_dim, c_dim, w_dim, num_ws, num_layers=8, embed_features=None, layer_features=None, activation='lrelu',
                 lr_multiplier=0.01, w_avg_beta=0.998):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.num_ws, self.num_layers, self.w_avg_beta = z_dim, c_dim, w_dim, num_ws, num_layers, w_avg_beta
        embed_features = 0 if c_dim == 0 else (w_dim if embed_features is None else embed_features)
        layer_features = w_dim if layer_features is None else layer_features
        feats = [z_dim + embed_features] + [layer_features] * (num_layers - 1) + [w_dim]
        if c_dim > 0:
            self.embed = FullyConnectedLayer(c_dim, embed_features)
        for i in range(num_layers):
            setattr(self, f'fc{i}', FullyConnectedLayer(feats[i], feats[i + 1], activation=activation, lr_multiplier=lr_multiplier))
        if num_ws is not None and w_avg_beta is not None:
            self.register_buffer('w_avg', torch.zeros([w_dim]))

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False):
        x = None
        if self.z_dim > 0:
            if z.ndim != 2 or z.shape[1] != self.z_dim:
                raise RuntimeError(f'MappingNetwork: z must be [N, {self.z_dim}]')
            x = normalize_2nd_moment(z.to(torch.float32))
        # SHERF disabled the label embedding (networks_stylegan2.py:239-242 are commented out): `c` is ignored and every SHERF
        # configuration builds the generator with c_dim = 0 (training_loop.py:192); `embed` exists only for the checkpoint contract.
        for i in range(self.num_layers):
            x = getattr(self, f'fc{i}')(x)
        if update_emas and self.w_avg_beta is not None:
            self.w_avg.copy_(x.detach().mean(dim=0).lerp(self.w_avg, self.w_avg_beta))
        if self.num_ws is not None:
            x = x.unsqueeze(1).repeat([1, self.num_ws, 1])
        if truncation_psi != 1:
            if self.num_ws is None or truncation_cutoff is None:
                x = self.w_avg.lerp(x, truncation_psi)
            else:
                x[:, :truncation_cutoff] = self.w_avg.lerp(x[:, :truncation_cutoff], truncation_psi)
        return x


class SynthesisLayer(torch.nn.Module):
    """style affine -> modulated 3x3 convolution (+ per-pixel noise) -> bias, lrelu * sqrt(2), clamp (networks_stylegan2.py:269-325)."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, kernel_size=3, up=1, use_noise=True, activation='lrelu',
                 resample_filter=[1, 3, 3, 1], conv_clamp=None, channels_last=False):
        super().__init__()
        self.in_channels, self.out_channels, self.w_dim, self.resolution, self.up = in_channels, out_channels, w_dim, resolution, up
        self.use_noise, self.activation, self.conv_clamp = use_noise, activation, conv_clamp
        self.register_buffer('resample_filter', _uf.setup_filter(resample_filter))
        self.padding = kernel_size // 2
        self.act_gain = _ba.activation_funcs[activation]['def_gain']
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt))
        if use_noise:
            self.register_buffer('noise_const', torch.randn([resolution, resolution]))
            self.noise_strength = torch.nn.Parameter(torch.zeros([]))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))

    def forward(self, x, w, noise_mode='random', fused_modconv=True, gain=1):
        if noise_mode not in ('random', 'const', 'none'):
            raise RuntimeError(f'SynthesisLayer: noise_mode {noise_mode!r}')
        r_in = self.resolution // self.up
        if tuple(x.shape[1:]) != (self.in_channels, r_in, r_in):
            raise RuntimeError(f'SynthesisLayer: expected [N, {self.in_channels}, {r_in}, {r_in}], got {tuple(x.shape)}')
        styles = self.affine(w)
        noise = None
        if self.use_noise and noise_mode == 'random':
            noise = torch.randn([x.shape[0], 1, self.resolution, self.resolution], device=x.device) * self.noise_strength
        if self.use_noise and noise_mode == 'const':
            noise = self.noise_const * self.noise_strength
        x = modulated_conv2d(x, self.weight, styles, noise=noise, up=self.up, padding=self.padding, resample_filter=self.resample_filter,
                             flip_weight=self.up == 1, fused_modconv=fused_modconv)
        return _bias_act(x, self.bias.to(x.dtype), act=self.activation, gain=self.act_gain * gain,
                         clamp=None if self.conv_clamp is None else self.conv_clamp * gain)


class ToRGBLayer(torch.nn.Module):
    """1x1 modulated convolution without demodulation + bias (networks_stylegan2.py:330-350)."""

    def __init__(self, in_channels, out_channels, w_dim, kernel_size=1, conv_clamp=None, channels_last=False):
        super().__init__()
        self.in_channels, self.out_channels, self.w_dim, self.conv_clamp = in_channels, out_channels, w_dim, conv_clamp
        self.affine = FullyConnectedLayer(w_dim, in_channels, bias_init=1)
        fmt = torch.channels_last if channels_last else torch.contiguous_format
        self.weight = torch.nn.Parameter(torch.randn([out_channels, in_channels, kernel_size, kernel_size]).to(memory_format=fmt))
        self.bias = torch.nn.Parameter(torch.zeros([out_channels]))
        self.weight_gain = 1 / math.sqrt(in_channels * kernel_size ** 2)

    def forward(self, x, w, fused_modconv=True):
        x = modulated_conv2d(x, self.weight, self.affine(w) * self.weight_gain, demodulate=False, fused_modconv=fused_modconv)
        return _bias_act(x, self.bias.to(x.dtype), clamp=self.conv_clamp)


class SynthesisBlock(torch.nn.Module):
    """One resolution of the pyramid (networks_stylegan2.py:355-443): [conv0 (x2 up)] -> conv1 -> torgb, image skip-summed."""

    def __init__(self, in_channels, out_channels, w_dim, resolution, img_channels, is_last, architecture='skip', resample_filter=[1, 3, 3, 1],
                 conv_clamp=256, use_fp16=False, fp16_channels_last=False, fused_modconv_default=True, **layer_kwargs):
        if architecture not in ('orig', 'skip', 'resnet'):
            raise RuntimeError(f'SynthesisBlock: architecture {architecture!r}')
        super().__init__()
        self.in_channels, self.w_dim, self.resolution, self.img_channels, self.is_last = in_channels, w_dim, resolution, img_channels, is_last
        self.architecture, self.use_fp16 = architecture, use_fp16
        self.channels_last = use_fp16 and fp16_channels_last
        self.fused_modconv_default = fused_modconv_default
        self.register_buffer('resample_filter', _uf.setup_filter(resample_filter))
        self.num_conv = self.num_torgb = 0
        if in_channels == 0:
            self.const = torch.nn.Parameter(torch.randn([out_channels, resolution, resolution]))
        else:
            self.conv0 = SynthesisLayer(in_channels, out_channels, w_dim=w_dim, resolution=resolution, up=2, resample_filter=resample_filter,
                                        conv_clamp=conv_clamp, channels_last=self.channels_last, **layer_kwargs)
            self.num_conv += 1
        self.conv1 = SynthesisLayer(out_channels, out_channels, w_dim=w_dim, resolution=resolution, conv_clamp=conv_clamp,
                                    channels_last=self.channels_last, **layer_kwargs)
        self.num_conv += 1
        if is_last or architecture == 'skip':
            self.torgb = ToRGBLayer(out_channels, img_channels, w_dim=w_dim, conv_clamp=conv_clamp, channels_last=self.channels_last)
            self.num_torgb += 1
        if in_channels != 0 and architecture == 'resnet':
            self.skip = Conv2dLayer(in_channels, out_channels, kernel_size=1, bias=False, up=2, resample_filter=resample_filter,
                                    channels_last=self.channels_last)

    def forward(self, x, img, ws, force_fp32=False, fused_modconv=None, update_emas=False, **layer_kwargs):
        if ws.ndim != 3 or ws.shape[1] != self.num_conv + self.num_torgb or ws.shape[2] != self.w_dim:
            raise RuntimeError(f'SynthesisBlock: ws must be [N, {self.num_conv + self.num_torgb}, {self.w_dim}]')
        w = iter(ws.unbind(dim=1))
        if not ws.is_cuda:
            force_fp32 = True
        dtype = torch.float16 if self.use_fp16 and not force_fp32 else torch.float32
        fmt = torch.channels_last if self.channels_last and not force_fp32 else torch.contiguous_format
        if fused_modconv is None:
            fused_modconv = self.fused_modconv_default
        if fused_modconv == 'inference_only':
            fused_modconv = not self.training
        if self.in_channels == 0:
            x = self.const.to(dtype=dtype, memory_format=fmt).unsqueeze(0).repeat([ws.shape[0], 1, 1, 1])
            x = self.conv1(x, next(w), fused_modconv=fused_modconv, **layer_kwargs)
        else:
            x = x.to(dtype=dtype, memory_format=fmt)
            if self.architecture == 'resnet':
                y = self.skip(x, gain=math.sqrt(0.5))
                x = self.conv0(x, next(w), fused_modconv=fused_modconv, **layer_kwargs)
                x = y + self.conv1(x, next(w), fused_modconv=fused_modconv, gain=math.sqrt(0.5), **layer_kwargs)
            else:
                x = self.conv0(x, next(w), fused_modconv=fused_modconv, **layer_kwargs)
                x = self.conv1(x, next(w), fused_modconv=fused_modconv, **layer_kwargs)
        if img is not None:
            img = _uf.upsample2d(img, self.resample_filter, impl=OPS_IMPL)
        if self.is_last or self.architecture == 'skip':
            y = self.torgb(x, next(w), fused_modconv=fused_modconv).to(dtype=torch.float32, memory_format=torch.contiguous_format)
            img = y if img is None else img + y
        return x, img


class SynthesisNetwork(torch.nn.Module):
    """ws [N, num_ws, w_dim] -> image [N, img_channels, R, R] (networks_stylegan2.py:448-533): blocks b4 .. b{R}, each fed its own
    slice of ws (the last w of a block is shared with the next block's first layer)."""

    def __init__(self, w_dim, img_resolution, img_channels, channel_base=32768, channel_max=512, num_fp16_res=4, **block_kwargs):
        if img_resolution < 4 or img_resolution & (img_resolution - 1):
            raise RuntimeError('SynthesisNetwork: img_resolution must be a power of two >= 4')
        super().__init__()
        self.w_dim, self.img_resolution, self.img_channels, self.num_fp16_res = w_dim, img_resolution, img_channels, num_fp16_res
        self.img_resolution_log2 = int(math.log2(img_resolution))
        self.block_resolutions = [2 ** i for i in range(2, self.img_resolution_log2 + 1)]
        channels = {r: min(channel_base // r, channel_max) for r in self.block_resolutions}
        fp16_resolution = max(2 ** (self.img_resolution_log2 + 1 - num_fp16_res), 8)
        self.num_ws = 0
        for r in self.block_resolutions:
            block = SynthesisBlock(channels[r // 2] if r > 4 else 0, channels[r], w_dim=w_dim, resolution=r, img_channels=img_channels,
                                   is_last=r == img_resolution, use_fp16=r >= fp16_resolution, **block_kwargs)
            self.num_ws += block.num_conv
            if r == img_resolution:
                self.num_ws += block.num_torgb
            setattr(self, f'b{r}', block)

    def forward(self, ws, **block_kwargs):
        if ws.ndim != 3 or ws.shape[1] != self.num_ws or ws.shape[2] != self.w_dim:
            raise RuntimeError(f'SynthesisNetwork: ws must be [N, {self.num_ws}, {self.w_dim}]')
        ws = ws.to(torch.float32)
        x = img = None
        i = 0
        for r in self.block_resolutions:
            block = getattr(self, f'b{r}')
            x, img = block(x, img, ws.narrow(1, i, block.num_conv + block.num_torgb), **block_kwargs)
            i += block.num_conv
        return img


class Generator(torch.nn.Module):
    """networks_stylegan2.py:538-561 (`StyleGAN2Backbone` in triplane.py:16,58): z, c -> ws -> image."""

    def __init__(self, z_dim, c_dim, w_dim, img_resolution, img_channels, mapping_kwargs={}, **synthesis_kwargs):
        super().__init__()
        self.z_dim, self.c_dim, self.w_dim, self.img_resolution, self.img_channels = z_dim, c_dim, w_dim, img_resolution, img_channels
        self.synthesis = SynthesisNetwork(w_dim=w_dim, img_resolution=img_resolution, img_channels=img_channels, **synthesis_kwargs)
        self.num_ws = self.synthesis.num_ws
        self.mapping = MappingNetwork(z_dim=z_dim, c_dim=c_dim, w_dim=w_dim, num_ws=self.num_ws, **mapping_kwargs)

    def forward(self, z, c, truncation_psi=1, truncation_cutoff=None, update_emas=False, **synthesis_kwargs):
        ws = self.mapping(z, c, truncation_psi=truncation_psi, truncation_cutoff=truncation_cutoff, update_emas=update_emas)
        return self.synthesis(ws, update_emas=update_emas, **synthesis_kwargs)
