"""TEST INFRASTRUCTURE (oracle/). Seeded weights and renderer-boundary inputs shared by
`oracle/make_golden.py` (which feeds them to the unmodified reference) and by `tests/` (which feed
the same values to the oracle restatement and to the HIP path). Only numpy RandomState is used so the
numbers are identical on every box; golden files therefore store OUTPUTS only.
"""
import zlib
import numpy as np

from . import synth

CONFIGS = {
    # name: H, W, samples/ray, plane resolution, novel pose?
    'tiny':  dict(H=32, W=32, S=16, plane_res=32, novel_pose=True, theta_tgt=0.4, theta_obs=-0.3),
    'tiny_nv': dict(H=24, W=40, S=12, plane_res=16, novel_pose=False, theta_tgt=1.1, theta_obs=-0.2),
    'cfg1':  dict(H=128, W=128, S=32, plane_res=256, novel_pose=True, theta_tgt=0.4, theta_obs=-0.3),
    'cfg2':  dict(H=512, W=512, S=64, plane_res=256, novel_pose=False, theta_tgt=0.4, theta_obs=-0.3),
    'cfg3':  dict(H=512, W=512, S=64, plane_res=256, novel_pose=True, theta_tgt=0.4, theta_obs=-0.3),
    # cfg2 framed tighter (the frame spans 1.55 m at the subject instead of 2.2 m): valid-sample fraction ~ SURVEY 8(d)'s probe value 0.076
    'cfg2_dense': dict(H=512, W=512, S=64, plane_res=256, novel_pose=False, theta_tgt=0.4, theta_obs=-0.3, fill=1.55),
}
# The "_ri" VARIANT of every configuration: same geometry, but the network the SURVEY section 8(d) names -- every parameter drawn from
# the distribution the reference's own constructors use (`refinit_param`), `alpha_linear.bias += 5` -- and band-limited feature tables
# (noise at an 8-texel pitch, interpolated) instead of white noise: the well-conditioned workload on which plain per-sample relative
# error <= 1e-3 is required outright.  The seeded variant stays as the adversarial case (weights 2.4 x the
# default scale, density head x 20, white-noise tables).
for _n in list(CONFIGS):
    CONFIGS[_n + '_ri'] = dict(CONFIGS[_n], variant='ri')


def _rs(name):
    return np.random.RandomState(zlib.crc32(name.encode()) & 0x7FFFFFFF)


def seeded_param(name, shape):
    """Deterministic value for a parameter/buffer called `name` (None = leave the module's own value)."""
    shape = tuple(int(s) for s in shape)
    leaf = name.split('.')[-1]
    if leaf in ('_freqs', '_phases', 'num_batches_tracked'):
        return None
    r = _rs(name)
    if leaf == 'running_var':
        return (1.0 + 0.2 * r.uniform(0, 1, shape)).astype(np.float32)
    if leaf == 'running_mean':
        return (0.1 * r.standard_normal(shape)).astype(np.float32)
    if len(shape) == 1:
        if leaf == 'weight':
            return (1.0 + 0.1 * r.standard_normal(shape)).astype(np.float32)
        out = (0.1 * r.standard_normal(shape)).astype(np.float32)
        if name.endswith('alpha_linear.bias'):
            out = out + 3.0
        return out.astype(np.float32)
    fan_in = int(np.prod(shape[1:]))
    w = (1.4 / np.sqrt(fan_in)) * r.standard_normal(shape)
    if name.endswith('alpha_linear.weight'):
        w = w * 20.0
    return w.astype(np.float32)


def refinit_param(name, shape, fan_in=None):
    """Value of parameter `name` drawn from the distribution the REFERENCE's constructors give it (numpy-seeded by name, so the numbers
    are the same on every box and no 10 MB weight file has to travel):
      nn.Linear / nn.Conv1d / spconv conv weights: kaiming_uniform_(a=sqrt(5)) = U(-1/sqrt(fan_in), 1/sqrt(fan_in)), fan_in = prod(shape[1:])
        (torch/nn/modules/linear.py reset_parameters; renderer.py:271-276, triplane.py:277-283; the sparse convolutions have no bias,
        renderer.py:803-870);
      their biases: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) with the WEIGHT's fan_in (pass `fan_in`);
      BatchNorm1d / LayerNorm: weight 1, bias 0, running_mean 0, running_var 1;
      decoder.alpha_linear.bias += 5 (a random-init density head gives a black image otherwise).
    tests/test_oracle_golden.py::test_refinit_matches_reference_constructors checks these claims against modules built by the
    unmodified reference under torch.manual_seed(0)."""
    shape = tuple(int(s) for s in shape)
    leaf = name.split('.')[-1]
    if leaf in ('_freqs', '_phases', 'num_batches_tracked'):
        return None
    norm = '.norm.' in name or (name.startswith('renderer.encoder_3d.') and len(shape) == 1)
    if leaf == 'running_var' or (norm and leaf == 'weight'):
        return np.ones(shape, np.float32)
    if leaf == 'running_mean' or (norm and leaf == 'bias'):
        return np.zeros(shape, np.float32)
    r = _rs('refinit/' + name)
    if len(shape) == 1:
        assert fan_in, f'bias {name}: fan_in of its weight needed'
        b = 1.0 / np.sqrt(fan_in)
        out = r.uniform(-b, b, shape)
        if name.endswith('alpha_linear.bias'):
            out = out + 5.0
        return out.astype(np.float32)
    b = 1.0 / np.sqrt(float(np.prod(shape[1:])))
    return r.uniform(-b, b, shape).astype(np.float32)


def param_value(variant, name, shape, shapes=None):
    """`seeded_param` ('seeded') or `refinit_param` ('ri'); `shapes` (name -> shape) supplies the fan-in of a bias."""
    if variant != 'ri':
        return seeded_param(name, shape)
    fan_in = None
    if len(tuple(shape)) == 1 and name.endswith('.bias') and shapes is not None:
        w = shapes.get(name[:-len('bias')] + 'weight')
        if w is not None and len(w) > 1:
            fan_in = int(np.prod(w[1:]))
    if len(tuple(shape)) == 1 and fan_in is None and name.endswith('.bias') and not ('.norm.' in name or name.startswith('renderer.encoder_3d.')):
        raise KeyError(f'{name}: weight shape unknown')
    return refinit_param(name, shape, fan_in)


def variant_of(cfg_name):
    return CONFIGS[cfg_name].get('variant', 'seeded') if cfg_name in CONFIGS else ('ri' if cfg_name.endswith('_ri') else 'seeded')


def load_seeded_state(module, prefix='', variant='seeded'):
    """Overwrite every parameter/buffer of a torch module in place from `seeded_param` (or `refinit_param`: variant 'ri')."""
    import torch
    named = list(module.named_parameters()) + list(module.named_buffers())
    shapes = {prefix + n: tuple(t.shape) for n, t in named}
    with torch.no_grad():
        for name, t in named:
            v = param_value(variant, prefix + name, t.shape, shapes)
            if v is not None:
                t.copy_(torch.from_numpy(v).to(t.dtype))


def smooth_noise(r, shape, pitch=8):
    """Band-limited noise [..., H, W]: standard normal values on a grid of `pitch` texels, bilinearly interpolated (plain numpy: the
    same bits on every box), rescaled to unit variance.  A texel step changes it by ~1/pitch of its range -- the opposite of the
    white-noise tables of the seeded variant, where a 1e-3-texel move of a tap is a 1e-3 change of an O(1) value."""
    *lead, H, W = shape
    gh, gw = H // pitch + 2, W // pitch + 2
    g = r.standard_normal((*lead, gh, gw))
    y = (np.arange(H) + 0.5) / pitch; x = (np.arange(W) + 0.5) / pitch
    y0 = np.floor(y).astype(int); x0 = np.floor(x).astype(int)
    fy = (y - y0)[:, None]; fx = (x - x0)[None, :]
    a = g[..., y0[:, None], x0[None, :]]; b = g[..., y0[:, None], x0[None, :] + 1]
    c = g[..., y0[:, None] + 1, x0[None, :]]; d = g[..., y0[:, None] + 1, x0[None, :] + 1]
    out = (a * (1 - fx) + b * fx) * (1 - fy) + (c * (1 - fx) + d * fx) * fy
    return (out / 0.67).astype(np.float32)          # E[w^2] of the bilinear weights = (2/3)^2 -> std 0.67


def renderer_inputs(cfg_name, smpl=None):
    """numpy inputs at the ImportanceRenderer.forward boundary (renderer.py:286)."""
    c = CONFIGS[cfg_name]
    smpl = smpl or synth.make_synth_smpl(0)
    d = synth.make_input_data(smpl, H=c['H'], W=c['W'], seed=1, theta_tgt=c['theta_tgt'],
                              theta_obs=c['theta_obs'], novel_pose=c['novel_pose'], fill=c.get('fill', 2.2))
    r = _rs('inputs/' + cfg_name)
    P = c['plane_res']
    if c.get('variant') == 'ri':
        planes = smooth_noise(r, (1, 3, 32, P, P))
        feat = smooth_noise(r, (1, 64, c['H'] // 2, c['W'] // 2))
        img = np.clip(0.5 + 0.25 * smooth_noise(r, (3, c['H'], c['W'])), 0.0, 1.0).astype(np.float32)
        d = dict(d); d['obs_img_all'] = img[None, None]
        # per-vertex features: a smooth function of the observation-pose vertex position (what projecting neighbouring vertices into a
        # smooth feature map gives, triplane.py:111-126): 32 channels x 6 plane waves of wavelength >= 8 cm
        v = d['obs_vertices'][0].astype(np.float64)
        k = r.standard_normal((32, 6, 3)); k = k / np.linalg.norm(k, axis=-1, keepdims=True) * (2 * np.pi / r.uniform(0.08, 0.4, (32, 6, 1)))
        ph = r.uniform(0, 2 * np.pi, (32, 6)); amp = r.standard_normal((32, 6)) * 0.3
        vfeat = (np.sin(np.einsum('vd,ckd->vck', v, k) + ph[None]) * amp[None]).sum(-1).astype(np.float32)
        vfeat[r.uniform(0, 1, synth.V) < 0.35] = 0.0
    else:
        planes = r.standard_normal((1, 3, 32, P, P)).astype(np.float32)
        feat = r.standard_normal((1, 64, c['H'] // 2, c['W'] // 2)).astype(np.float32)
        vfeat = (0.5 * r.standard_normal((synth.V, 32))).astype(np.float32)
        vfeat[r.uniform(0, 1, synth.V) < 0.35] = 0.0          # back-facing vertices carry zeros (triplane.py:126)
    opts = dict(depth_resolution=c['S'], disparity_space_sampling=False, depth_resolution_importance=0,
                clamp_mode='relu', white_back=False, density_noise=0)
    return dict(input_data=d, planes=planes, obs_feat=feat, vertex_feat=vfeat, options=opts, cfg=c, smpl=smpl)


def to_torch(x):
    import torch
    if isinstance(x, dict):
        return {k: to_torch(v) for k, v in x.items()}
    if isinstance(x, np.ndarray):
        return torch.from_numpy(np.ascontiguousarray(x))
    return x
