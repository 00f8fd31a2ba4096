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


def load_seeded_state(module, prefix=''):
    """Overwrite every parameter/buffer of a torch module in place from `seeded_param`."""
    import torch
    with torch.no_grad():
        for name, t in list(module.named_parameters()) + list(module.named_buffers()):
            v = seeded_param(prefix + name, t.shape)
            if v is not None:
                t.copy_(torch.from_numpy(v).to(t.dtype))


def renderer_inputs(cfg_name, smpl=None):
    """numpy inputs at the ImportanceRenderer.forward boundary (renderer.py:286)."""
    c = CONFIGS[cfg_name]
    smpl = smpl or synth.make_synth_smpl(0)
    d = synth.make_input_data(smpl, H=c['H'], W=c['W'], seed=1, theta_tgt=c['theta_tgt'],
                              theta_obs=c['theta_obs'], novel_pose=c['novel_pose'], fill=c.get('fill', 2.2))
    r = _rs('inputs/' + cfg_name)
    P = c['plane_res']
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
