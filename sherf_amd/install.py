"""`sherf_amd.install()`: put the MI355X path under an UNMODIFIED checkout of the reference (second recipe).

    import sys; sys.path.insert(0, '<SHERF>/sherf')
    import sherf_amd.install; sherf_amd.install.install()
    # ... then the reference's own entry points: train.py / test loops construct `training.triplane.TriPlaneGenerator`
    # (train.py:310 -> training_loop.py:193), whose renderer / decoder / sparse tensor are now the classes of this package.

What is rebound (names in the reference's modules; none of its files is edited, none of its objects mutated):

    training.volumetric_rendering.renderer.ImportanceRenderer, training.triplane.ImportanceRenderer -> sherf_amd.renderer.ImportanceRenderer
    training.triplane.NeRFDecoder                                                                 -> sherf_amd.triplane.NeRFDecoder
    training.triplane.RaySampler                                                                  -> sherf_amd.ray_sampler.RaySampler
    training.triplane.spconv (only the attribute path `spconv.core.SparseConvTensor`, triplane.py:137) -> sherf_amd.voxel.SparseConvTensor

The reference's `TriPlaneGenerator` class itself keeps running: its `synthesis` (triplane.py:81-172) calls `renderer.projection`,
`renderer.coarse_deform_target2c`, `renderer.rgb_enc`, `renderer.SMPL_NEUTRAL['f']` and `renderer(...)`, all of which
`sherf_amd.renderer.ImportanceRenderer` provides.  With `stub_missing=True` (default) the two CUDA-only third-party packages the
reference imports at module level -- pytorch3d (K-NN) and spconv -- are replaced by import stubs when they are not installed: every
use of them sits inside the classes rebound above, so on a ROCm machine the reference then imports without them (calling a stub
raises).  tests/test_reference_dropin.py runs the reference's generator this way, end to end, against its own unpatched output.
"""
import importlib
import sys
import types


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__sherf_amd_stub__ = True
    sys.modules[name] = m
    return m


def _missing(*a, **k):
    raise RuntimeError('this third-party CUDA package is not installed; sherf_amd replaces the reference code that used it')


def stub_missing_packages():
    """Import stubs for pytorch3d.ops.knn and spconv.pytorch / spconv.core when the real packages are absent."""
    made = []
    try:
        importlib.import_module('pytorch3d.ops.knn')
    except Exception:
        p3d, ops = _stub('pytorch3d'), _stub('pytorch3d.ops')
        knn = _stub('pytorch3d.ops.knn', knn_points=_missing, knn_gather=_missing)
        p3d.ops, ops.knn = ops, knn
        made.append('pytorch3d')
    try:
        importlib.import_module('spconv.pytorch')
    except Exception:
        from . import voxel
        core = _stub('spconv.core', SparseConvTensor=voxel.SparseConvTensor)
        cont = _checkpoint_containers()
        pt = _stub('spconv.pytorch', core=core, SparseConvTensor=voxel.SparseConvTensor, **cont)
        # the module paths spconv 2.x's own classes live at (what a checkpoint pickled with the real package refers to)
        pt.conv = _stub('spconv.pytorch.conv', **{k: v for k, v in cont.items() if 'Conv' in k})
        pt.modules = _stub('spconv.pytorch.modules', SparseModule=cont['SparseModule'], SparseSequential=cont['SparseSequential'])
        sp = _stub('spconv', pytorch=pt, core=core)
        made.append('spconv')
    return made


def _checkpoint_containers():
    """Classes under spconv's names that can only be UNPICKLED: a network snapshot written where spconv was installed
    (training_loop.py:563-579 pickles the whole generator; the sparse layers go in by import path) then loads on a machine without it --
    `legacy.load_network_pkl` (legacy.py:24-62) needs the classes to exist, `misc.copy_params_and_buffers` (training_loop.py:207-208)
    only reads their parameters.  Constructing or calling one raises."""
    import torch.nn as nn

    def make(name):
        def __init__(self, *a, **k):
            _missing()
        return type(name, (nn.Module,), dict(__init__=__init__, forward=lambda self, *a, **k: _missing(), __module__='spconv.pytorch',
                                             __doc__='parameter container standing in for spconv.pytorch.%s in unpickled checkpoints' % name))
    return {n: make(n) for n in ('SparseModule', 'SparseSequential', 'SubMConv3d', 'SparseConv3d', 'SparseInverseConv3d', 'SubMConv2d', 'SparseConv2d')}


def install(stub_missing=True):
    """Rebinds the reference's names to this package's classes (see the module docstring). Idempotent. -> dict of what was done."""
    from . import ray_sampler, renderer, triplane, voxel
    stubs = stub_missing_packages() if stub_missing else []
    R = importlib.import_module('training.volumetric_rendering.renderer')
    T = importlib.import_module('training.triplane')
    R.ImportanceRenderer = T.ImportanceRenderer = renderer.ImportanceRenderer
    # under torch.enable_grad() the hosted renderer records itself as ONE autograd node whose backward is the HIP backward pipeline
    # (what the reference's loss.backward() at loss.py:173 then runs through); under no_grad -- every test / inference entry point of the
    # reference -- nothing changes
    renderer.ImportanceRenderer.enable_autograd = True
    T.NeRFDecoder = triplane.NeRFDecoder
    T.RaySampler = ray_sampler.RaySampler
    T.spconv = types.SimpleNamespace(core=types.SimpleNamespace(SparseConvTensor=voxel.SparseConvTensor))
    return dict(stubs=stubs, modules=(R.__name__, T.__name__))
