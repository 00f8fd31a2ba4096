"""One rank of tests/test_hipcpu_frame.py::test_ray_tile_sharding_two_ranks: ONE frame sharded over two ranks by interleaved ray tiles
(sherf_amd.dist.render_ray_tiles) through the REAL renderer (kernels' source on the host build), gloo backend; every rank checks the
gathered frame against its own single-process render of the whole frame: rgb, DEPTH (frame-wide clamp range) and acc bit for bit."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from sherf_amd import _lib, dist as sdist
    import sherf_amd.renderer as AR
    _lib.LIB_PATH, _lib._lib = os.environ['SHERF_HIPCPU_LIB'], None
    _lib.ptr = lambda t, dtype=None, channels_last_ok=False: None if t is None else ctypes.c_void_p(t.data_ptr())
    _lib.addr = lambda t, dtype=None: None if t is None else t.data_ptr()
    _lib.stream = lambda: ctypes.c_void_p(0)
    torch.cuda.current_stream = lambda dev=None: type('S', (), {'cuda_stream': 0})()
    torch.cuda.synchronize = lambda dev=None: None
    AR.ImportanceRenderer._side = lambda self, dev, idx=0: type('HostStream', (), {'cuda_stream': 8 + 8 * idx})()
    AR.ImportanceRenderer.SMPL_NEUTRAL = property(lambda self: self._smpl(torch.device('cpu')))
    torch.distributed.init_process_group('gloo')
    from tests import gpu_common as G
    G.CPU_SHIM = True
    cfg = 'tiny'
    fx = G.fixture(cfg)
    sp_input = G.oracle_render(cfg)['sp_input']
    whole = G.hip_render(cfg)                                            # this process, every ray
    d = G.to_cuda(fx['input_data'])

    def render(ro, rd, nr, fr, extra):
        import numpy as np
        sub = {k: (dict(v) if isinstance(v, dict) else v) for k, v in fx['input_data'].items()}
        sub.update(ray_o_all=G.plain(ro)[None].numpy(), ray_d_all=G.plain(rd)[None].numpy(), near_all=G.plain(nr)[None].numpy(),
                   far_all=G.plain(fr)[None].numpy())
        f2 = dict(fx); f2['input_data'] = sub
        h = G.hip_render(cfg, fx=f2, sp_input=sp_input, options=extra)
        return h['rgb'][None], h['depth'][None, :, None], h['acc'][None, :, None]

    full = sdist.render_ray_tiles(render, d['ray_o_all'][:, 0], d['ray_d_all'][:, 0], d['near_all'][:, 0], d['far_all'][:, 0], tile=64)
    full = G.plain(full)
    assert torch.equal(full[:, :3], whole['rgb']) and torch.equal(full[:, 4], whole['acc'])
    assert torch.equal(full[:, 3], whole['depth']), float((full[:, 3] - whole['depth']).abs().max())
    # without the frame-wide range the subset's own depth extrema would be used: the clamp must actually have been exercised
    rng = G.plain(sdist.depth_range(d['near_all'][:, 0], d['far_all'][:, 0]))
    assert float(rng[0]) <= float(whole['depth'].min()) and float(whole['depth'].max()) <= float(rng[1])
    print(f'RAYTILE_OK rank {torch.distributed.get_rank()} depth range [{float(rng[0]):.4f}, {float(rng[1]):.4f}]', flush=True)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
