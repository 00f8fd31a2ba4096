"""One rank of tests/test_hipcpu_frame.py::test_bench_two_ranks_dry_run: bench.py's N > 1 path (process group, per-step all_gather of
the rendered tiles, barrier-bracketed timing, max over ranks) run with the gloo backend on the host build of the kernels."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from sherf_amd import _lib
    import sherf_amd.renderer as AR
    import bench
    _lib.LIB_PATH, _lib._lib = os.environ['SHERF_HIPCPU_LIB'], None
    _lib.ptr = lambda t, dtype=None, channels_last_ok=False: None if t is None else ctypes.c_void_p(t.data_ptr())
    _lib.addr = lambda t, dtype=None: None if t is None else t.data_ptr()
    _lib.stream = lambda: ctypes.c_void_p(0)
    torch.cuda.current_stream = lambda dev=None: type('S', (), {'cuda_stream': 0})()
    torch.cuda.synchronize = lambda dev=None: None
    torch.Tensor.is_cuda = property(lambda self: True)
    AR.ImportanceRenderer._side = lambda self, dev, idx=0: type('HostStream', (), {'cuda_stream': 8 + 8 * idx})()
    AR.ImportanceRenderer.SMPL_NEUTRAL = property(lambda self: self._smpl(torch.device('cpu')))
    bench._device = lambda lrank: torch.device('cpu')
    sys.argv = ['bench.py', '--gpus', os.environ['WORLD_SIZE'], '--config', 'tiny', '--steps', os.environ.get('SHERF_BENCH_STEPS', '1'), '--warmup', '1', '--no-cpu-baseline',
                '--streams', os.environ.get('SHERF_BENCH_STREAMS', '1'), '--no-torch-gpu-baseline'] + (['--partition', os.environ['SHERF_BENCH_PARTITION']] if os.environ.get('SHERF_BENCH_PARTITION') else [])
    bench.main()


if __name__ == '__main__':
    main()
