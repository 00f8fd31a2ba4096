"""Single-process A/B of sherf_gather_mlp builds on ONE frame's buffers (diagnostic):

    gpurun -- 'python tools/fused_ab.py'

Renders cfg1 with the product library, keeps the frame's workspace, then launches sherf_nerf_mlp (reference: tokens from the frame)
and every libsherf_hip*.so's sherf_gather_mlp on the same inputs several times, comparing sample_out per tile.
"""
import ctypes as ct
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from sherf_amd import _lib
    from sherf_amd.renderer import MLP_PRECISIONS
    from tests import gpu_common as G
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg1'
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    r = G.hip_render(cfg, options=dict(split_gather=True))
    rend, dec, last = r['rend'], r['dec'], r['last']
    dev = torch.device('cuda', 0)
    ws, cap = last['ws'], int(last['cap'])
    rows, planes_f, feat_f, img4 = last['keep']
    b = last['bwd']
    wc = rend._weights(dec, dev, 'f16x3')
    nv = int(ws['counters'][0]); nt = (nv + 31) // 32
    f32 = lambda t: t.detach().to(device=dev, dtype=torch.float32).contiguous()
    bounds, vox_min = f32(b['bounds']).view(6), f32(b['vox_min']).view(3)
    vox_sh = (ct.c_int32 * 3)(*b['vox_sh'])
    P_, (Hf, Wf) = planes_f.shape[1], feat_f.shape[:2]
    A = lambda t: ct.c_void_p(t.data_ptr())
    st = ct.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    tok_ref, ext_ref = ws['tokens'].clone(), ws['extras'].clone()

    def bind(path):
        lib = ct.CDLL(path)
        for name, (ret, args) in _lib.parse_header(_lib.HEADER).items():
            if hasattr(lib, name):
                fn = getattr(lib, name); fn.restype = ret; fn.argtypes = [a[0] for a in args]
        return lib

    def mlp_only(lib, out):
        assert lib.sherf_nerf_mlp(A(ws['counters']), A(tok_ref), A(ext_ref), A(wc['stream']), A(wc['wbias']), MLP_PRECISIONS['f16x3'], cap, A(out), st) == 0

    def fused(lib, out, tok, ext):
        rc = lib.sherf_gather_mlp(A(ws['counters']), A(ws['geom']), A(planes_f), P_, A(feat_f), Hf, Wf, A(img4), b['H'], b['W'], last['levels_struct'],
                                  A(wc['tok_bias']), A(bounds), A(vox_min), vox_sh, A(wc['stream']), A(wc['wbias']), MLP_PRECISIONS['f16x3'], cap,
                                  A(tok), A(ext), A(out), st)
        assert rc == 0, rc

    prod = bind(os.path.join(ROOT, 'sherf_amd', 'libsherf_hip.so'))
    ref = torch.zeros_like(ws['sample_out'])
    mlp_only(prod, ref); torch.cuda.synchronize()
    ref = ref[:nv].clone()
    # control: the unfused launch against itself
    bad_c = 0
    for _ in range(reps):
        o = torch.zeros_like(ws['sample_out']); mlp_only(prod, o); torch.cuda.synchronize()
        bad_c += int(((o[:nv] - ref).abs().amax(1) > 0).sum())
    print(f'{cfg}: nv {nv} tiles {nt}; control sherf_nerf_mlp x{reps}: differing samples {bad_c}')
    libs = {'product': os.path.join(ROOT, 'sherf_amd', 'libsherf_hip.so')}
    for path in sorted(glob.glob(os.path.join(ROOT, 'sherf_amd', 'libsherf_hip_*.so'))):
        if os.path.basename(path) in ('libsherf_hip_bwd.so', 'libsherf_hip_ops.so'):
            continue
        libs[os.path.basename(path)[len('libsherf_hip_'):-3]] = path
    for tag, path in [(t, p) for t, p in libs.items()] + [('product+prefilled', libs['product'])]:
        lib = prod if tag.startswith('product') else bind(path)
        tiles_bad, tok_bad, maxd, first, self_bad = [], 0, 0.0, None, []
        for _ in range(reps):
            o = torch.zeros_like(ws['sample_out'])
            # 'prefilled': the token / extras buffers already hold the right values when the launch starts -- a stale read is then harmless
            tok, ext = (tok_ref.clone(), ext_ref.clone()) if True else None      # (every variant starts from the right tokens / extras: some do not write them)
            fused(lib, o, tok, ext); torch.cuda.synchronize()
            if first is None:
                first = o[:nv].clone()
            self_bad.append(len(np.unique(((o[:nv] - first).abs().amax(1) > 0).nonzero().flatten().cpu().numpy() // 32)))
            d = (o[:nv] - ref).abs().amax(1)
            bad = (d > 0).nonzero().flatten().cpu().numpy()
            tiles_bad.append(len(np.unique(bad // 32)))
            maxd = max(maxd, float(d.max()))
            tok_bad += int(not torch.equal(tok[:nt * 3072], tok_ref[:nt * 3072])) + int(not torch.equal(ext[:nt * 384], ext_ref[:nt * 384]))
        print(f'   {tag:10s} sherf_gather_mlp x{reps}: wrong tiles per launch vs the split launches {tiles_bad} of {nt}; vs its own first launch {self_bad}; token/extras mismatches {tok_bad}; max |diff| {maxd:.3e}')


if __name__ == '__main__':
    main()
