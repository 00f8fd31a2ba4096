"""Stage-by-stage diagnostic of the HIP path against the oracle (prints errors, never asserts). GPU box only."""
import sys, os, time, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import sherf_oracle as O
from tests import gpu_common as G


def p(*a):
    print(*a, flush=True)


def stage(name, fn):
    try:
        fn()
    except Exception:
        p(f'[{name}] EXCEPTION'); traceback.print_exc(); sys.stdout.flush()


def main(cfg='tiny', prec='f16x3'):
    p('device', torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).gcnArchName)
    t0 = time.time(); o = G.oracle_render(cfg); p(f'oracle {cfg} done in {time.time()-t0:.1f}s, nv={o["valid"].numel()}')
    h = G.hip_render(cfg, precision=prec)
    ws = h['last']['ws']; nv_o = o['valid'].numel()
    nv = int(ws['counters'][0]); p('nv hip', nv, 'oracle', nv_o, 'dmin/dmax ord', ws['counters'][1:3].tolist())
    n = min(nv, nv_o)

    def s_mask():
        p('cs_idx equal', torch.equal(ws['cs_idx'][:n].cpu().long(), o['valid'][:n]))
        p('vid mismatches', int((ws['cs_vid'][:n].cpu().long() != o['vert_id'][:n]).sum()), 'tvid mismatches', int((ws['cs_tvid'][:n].cpu().long() != o['t_vert_id'][:n]).sum()))
        p('x_s max err', float((ws['cs_xs'][:n, :3].cpu() - o['x_s'][:n]).abs().max()))
    stage('mask', s_mask)

    def s_geom():
        g = ws['geom'][:n].cpu()
        p('x_c err', float((g[:, :3] - o['x_c'][:n]).abs().max()), 'v_c err', float((g[:, 3:6] - o['v_c'][:n]).abs().max()), 'uv rel', G.rel(g[:, 6:8], o['uv'][:n]))
    stage('geom', s_geom)

    def s_vox():
        vd = h['last']['vox']
        for (keys, feats, shape), (lev, raw, bnp, C) in zip(o['taps'], vd['taps']):
            L = vd['levels'][lev]; m = int(L['n_rows'][0])
            ke = m == keys.numel() and torch.equal(L['keys'][:m].cpu().long(), keys)
            mm = min(m, keys.numel())
            act = torch.relu(raw[:mm] * bnp[0] + bnp[1]).cpu()
            p(f'level {lev}: rows hip {m} oracle {keys.numel()} keys_equal {ke}', 'feat rel', G.rel(act, feats[:mm]))
    stage('vox', s_vox)

    def s_tok():
        tok = G.untile_tokens(ws['tokens'].cpu(), n); ex = G.untile_extras(ws['extras'].cpu(), n)
        st = G.seeded_state(); Wb = st['renderer.conv1d_reprojection.weight'][:, 32:64, 0]
        ref = o['tokens_in'][:n].clone(); ref[:, 2] -= O.positional_encoding(o['tap_rgb'][:n], 5)[:, :32] @ Wb.t()
        for s in range(3):
            p(f'token slot {s} rel', G.rel(tok[:, s], ref[:, s]))
        p('extras rgb err', float((ex[:, 6:9] - o['tap_rgb'][:n]).abs().max()), 'xc err', float((ex[:, :3] - o['x_c'][:n]).abs().max()), 'vc err', float((ex[:, 3:6] - o['v_c'][:n]).abs().max()))
    stage('tokens', s_tok)

    def s_mlp():
        for pr in ('f16x3', 'bf16'):
            hh = h if pr == prec else G.hip_render(cfg, precision=pr)
            out = hh['last']['ws']['sample_out'][:n].cpu()
            sr = torch.relu(o['sample_sigma'][:n])
            p(pr, 'sigma raw rel', G.rel(out[:, 3], o['sample_sigma'][:n]), 'sigma+ rel-to-max', float((torch.relu(out[:, 3]) - sr).abs().max() / sr.max()),
              'rgb max abs', float((out[:, :3] - o['sample_rgb'][:n]).abs().max()), 'nan', int(torch.isnan(out).sum()))
    stage('mlp', s_mlp)

    def s_final():
        p('rgb rel', G.rel(h['rgb'], o['rgb']), 'acc rel', G.rel(h['acc'], o['acc']), 'depth max abs', float((h['depth'] - o['depth']).abs().max()), 'psnr', O.psnr(h['rgb'], o['rgb']))
    stage('final', s_final)


if __name__ == '__main__':
    main(*(sys.argv[1:]))
