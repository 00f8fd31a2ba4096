"""The drop-in's real entry point at real size: `TriPlaneGenerator.forward(input_data, z, c, ...)` exactly as the reference's evaluation loop
calls it (sherf/training/test_loop.py:189-190) -- ResNet-18 code -> mapping network -> StyleGAN2 tri-plane synthesis at the size SHERF
instantiates (channel_base 32768, channel_max 512, 28.7 M parameters, planes [1, 96, 256, 256]; triplane.py:58, train.py:278-280) ->
ResNet-18 feature map of the 512 x 512 observation (triplane.py:320-343) -> per-frame glue (triplane.py:105-137, 174-217) ->
ImportanceRenderer.forward on 512 x 512 x 64 -- timed on one MI355X with a per-stage breakdown (HIP events on the caller's stream):

    python bench_generator.py [--steps K --warmup W --config cfg2_dense_ri]

Prints ONE JSON line: rays/s and ms of forward() (i) with every producer recomputed per frame and (ii) with `use_cached_backbone=True`
(triplane.py:99-104: the tri-planes of the previous frame are reused -- they depend on the observation image only), the stage table for both,
and which stage is the floor.  bench.py runs this file as a child and reports it as `secondary.generator_forward` (SURVEY 8(f) ranks 1-2 exist
"so that synthesis() end to end, not just renderer(), hits the rays/s target").  Weights: the reference constructors' own initialisation
(random, seeded): no pretrained pickle / torchvision weights exist offline; timings do not depend on the values."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STAGES = ('encoder_2d', 'mapping', 'backbone_synthesis', 'encoder_2d_feature', 'glue', 'renderer')


def build_generator(w, dev, small=False, num_fp16_res=0):
    """The generator of train.py:237-428 / training_loop.py:192 (z 512, c_dim 0, w 512, map_depth 2, cbase 32768, cmax 512, fp32) around the bench workload's renderer and
    decoder (reference-init network); its own producers, initialised by their constructors under a fixed seed."""
    from sherf_amd.triplane import TriPlaneGenerator
    from synthdata import synth
    torch.manual_seed(0)
    gen = TriPlaneGenerator(512, 0, 512, True, True, True, True, True, img_resolution=512, img_channels=3, mapping_kwargs=dict(num_layers=2),
                            rendering_kwargs=dict(w['opts']), smpl=synth.make_synth_smpl(0), channel_base=512 if small else 32768,
                            channel_max=16 if small else 512, num_fp16_res=num_fp16_res, conv_clamp=256 if num_fp16_res > 0 else None,   # (train.py:427-428)
                            fused_modconv_default='inference_only')
    gen.renderer, gen.decoder = w['rend'], w['dec']
    gen = gen.to(dev).eval()
    torch.manual_seed(0)
    w['rend'].train(); w['dec'].train()          # BatchNorm of the voxel encoder on batch statistics: the mode the reference renders in (bench.py --bn-mode)
    return gen


class _Mark:
    """A point on the caller's stream: a HIP event on the GPU; wall clock on the host build of the tests (every launch is synchronous there)."""
    on_gpu = True

    def __init__(self):
        self.e = torch.cuda.Event(enable_timing=True) if _Mark.on_gpu else None
        self.t = None

    def record(self):
        if self.e is not None:
            self.e.record()
        else:
            self.t = time.perf_counter()

    def elapsed_time(self, other):
        return self.e.elapsed_time(other.e) if self.e is not None else 1e3 * (other.t - self.t)


class StageClock:
    """HIP events around the generator's stages on the caller's stream: every wrapped call records (start, end); `glue` = from the end of the
    feature encoder to the start of the renderer.  One clock per forward()."""

    def __init__(self, gen):
        self.gen, self.ev, self.saved = gen, {}, []

    def _wrap(self, obj, attr, name):
        fn = getattr(obj, attr)

        def timed(*a, **k):
            e0, e1 = _Mark(), _Mark()
            e0.record()
            out = fn(*a, **k)
            e1.record()
            self.ev[name] = (e0, e1)
            return out
        self.saved.append((obj, attr, obj.__dict__.get(attr)))
        setattr(obj, attr, timed)

    def __enter__(self):
        g = self.gen
        # the three producers go through TriPlaneGenerator._producer (eager call or hipGraph replay): timed there, so that both modes are covered
        names = {'encoder_2d': 'encoder_2d', 'backbone.synthesis': 'backbone_synthesis', 'encoder_2d_feature': 'encoder_2d_feature'}
        orig = g._producer

        def producer(name, module, fn):
            inner = orig(name, module, fn)

            def timed(*a, **k):
                e0, e1 = _Mark(), _Mark()
                e0.record()
                out = inner(*a, **k)
                e1.record()
                self.ev[names[name]] = (e0, e1)
                return out
            return timed
        self.saved.append((g, '_producer', g.__dict__.get('_producer')))
        g._producer = producer
        self._wrap(g.backbone.mapping, 'forward', 'mapping')                 # (sub-modules: their `forward` is wrapped on the instance)
        self._wrap(g.renderer, 'forward', 'renderer')
        self.t0 = _Mark(); self.t1 = _Mark()
        self.t0.record()
        return self

    def __exit__(self, *exc):
        self.t1.record()
        for obj, attr, old in self.saved:
            if old is None:
                obj.__dict__.pop(attr, None)
            else:
                setattr(obj, attr, old)

    def read(self):
        out = {k: self.ev[k][0].elapsed_time(self.ev[k][1]) for k in self.ev}
        if 'encoder_2d_feature' in self.ev and 'renderer' in self.ev:
            out['glue'] = self.ev['encoder_2d_feature'][1].elapsed_time(self.ev['renderer'][0])
        out['total'] = self.t0.elapsed_time(self.t1)
        return out


def run(gen, d, dev, steps, warmup, cached):
    z, c = torch.randn(1, 512, generator=torch.Generator().manual_seed(1)).to(dev), torch.zeros(1, 25, device=dev)
    kw = dict(use_sr_module=False, test_flag=True, noise_mode='const')
    if cached:
        kw.update(cache_backbone=True, use_cached_backbone=True)

    def forward():
        with torch.no_grad():
            return gen(d, z, c, **kw)
    for _ in range(warmup):
        forward()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        out = forward()
    torch.cuda.synchronize(dev)
    ms = 1e3 * (time.perf_counter() - t0) / steps
    rows = []
    for _ in range(min(steps, 10)):                       # the stage table: separate forwards (the events' own cost stays out of `ms`)
        with StageClock(gen) as clk:
            forward()
        torch.cuda.synchronize(dev)
        rows.append(clk.read())
    stages = {k: float(np.mean([r[k] for r in rows if k in r])) for k in STAGES + ('total',) if any(k in r for r in rows)}
    return ms, stages, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--config', default='cfg2_dense_ri')
    ap.add_argument('--precision', default='auto')
    ap.add_argument('--small-backbone', action='store_true', help='TEST ONLY: channel_base 512 / channel_max 16 (the host-build dry run of this script)')
    ap.add_argument('--miopen-benchmark', action='store_true', help='torch.backends.cudnn.benchmark = True: MIOpen searches its convolution algorithms on first use')
    ap.add_argument('--eager-producers', action='store_true', help='TriPlaneGenerator.graph_producers = False in every variant (default: the product default, '
                                                                   'producers replayed as hipGraphs; the eager number is measured beside it either way)')
    ap.add_argument('--fp16-res', type=int, default=4, help='also time the forward with this many fp16 resolutions in the tri-plane generator (the reference\'s --g_num_fp16_res; 0: skip)')
    ap.add_argument('--channels-last', action='store_true', help='producers (backbone, encoders) in channels_last memory format')
    a = ap.parse_args()
    import bench
    lrank = int(os.environ.get('LOCAL_RANK', 0))
    if os.environ.get('SHERF_HIPCPU_LIB'):
        bench._use_host_build()
        _Mark.on_gpu = False
    dev = bench._device(lrank)
    w = bench.make_workload(argparse.Namespace(config=a.config, precision=a.precision, bn_mode='train', table_precision=None, encoder_precision=None), 0.4, dev)
    if a.miopen_benchmark:
        torch.backends.cudnn.benchmark = True
    gen = build_generator(w, dev, small=a.small_backbone)
    graph_default = bool(gen.graph_producers) and not a.eager_producers and dev.type == 'cuda'
    if a.channels_last:
        for m in (gen.backbone, gen.encoder_2d, gen.encoder_2d_feature):
            m.to(memory_format=torch.channels_last)
    d = w['d']
    R = d['ray_o_all'].shape[2]
    n_params = {k: int(sum(p.numel() for p in m.parameters())) for k, m in (('backbone', gen.backbone), ('encoder_2d', gen.encoder_2d),
                                                                             ('encoder_2d_feature', gen.encoder_2d_feature))}
    res = dict(metric='rays/s of TriPlaneGenerator.forward at 512x512x64 (test_loop.py:189-190), full-size producers', unit='rays/s', steps=a.steps, warmup=a.warmup,
               config=dict(workload=a.config, miopen_benchmark=bool(a.miopen_benchmark), channels_last=bool(a.channels_last), graph_producers=None, rays=R, image=list(d['obs_img_all'].shape[-2:]), backbone='StyleGAN2 Generator z512 w512 map_depth 2 channel_base 32768 channel_max 512 -> planes [1,96,256,256]',
                           parameters=n_params, weights='constructor initialisation under torch.manual_seed(0) (no pretrained pickle offline)',
                           mlp_precision=None, mlp_form=None))
    res['config']['graph_producers'] = graph_default
    for name, cached, graphed in (('recomputed_every_frame', False, graph_default), ('recomputed_every_frame_eager_producers', False, False),
                                  ('use_cached_backbone', True, graph_default)):
        if name.endswith('eager_producers') and not graph_default:
            continue                                        # (identical to the default variant)
        gen.graph_producers = graphed
        ms, stages, out = run(gen, d, dev, a.steps, a.warmup, cached)
        floor = max((k for k in STAGES if k in stages), key=lambda k: stages[k])
        res[name] = dict(ms_per_forward=ms, rays_per_s=R / (ms * 1e-3), stages_ms={k: round(v, 4) for k, v in stages.items()}, floor=floor,
                         floor_share=stages[floor] / max(stages.get('total', ms), 1e-9))
    img = out['image_raw']
    res['value'] = res['recomputed_every_frame']['rays_per_s']; res['ms_per_step'] = res['recomputed_every_frame']['ms_per_forward']
    if a.fp16_res > 0 and dev.type == 'cuda':
        # round 6: the same forward with the reference's OWN fp16 path in the tri-plane generator (train.py --g_num_fp16_res, networks_stylegan2.py:
        # 423-431, 471-536: the last `num_fp16_res` resolutions in fp16, conv_clamp 256) -- an option of the reference, not its shipped default (0).  Same seed: same weights.
        gen16 = build_generator(w, dev, small=a.small_backbone, num_fp16_res=a.fp16_res)
        gen16.graph_producers = graph_default
        ms16, stages16, out16 = run(gen16, d, dev, a.steps, a.warmup, False)
        i16 = out16['image_raw']
        res['recomputed_every_frame_backbone_fp16'] = dict(num_fp16_res=a.fp16_res, conv_clamp=256, ms_per_forward=ms16, rays_per_s=R / (ms16 * 1e-3),
                                                           stages_ms={k: round(v, 4) for k, v in stages16.items()}, finite=bool(torch.isfinite(i16).all()),
                                                           image_rel_diff_vs_fp32_backbone=float((i16 - img).abs().max() / (img.abs().max() + 1e-12)))
        del gen16
    res['output'] = dict(image_raw=list(img.shape), finite=bool(torch.isfinite(img).all()), mean=float(img.mean()), weights_mean=float(out['weights_image'].mean()))
    res['config']['mlp_precision'] = w['rend'].last.get('mlp_precision'); res['config']['mlp_form'] = w['rend'].last.get('mlp_form')
    res['renderer_ms_inside_forward'] = res['recomputed_every_frame']['stages_ms'].get('renderer')
    if graph_default:
        res['graphed'] = {k: ('off: ' + getattr(v, 'error', '?')) if v.off else 'captured' for k, v in gen.__dict__.get('_graphed', {}).items()}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
