"""Per-op HIP-event timing of the key and cur plans of one model: python scripts/microbench/prof_ops.py 18 [clips per call] [HxW]
(ACCEL_CONV_DTYPE=f16 for the fp16 mode: `ACCEL_CONV_DTYPE=f16 python scripts/microbench/prof_ops.py 50 1 2048x4096` is config 5)"""
import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from accel_amd import demo, runtime
from accel_amd.config.config import config, update_config
from accel_amd.core import tester
from accel_amd.utils import synth
update_config(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..', 'tests', 'golden', 'dff_deeplab_vid_demo.yaml'))
ver = sys.argv[1] if len(sys.argv) > 1 else '18'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
H, W = [int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else '1024x2048').split('x')]
config.SCALES[0] = (H, W)
arg, aux = synth.model_params(ver, H, W, config)
runner = demo.ClipRunner(ver, config, arg, aux, (H, W), batch=B)
for nm, pred in (('key', runner.key_predictor), ('cur', runner.cur_predictor)):
    plan, lw = pred.plan_for(H, W, B)
    ms = plan.profile(3)
    ops = plan.ops()
    tot = ms.sum()
    print('==== %s plan: %d ops, %.3f ms' % (nm, len(ops), tot))
    rows = []
    for (kind, args), o, t in zip(lw.ops, ops, ms):
        tf = o['flops'] / (t * 1e-3) / 1e12 if o['flops'] else 0
        gbs = o['bytes'] / (t * 1e-3) / 1e9 if o['bytes'] else 0
        extra = ''
        if kind == 'conv':
            extra = '%s%s%s ' % ('x:h ' if args['in'].buf.esize == 2 else '', 'y:h ' if args['out'].buf.esize == 2 else '', 'r:h' if 'res' in args and args['res'].buf.esize == 2 else '') if lw.half_bufs else ''
            extra += 'cin=%s cout=%s k=%s s=%s %s out=%s tile=%s ksplit=%d' % (args.get('cin'), args.get('cout'), args.get('k'), args.get('s'), args.get('mode'), args['out'].ref().split(':', 2)[2], 'narrow' if o['narrow'] else o['tile'], o['ksplit'])
        rows.append((t, '%-10s %-28s %8.1f us %6.1f TF %7.0f GB/s  %s' % (kind, o['name'][:28], t * 1e3, tf, gbs, extra)))
    for t, r in rows:
        print(r)
