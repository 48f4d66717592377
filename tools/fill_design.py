#!/usr/bin/env python3
"""Fill the {{PLACEHOLDER}} figures of tools/design_front.md from the bench lines under profiles/ (so that DESIGN.md quotes the files, not a
memory of them):  python tools/fill_design.py r05 [--targets "text"] > /tmp/front_filled.md ; then tools/assemble_design.py."""
import argparse
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument('tag')
ap.add_argument('--targets', default=None)
a = ap.parse_args()
P = os.path.join(ROOT, 'profiles')


def line(name):
    return json.loads(open(os.path.join(P, name)).read().strip().splitlines()[-1])


HB = 256          # the headline per-GPU batch (bench.py default, tools/measure_round.sh)
final = line(f'{a.tag}_bench_b{HB}_final.json')
first = line(f'{a.tag}_bench_b{HB}.json')
v = {'TAG': a.tag, 'HEAD_IPS': f"{final['value']:.0f}", 'HEAD_MS': f"{final['ms_per_step']:.2f}", 'HEAD_IPS_A': f"{first['value']:.0f}",
     'HEAD_X': f"{final['value'] / 200:.0f}"}
for b in (4, 16, 32, 128):
    v[f'B{b}'] = f"{line(f'{a.tag}_bench_b{b}.json')['value']:.0f}"
v['B128_MS'] = f"{line(f'{a.tag}_bench_b128.json')['ms_per_step']:.1f}"
v['HEAD_MS_HALF'] = f"{final['ms_per_step'] * 128 / HB:.1f}"
r, g = final['roofline'], final['roofline_gemm']
v.update(LOSS_GBPS=f"{r['achieved']:.0f}", LOSS_FRAC=f"{r['frac']:.3f}", LOSS_TRAFFIC=f"{(r.get('traffic') or 0) / 1e9:.2f}",
         LOSS_TOA=str(r.get('traffic_over_algorithmic')), GEMM_GBPS=f"{g['achieved']:.0f}", GEMM_FRAC=f"{g['frac']:.2f}",
         GEMM_TOA=str(g.get('traffic_over_algorithmic')))
cfg = {'CFG3': 'cfg3_b64', 'CFG3F': 'cfg3_b64_fp8', 'CFG4': 'cfg4_b32', 'CFG5': 'cfg5_b32', 'CFG5F': 'cfg5_b32_fp8'}
vals = {}
for k, f in cfg.items():
    d = line(f'{a.tag}_bench_{f}.json')
    vals[k] = d['value']
    v[k] = f"{d['value']:.1f}"
    if k == 'CFG3':
        v['CFG3_TF'] = f"{d['roofline']['achieved']:.0f}"
goal = {'CFG3': 320, 'CFG3F': 430, 'CFG4': 160, 'CFG5': 100, 'CFG5F': 120}
met = [k for k in goal if vals[k] >= goal[k]]
names = {'CFG3': 'cfg3', 'CFG3F': 'cfg3 fp8', 'CFG4': 'cfg4', 'CFG5': 'cfg5', 'CFG5F': 'cfg5 fp8'}
v['TARGETS'] = a.targets or ('met for ' + (', '.join(names[k] for k in met) or 'none') + '; missed for '
                             + ', '.join(f'{names[k]} ({vals[k]:.0f})' for k in goal if k not in met))
tl = {leg['per_gpu_batch']: leg for leg in final['train_loop'] if leg['leg'] == 'train_loop'}
v['TLH'], v['TL4'] = f"{tl[HB]['ratio_to_replay_only']:.3f}", f"{tl[4]['ratio_to_replay_only']:.3f}"
cli = next(leg for leg in final['train_loop'] if leg['leg'] == 'default_cli')
v['CLI_AUTO'], v['CLI_EAGER'] = f"{cli['auto']['images_per_sec']:.0f}", f"{cli['eager']['images_per_sec']:.0f}"
ev = {leg['per_gpu_batch']: leg for leg in final['eval']}
v['EV1_F32'], v['EV32_F32'] = f"{ev[1]['fp32']['graph']['images_per_sec']:.0f}", f"{ev[32]['fp32']['graph']['images_per_sec']:.0f}"
v['EV1_BF'], v['EV32_BF'] = f"{ev[1]['bf16']['graph']['images_per_sec']:.0f}", f"{ev[32]['bf16']['graph']['images_per_sec']:.0f}"

# socket power / shader clock / energy per image during each configuration's replay (tools/measure_round.sh, section power)
for ln in open(os.path.join(P, f'{a.tag}_power_clock.txt')):
    m = re.match(r'(cfg\d) batch \d+: \d+ samples, mean power (\d+) W, mean sclk (\d+) MHz.*?(?:, ([\d.]+) J per image)?$', ln.strip())
    if m:
        c = m.group(1).upper()
        v[f'{c}_W'], v[f'{c}_GHZ'], v[f'{c}_J'] = m.group(2), f'{int(m.group(3)) / 1000:.2f}', m.group(4) or '?'
src = open(os.path.join(ROOT, 'tools', 'design_front.md')).read()
missing = sorted(set(re.findall(r'\{\{(\w+)\}\}', src)) - set(v))
if missing:
    sys.exit(f'no value for {missing}')
sys.stdout.write(re.sub(r'\{\{(\w+)\}\}', lambda m: v[m.group(1)], src))
