#!/usr/bin/env python3
"""dev helper: fill the @PLACEHOLDERS@ of a DESIGN.md template from a bench.py JSON line.  usage: fill_design.py template bench.json out"""
import json, sys
tpl = open(sys.argv[1]).read()
d = json.load(open(sys.argv[2]))
r = d['roofline']; c = d.get('configs', {})
rows = sorted(r['per_kernel'].items(), key=lambda kv: -kv[1]['ms_per_step'])
tab = ['| class | ms / step | TFLOP/s | bound | frac | algorithmic GB / step |', '|---|---|---|---|---|---|']
for k, v in rows[:8]:
    bound = 'MFMA' if v['bound'] == 'mfma' else 'HBM (%.2f TB/s)' % (v['achieved'] / 1e3)
    tab.append('| %s | %.2f | %s | %s | %.2f | %.1f |' % (k.replace('elementwise', 'element-wise'), v['ms_per_step'], ('%.0f' % v['tflops']) if v['tflops'] else '—', bound, v['frac'], v['algorithmic_gb_per_step']))
rep = {
    'VALUE': '%.1f' % d['value'], 'MS': '%.1f' % d['ms_per_step'], 'ALGTF': '%.0f' % d['pass_tflops_algorithmic'],
    'FP32FRAC': '%.0f' % (100 * r['step']['frac_of_fp32_mfma_peak']), 'SWEEP': '%.1f' % d['sweep_fwd_ms'],
    'C4MS': '%.0f' % c['config4_share_8_scenes']['ms_per_step'], 'C4PS': '%.1f' % c['config4_share_8_scenes']['passes_per_s'],
    'C3MS': '%.1f' % c['config3_three_heads_bf16']['ms'], 'C3F32': '%.1f' % c['config3_three_heads_bf16']['ms_fp32_class'],
    'C5MS': '%.1f' % c['config5_share_1824_fp16']['ms_per_step'], 'C5F32': '%.1f' % c['config5_share_1824_fp16']['ms_fp32_class'],
    'CPUS': '%.1f' % d['cpu_baseline']['seconds_per_pass'], 'CPUSWEEP': '%.1f' % d['cpu_baseline'].get('sweep_fwd_seconds', float('nan')), 'HOSTMS': '%.1f' % d.get('ms_per_step_host_inputs', float('nan')), 'ENQ': '%.1f' % d.get('host_enqueue_ms_per_step', float('nan')), 'TABLE': '\n'.join(tab), 'STEPFRAC': '%.2f' % r['step']['frac'],
}
for k, v in rep.items(): tpl = tpl.replace('@' + k + '@', v)
open(sys.argv[3], 'w').write(tpl)
