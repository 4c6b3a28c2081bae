"""GPU-box helper: one bench.py run, the figures an A/B needs on one line.  usage: python tools/gpu/ab_line.py <label> [bench.py args...]"""
import json, subprocess, sys
label, args = sys.argv[1], sys.argv[2:]
out = subprocess.run([sys.executable, 'bench.py', '--no-cpu-baseline', '--no-8b'] + args, capture_output=True, text=True).stdout.strip().splitlines()
d = json.loads(out[-1])
ph, qa, sft = d.get('phases') or {}, d.get('qa') or {}, d.get('sft') or {}
print(label, 'chunk_ms', d.get('ms_per_step'), 'euler_us', ph.get('euler_us_per_layer_step'), 'vit', ph.get('vit_ms'), 'prefill', ph.get('prefill_ms'),
      'frac', (d.get('roofline') or {}).get('frac'), 'dec1', (qa.get('batch1') or {}).get('decode_ms_per_step'), 'dec8', (qa.get('batch8') or {}).get('decode_ms_per_step'),
      'sft', sft.get('ms_per_step'), 'fwd_bwd', sft.get('fwd_bwd_ms'), 'b4', (d.get('batched') or {}))
