"""Run bench.py in N fresh processes with SGNN_BENCH_CHECKSUMS=1 and compare the per-step losses and the final parameters
bit for bit (the driver never touches the GPU).  usage: python tools/bench_repeat.py N [bench.py args ...]"""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1])
extra = sys.argv[2:]
runs = []
for i in range(n):
    env = dict(os.environ, SGNN_BENCH_CHECKSUMS=os.environ.get('SGNN_BENCH_CHECKSUMS', '1'))
    r = subprocess.run([sys.executable, os.path.join(REPO, 'bench.py'), '--no-cpu-baseline', '--no-extras'] + extra, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.DEVNULL, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith('{')]
    if not line:
        print('run', i, 'failed', r.returncode)
        continue
    d = json.loads(line[-1])
    runs.append(d['checksums'])
    print('run', i, 'ms', round(d['ms_per_step'], 3), 'final loss', repr(d['loss']), flush=True)
ref = runs[0]
for i, c in enumerate(runs[1:], 1):
    if c == ref:
        continue
    step = next((k for k, (a, b) in enumerate(zip(ref['loss_bits'], c['loss_bits'])) if a != b), None)
    bad = [k for k in ref['params'] if ref['params'][k] != c['params'][k]]
    for pi, (pa, pb) in enumerate(zip(ref.get('passes', []), c.get('passes', []))):
        if pa != pb:
            names = [k for k in pa['grads'] if pa['grads'][k] != pb['grads'].get(k)]
            same = [k for k in pa['grads'] if pa['grads'][k] == pb['grads'].get(k) and not k.startswith(('prep/', 'param_before/'))]
            print('   differing prepared tensors / parameters before the pass / forward values:',
                  [k for k in names if k.startswith(('prep/', 'param_before/', 'fwd/'))][:20])
            print('run %d: first differing pass (priming passes included) %d: loss differs %s; %d of %d gradients differ; EQUAL: %s'
                  % (i, pi, pa['loss'] != pb['loss'], len(names), len(pa['grads']), same))
            break
    print('run %d differs from run 0: first differing timed step %s (losses %r vs %r); %d of %d parameters differ: %s'
          % (i, step, ref['losses'][step] if step is not None else None, c['losses'][step] if step is not None else None,
             len(bad), len(ref['params']), bad[:6]))
print('distinct outcomes:', len({json.dumps(c, sort_keys=True) for c in runs}), 'of', len(runs))
