#!/bin/bash
# kernel trace of a stand-in's replayed training steps: kernels per step and time by family
export TMPDIR=/tmp
C=${1:-ppi_bp}
D=gpurun_out/sk_prof
rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 tools/bench_standin.py --config $C > gpurun_out/sk_$C.json 2> gpurun_out/sk.err
KT=$(find $D -name "*kernel_trace.csv" | head -1)
python - <<PY
import csv, re, collections
rows = list(csv.DictReader(open("$KT")))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the last 2000 kernels are replayed steps (30 timed replays at the end)
tail = rows[-6000:]
names = [r['Kernel_Name'] for r in tail]
# period detection: find the step length by the repetition of a distinctive kernel
key = 'nll_loss_forward' if any('nll_loss_forward' in n for n in names) else 'binary_cross'
idx = [i for i, n in enumerate(names) if key in n]
per = idx[-1] - idx[-2] if len(idx) > 2 else len(names)
step = tail[idx[-2]:idx[-1]]
tot = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in step) / 1e3
span = (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e3
print('$C: kernels per step', per, 'kernel time %.0f us, span %.0f us' % (tot, span))
c = collections.Counter(); n = collections.Counter()
for r in step:
    nm = re.sub(r'void |at::native::|\(anonymous namespace\)::|rocprim::ROCPRIM_\d+_NS::detail::', '', r['Kernel_Name'])
    k = 'GEMM' if nm.startswith('Cijk') else re.sub(r'<.*', '', nm)[:44]
    c[k] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3; n[k] += 1
for k, v in c.most_common(28):
    print('%8.1f us %4d  %s' % (v, n[k], k))
PY
rm -rf $D
