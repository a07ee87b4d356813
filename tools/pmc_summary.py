#!/usr/bin/env python3
"""Per-kernel averages of the counters in rocprofv3 --pmc output (``--output-format csv``): every
``*counter_collection.csv`` under the given directories.  Prints one JSON object
{kernel (short name): {counter: {"mean": v, "n": dispatches}}}.

    python tools/pmc_summary.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_x_pmc.json
"""
import csv
import json
import os
import re
import sys


def short(name):
    m = re.match(r'(?:void\s+)?([A-Za-z0-9_:]+(?:<[^(]*>)?)', name)
    return (m.group(1) if m else name)[:120]


def main(dirs):
    acc = {}
    for d in dirs:
        for root, _, files in os.walk(d):
            for f in files:
                if not f.endswith('counter_collection.csv'):
                    continue
                with open(os.path.join(root, f), newline='') as fh:
                    for row in csv.DictReader(fh):
                        k = short(row['Kernel_Name'])
                        c = row['Counter_Name']
                        v = float(row['Counter_Value'])
                        a = acc.setdefault(k, {}).setdefault(c, [0.0, 0])
                        a[0] += v
                        a[1] += 1
    out = {k: {c: {'mean': s / n, 'n': n} for c, (s, n) in cs.items()} for k, cs in acc.items()}
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == '__main__':
    main(sys.argv[1:])
