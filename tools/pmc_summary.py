"""Summarise rocprofv3 --pmc CSV output (counter_collection.csv files) into a per-kernel / per-counter table."""
import csv
import glob
import re
import sys
from collections import defaultdict


def summarise(paths, kernel_filter=None):
    acc = defaultdict(list)
    for p in paths:
        with open(p, newline='') as f:
            for row in csv.DictReader(f):
                name = row.get('Kernel_Name', '')
                if kernel_filter and not re.search(kernel_filter, name):
                    continue
                acc[(name, row['Counter_Name'])].append(float(row['Counter_Value']))
    out = ['| kernel | counter | launches | mean per launch |', '|---|---|---|---|']
    for (name, ctr), vals in sorted(acc.items()):
        short = re.sub(r'\(.*', '', name)[:70]
        out.append('| %s | %s | %d | %.4g |' % (short, ctr, len(vals), sum(vals) / len(vals)))
    return '\n'.join(out)


if __name__ == '__main__':
    root = sys.argv[1]
    flt = sys.argv[2] if len(sys.argv) > 2 else None
    print(summarise(glob.glob(root + '/**/*counter_collection.csv', recursive=True), flt))
