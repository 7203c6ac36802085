"""Reduce a rocprofv3 kernel_stats.csv to per-kernel and per-family ms/step.  Usage: kstats.py <csv> <steps-in-trace> [min_ms]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.12
print("total kernel ms/step %.2f" % (sum(float(r['TotalDurationNs']) for r in rows) / steps / 1e6))
fam = {}
for r in rows:
    n = r['Name']; t = float(r['TotalDurationNs']) / steps / 1e6; c = int(r['Calls']) / steps
    short = re.sub(r'\(anonymous namespace\)::|void |bdgemm::', '', n)
    short = re.sub(r'\(.*', '', short)[:110]
    if t > min_ms:
        print(f"{t:7.3f} ms {c:6.1f} calls {float(r['AverageNs'])/1e3:8.1f} us  {short}")
    key = ('igemm' if 'igemm_kernel' in n else 'hconv' if 'hconv_kernel' in n else 'sgemm' if 'sgemm_kernel' in n else 'bn' if re.search(r'bn_|BnBwd|sum_partials|fold_partials|StatFn|stem_|StemBwd', n)
           else 'p16pack' if 'p16_' in n else 'attn' if 'attn' in n else 'other')
    fam[key] = fam.get(key, 0) + t
print({k: round(v, 2) for k, v in sorted(fam.items())})
