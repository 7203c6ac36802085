"""Average the counters of the GEMM-family dispatches (igemm_kernel / sgemm_kernel) in rocprofv3 --pmc output directories."""
import collections
import csv
import glob
import os
import sys

for d in sys.argv[1:]:
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "igemm_kernel" not in r["Kernel_Name"] and "sgemm_kernel" not in r["Kernel_Name"]:
                continue
            a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
    print(d, {k_: round(v[1] / v[0], 1) for k_, v in acc.items()})
