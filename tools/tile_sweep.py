"""Per-layer times of tools/p16_bench.py under forced sgemm tiles (BDETR_STILE), side by side: which tile each launch class would
want.  Usage: python tools/tile_sweep.py   (runs p16_bench once per tile in a child process)"""
import os
import re
import subprocess
import sys

TILES = ["auto", "128x128", "128x64", "64x64"]
rows = {}
for t in TILES:
    env = dict(os.environ)
    env.pop("BDETR_STILE", None)
    if t != "auto":
        env["BDETR_STILE"] = t
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "p16_bench.py"), "16", "p16"],
                         env=env, capture_output=True, text=True).stdout
    for line in out.splitlines():
        m = re.match(r"\s*(\d+x\d+\s+C\d+\s+K\d+\s+\dx\d s\d) x\d+\s+[\d.]+ \|.*?\|\s+([\d.]+)\(\s*\d+\)\s+([\d.]+)\(\s*\d+\)\s+([\d.]+)\(", line)
        if m:
            rows.setdefault(m.group(1), {})[t] = tuple(float(m.group(i)) for i in (2, 3, 4))
print(f"{'layer':30s} " + "   ".join(f"{t:>22s}" for t in TILES) + "    (fwd / dgrad / wgrad ms)")
for k, v in rows.items():
    print(f"{k:30s} " + "   ".join("/".join(f"{x:6.3f}" for x in v.get(t, (0, 0, 0))) for t in TILES))
