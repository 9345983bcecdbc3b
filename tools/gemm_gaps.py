"""Summarise a tools/gemm_bench.py log: the dispatcher's choice (v0) against the best forced variant of each shape."""
import re, sys
for l in open(sys.argv[1]):
    vs = re.findall(r"v(\d+):\s+([\d.]+) us\s+(\d+) TF", l)
    if not vs: continue
    best = min(vs[1:], key=lambda x: float(x[1])); v0 = vs[0]
    same = [x[0] for x in vs[1:] if abs(float(x[1]) - float(v0[1])) / float(v0[1]) < 0.015]
    print(f"{l[:38]:38s} auto {v0[1]:>7s} us  best v{best[0]:>3s} {best[1]:>7s} us  gap {100 * (float(v0[1]) / float(best[1]) - 1):5.1f}%  auto~v{same}")
