"""GPU busy / idle time and per-family wall-clock attribution of the last step in a rocprofv3 --kernel-trace CSV of bench.py
(a step ends with adam_kernel).  usage: python tools/trace_idle.py <dir with *_kernel_trace.csv>"""
import collections, csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
adam = [i for i, e in enumerate(ev) if e[2].startswith("adam_kernel")]
seg = ev[adam[-2] + 1:adam[-1] + 1]
def cat(n):
    return "gemm" if "gemm" in n else "attn" if "attn" in n else "other"
pts = sorted(set([e[0] for e in seg] + [e[1] for e in seg]))
tot = collections.Counter()
for a, b in zip(pts[:-1], pts[1:]):
    cats = set(cat(e[2]) for e in seg if e[0] <= a and e[1] >= b)
    tot["+".join(sorted(cats)) if cats else "idle"] += b - a
span = seg[-1][1] - seg[0][0]
print(f"step span {span / 1e6:.3f} ms, {len(seg)} kernels")
for k, v in tot.most_common():
    print(f"  {k:14s} {v / 1e6:8.3f} ms  {100.0 * v / span:5.1f} %")
