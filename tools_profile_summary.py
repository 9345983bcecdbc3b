"""Summarise a tools_profile.sh output directory into profiles/<tag>_*.{csv,md}."""
import collections, csv, glob, json, os, sys
src, tag = sys.argv[1], sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 4   # warmup + timed steps executed under the profiler
os.makedirs("profiles", exist_ok=True)
stats = list(csv.DictReader(open(glob.glob(f"{src}/trace/*/*_kernel_stats.csv")[0])))
with open(f"profiles/{tag}_kernel_stats.csv", "w") as f:
    w = csv.writer(f); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in stats: w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
def pmc(kind, counter):
    fs = glob.glob(f"{src}/pmc_{kind}/*/*_counter_collection.csv")
    agg = collections.defaultdict(lambda: [0.0, 0])
    if not fs: return agg
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == counter:
            a = agg[r["Kernel_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    return agg
fetch, write = pmc("fetch", "FETCH_SIZE"), pmc("write", "WRITE_SIZE")
lines = [f"# {tag}: rocprofv3 --kernel-trace --stats summary + HBM traffic (separate --pmc passes)", "",
         "FETCH_SIZE is doubled (gfx950 counts 128-B requests as 64 B, MI355X_MICROARCH.md HBM section); units KiB -> bytes.", "",
         "| kernel | calls | ms/step | avg us | % | HBM read MB/launch (2x FETCH_SIZE) | HBM write MB/launch |", "|---|---|---|---|---|---|---|"]
for r in stats[:16]:
    n = r["Name"]
    fr = fetch.get(n); wr = write.get(n)
    frs = f"{2 * fr[0] * 1024 / fr[1] / 1e6:.1f}" if fr and fr[1] else "-"
    wrs = f"{wr[0] * 1024 / wr[1] / 1e6:.1f}" if wr and wr[1] else "-"
    lines.append(f"| `{n[:70]}` | {r['Calls']} | {int(r['TotalDurationNs']) / steps / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.1f} | {r['Percentage']} | {frs} | {wrs} |")
log = open(f"{src}/trace.log").read().strip().splitlines()[-1]
try:
    js = json.loads(log); lines += ["", "bench line under the profiler:", "```", json.dumps(js), "```"]
except Exception:
    pass
open(f"profiles/{tag}_summary.md", "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:24]))
