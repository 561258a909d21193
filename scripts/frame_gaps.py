"""Diagnostic: where the registration stream idles inside a frame, from a rocprofv3 kernel trace of bench.py: for the
queue that runs k_search_all, the idle time in front of every kernel, summed per frame by the kernel it precedes."""
import csv, glob, re, sys, collections
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    return re.sub(r"\(anonymous namespace\)::|void |lsa::", "", n).split("(")[0].split("<")[0][:28]
qs = collections.Counter(r["Queue_Id"] for r in rows if "k_search_all" in r["Kernel_Name"])
q = qs.most_common(1)[0][0]
mine = [r for r in rows if r["Queue_Id"] == q]
# frames: a k_loc_start marks the middle of every frame; use the last 30
starts = [i for i, r in enumerate(mine) if "k_loc_start" in r["Kernel_Name"]]
lo, hi = starts[-31], starts[-1]
gaps, durs = collections.defaultdict(float), collections.defaultdict(float)
prev_end, prev_name = None, None
for r in mine[lo:hi]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = short(r["Kernel_Name"])
    if prev_end is not None:
        gaps[(prev_name, n)] += max(0, s - prev_end) / 1e3
    durs[n] += (e - s) / 1e3
    prev_end, prev_name = max(prev_end or 0, e), n
nf = 30
span = (int(mine[hi]["Start_Timestamp"]) - int(mine[lo]["Start_Timestamp"])) / 1e3 / nf
print("frame %.1f us on queue %s; busy %.1f us, idle %.1f us" % (span, q, sum(durs.values()) / nf, sum(gaps.values()) / nf))
for k, v in sorted(gaps.items(), key=lambda kv: -kv[1])[:14]:
    print("  idle %6.1f us/frame between %-28s and %s" % (v / nf, k[0], k[1]))
for k, v in sorted(durs.items(), key=lambda kv: -kv[1])[:10]:
    print("  busy %6.1f us/frame %s" % (v / nf, k))
