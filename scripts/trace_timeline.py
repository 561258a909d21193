"""Diagnostic: timeline of the kernels of a few frames from a rocprofv3 kernel trace (queue, start, duration, gap)."""
import csv, glob, re, sys
f = sorted(glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = re.sub(r"\(anonymous namespace\)::|void |lsa::", "", n)
    n = re.sub(r"rocprim::ROCPRIM_\d+_NS::detail::", "rp:", n)
    return n.split("(")[0][:34]
# the last but 3rd k_add_commit ... pick a window of ~2 frames near the end
commits = [i for i, r in enumerate(rows) if "k_add_commit" in r["Kernel_Name"]]
lo = commits[-7] if len(commits) > 7 else 0
hi = commits[-3] if len(commits) > 3 else len(rows) - 1
t0 = int(rows[lo]["Start_Timestamp"])
last_end = {}
for r in rows[lo:hi + 1]:
    q = r["Queue_Id"]
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = s - last_end.get(q, s)
    last_end[q] = e
    print("q%-3s %9.1f us  +%7.1f dur %6.1f  gap %7.1f  %s" % (q, s / 1e3, 0, (e - s) / 1e3, gap / 1e3, short(r["Kernel_Name"])))
