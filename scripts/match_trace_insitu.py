"""Timeline of the hardware blocks of the LAST fused match launch of a replayed sequence (as bench.py runs it)."""
import os, sys
os.environ["LSA_ROUTE_STATS"] = "1"
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lidarslam_amd as L

lookahead = int(sys.argv[1]) if len(sys.argv) > 1 else 1
slam = L.Slam(0, EgoMotion=3)
n = 30
stamps = []
for f in range(n):
    pts, stamp = L.synth_frame(128, 1000, f)
    slam.store_frame(f, pts)
    stamps.append(stamp)
ctx = slam.context()
for f in range(n):
    if lookahead and f + 1 < n:
        slam.hint_next_stored_frame(f + 1)
    slam.add_stored_frame(f, stamps[f], f)
ctx.sync()
tr = ctx.match_trace(2048)
ok = tr[:, 0] > 0
# only the blocks of the last launch: those that started within 1 ms of the latest start
last = tr[ok, 0].max()
ok &= tr[:, 0] + 100000 > last
t0 = tr[ok, 0].min()
start, mid, end = (tr[:, 0] - t0) / 100.0, (tr[:, 1] - t0) / 100.0, (tr[:, 2] - t0) / 100.0
print("lookahead", lookahead, "blocks", int(ok.sum()), "span us %.1f" % end[ok].max(), "search p50/max %.1f %.1f" % (np.median((mid - start)[ok]), (mid - start)[ok].max()),
      "model p50/max %.1f %.1f" % (np.median((end - mid)[ok]), (end - mid)[ok].max()), "start p50/max %.1f %.1f" % (np.median(start[ok]), start[ok].max()))
for t in range(0, int(end[ok].max()) + 10, 10):
    print("t=%3d us running blocks %4d" % (t, int(((start <= t) & (end > t) & ok).sum())))

lt = ctx.solve_device_trace().astype(np.float64)
if lt[6] > 0:
    print("LM solves %d, evaluations/solve %.2f, in-kernel us/solve %.1f; per evaluation us: evaluate %.2f exchange %.2f fold %.2f step %.2f" % (
        lt[6], lt[4] / lt[6], lt[5] / lt[6] / 100, lt[0] / lt[4] / 100, lt[1] / lt[4] / 100, lt[2] / lt[4] / 100, lt[3] / lt[4] / 100))
    print("   of evaluate: residual blocks %.2f us" % (lt[7] / lt[4] / 100))
    print("   of step: decision %.2f, scaled system %.2f, Cholesky solve %.2f, model change + candidate %.2f us" % tuple(lt[8 + i] / lt[4] / 100 for i in range(4)))

dur = (mid - start)
idx = np.argsort(-np.where(ok, dur, -1))[:25]
print("slowest blocks: dur us | second scans, beyond shell2, candidates, far, shell0, longest lane walk")
for i in idx:
    print("  %5.1f | %s" % (dur[i], " ".join("%6d" % int(v) for v in tr[i, 4:10])))
sel = np.nonzero(ok)[0]
for name, col in (("second", 4), ("beyond2", 5), ("cands", 6), ("far", 7), ("shell0", 8), ("lanewalk", 9)):
    c = np.corrcoef(dur[sel], tr[sel, col].astype(np.float64))[0, 1]
    print("corr(dur, %s) = %.2f   mean %.1f" % (name, c, tr[sel, col].mean()))
fast = sel[dur[sel] < np.percentile(dur[sel], 50)]
print("fast half means:", [round(float(tr[fast, c].mean()), 1) for c in range(4, 10)])

xcc = (tr[:, 3] >> 32).astype(np.int64)
print("per XCD: blocks, mean duration us, last end us")
for x in range(8):
    m = ok & (xcc == x)
    if m.any():
        print("  xcd %d: %4d blocks, mean %.1f, p90 %.1f, last end %.1f, candidates %d" % (x, int(m.sum()), dur[m].mean(), np.percentile(dur[m], 90), end[m].max(), int(tr[m, 6].sum())))

# which keypoint type a hardware block served (k_search_all: j = block / 8 < ceil(edge blocks / 8) -> edges)
kp = [slam.keypoints(k).size for k in range(3)]
se = ((kp[0] + 31) // 32 + 7) // 8
bidx = np.arange(tr.shape[0])
is_edge = (bidx // 8) < se
for name, m in (("edge", ok & is_edge), ("plane", ok & ~is_edge)):
    if m.any():
        print("%s blocks %d: duration mean %.1f p90 %.1f max %.1f us, last end %.1f" % (name, int(m.sum()), dur[m].mean(), np.percentile(dur[m], 90), dur[m].max(), end[m].max()))
        md = (end - mid)[m]
        print("   model fits behind the search: mean %.1f p50 %.1f p90 %.1f max %.1f us; search + fits max %.1f" % (md.mean(), np.median(md), np.percentile(md, 90), md.max(), (end - start)[m].max()))

# phases of the search inside a block (thread 0's wavefront): rows known, first scan, its merge, second scan | staged, barrier passed
ph = np.stack([(tr[:, 10] >> (16 * i)) & 0xffff for i in range(4)] + [(tr[:, 11] >> (16 * i)) & 0xffff for i in range(2)], 1).astype(np.float64) / 100.0
tot = (end - start)
names = ["rows", "scan1", "merge1", "scan2", "staged", "barrier"]
for label, m in (("all", ok), ("slowest 5 %", ok & (tot >= np.percentile(tot[ok], 95))), ("median band", ok & (tot >= np.percentile(tot[ok], 40)) & (tot <= np.percentile(tot[ok], 60)))):
    print("%-12s n=%4d  " % (label, int(m.sum())) + "  ".join("%s %.1f" % (n, ph[m, i].mean()) for i, n in enumerate(names)) + "  end %.1f" % tot[m].mean())
