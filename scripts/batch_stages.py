"""Diagnostic: stage times of sequence 0 while S sequences run side by side (host maps)."""
import sys, threading, time
import numpy as np
sys.path.insert(0, ".")
import lidarslam_amd as L
L.bind_host_to_device(0)
S, frames, warm = 8, 40, 8
names = ["total", "extract", "ego_icp", "ego_lm", "loc_icp", "loc_lm", "undistort", "submap", "maps", "ego_it", "loc_it", "lm_evals", "matches", "kf", "maps_wait", "maps_async"]
slams, stamps = [], []
for s in range(S):
    sl = L.Slam(0, EgoMotion=3, MapsOnDevice=0)
    st = []
    for f in range(frames):
        pts, stamp = L.synth_frame(128, 1000 + s, f)
        sl.store_frame(f, pts)
        st.append(stamp)
    slams.append(sl); stamps.append(st)
acc = np.zeros((S, 16))
gate = threading.Barrier(S + 1)
def worker(s):
    for f in range(frames):
        if f == warm: gate.wait()
        if f + 1 < frames: slams[s].hint_next_stored_frame(f + 1)
        slams[s].add_stored_frame(f, stamps[s][f], f)
        if f >= warm: acc[s] += slams[s].stats()
    slams[s].context().sync()
    gate.wait()
ts = [threading.Thread(target=worker, args=(s,)) for s in range(S)]
for t in ts: t.start()
gate.wait(); t0 = time.perf_counter(); gate.wait(); dt = time.perf_counter() - t0
for t in ts: t.join()
print("fps", round(S * (frames - warm) / dt, 1))
m = acc.mean(axis=0) / (frames - warm)
print({n: round(1e3 * m[i], 3) if i < 9 or i >= 14 else round(m[i], 2) for i, n in enumerate(names)})
for s in slams: s.close()
