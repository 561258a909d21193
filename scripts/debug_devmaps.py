"""Device maps beside host maps, frame by frame: maps, sub-maps and poses (diagnostic)."""
import sys

import numpy as np

sys.path.insert(0, ".")
import lidarslam_amd as L

a = L.Slam(0, MapsOnDevice=1)
b = L.Slam(0, MapsOnDevice=0)
for f in range(4):
    pts, stamp = L.synth_frame(16, 1000, f)
    a.add_frame(pts, stamp, f)
    b.add_frame(pts, stamp, f)
    Ta, Tb = a.world_transform(), b.world_transform()
    print("frame", f, "pose diff", np.abs(Ta - Tb).max())
    for k in range(3):
        ma, mb = a.map(k), b.map(k)
        sa, sb = a.target_submap(k), b.target_submap(k)
        print("  type", k, "map", ma.size, mb.size, ma.tobytes() == mb.tobytes(), "submap", sa.size, sb.size, sa.tobytes() == sb.tobytes())
        if ma.size == mb.size and ma.tobytes() != mb.tobytes():
            d = np.nonzero([x.tobytes() != y.tobytes() for x, y in zip(ma, mb)])[0]
            print("    first differing", d[:5], ma[d[:2]], mb[d[:2]])
        if sa.size != sb.size or sa.tobytes() != sb.tobytes():
            sa_set = set(x.tobytes() for x in sa)
            sb_set = set(x.tobytes() for x in sb)
            print("    submap only-dev", len(sa_set - sb_set), "only-host", len(sb_set - sa_set))
