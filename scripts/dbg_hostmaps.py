import os, sys, numpy as np
sys.path.insert(0,'/root/repo')
import lidarslam_amd as L
keep = L.Context(0)
frames = [L.synth_frame(8, 1000, f) for f in range(4)]
sg = L.Slam(0, EgoMotion=3, MapsOnDevice=0, ICPAhead=int(sys.argv[1]))
for f, (pts, stamp) in enumerate(frames):
    sg.add_frame(pts, stamp, f)
sg.close()
