import sys, os, time, threading, numpy as np
sys.path.insert(0, os.getcwd())
import thesia_amd as ta
from tests.synth import synth_track
cmap = open("tests/golden/colormap_inferno_rgba258.bin", "rb").read()
with ta.Context(0) as ctx:
    tm = ta.TrackManager(ctx); tm.set_setting(2048/48, 4, 1, ta.LINEAR); tm.set_colormap(cmap)
    x = synth_track(1, 48000, 48000*30)
    tm.add_tracks([(i, 48000, x[None]) for i in range(4)]); tm.apply_track_list_changes()
    def run(n, out):
        ts = []
        for i in range(n):
            t0 = time.perf_counter(); tm.get_spectrogram_tile(i % 4, 0, 0, 0, i % 5, 0); ts.append(time.perf_counter() - t0)
        out.extend(ts)
    for nt in (1, 8):
        run(50, [])
        outs = [[] for _ in range(nt)]
        th = [threading.Thread(target=run, args=(300, outs[i])) for i in range(nt)]
        [t.start() for t in th]; [t.join() for t in th]
        a = np.sort(np.concatenate(outs)) * 1e6
        print(f"threads {nt}: p50 {a[len(a)//2]:.1f} us  p99 {a[int(len(a)*0.99)]:.1f} us")
    tm.close()
