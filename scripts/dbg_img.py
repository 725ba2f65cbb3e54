import numpy as np, sys
sys.path.insert(0,'.')
import thesia_amd as ta
from oracle import oracle as orc
ctx=ta.Context(0)
for (T,H,i0,i1,cm) in [(300,1025,0,1025,258),(129,513,0,1026,4),(64,1025,0,1025,258),(64,128,0,128,258),(64,129,0,129,258)]:
    rng=np.random.default_rng(T*7+H)
    spec=rng.uniform(-140,10,(T,H)).astype(np.float32)
    got=ctx.spec_to_img(spec,(i0,i1),(-100.0,0.0),cm)
    want=orc.convert_spectrogram_to_img(spec,(i0,i1),(-100.0,0.0),cm)
    bad=np.argwhere(got!=want)
    print(T,H,len(bad), bad[:8].tolist(), [ (int(got[tuple(b)]),int(want[tuple(b)]), float(spec[b[1], b[0]+i0]) if b[0]+i0<H else None) for b in bad[:8]])
