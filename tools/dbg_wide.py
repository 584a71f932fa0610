import os, sys
os.environ["XPNG_WIDE_RANS"]="1"
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT","/root/repo"))
import numpy as np, tempfile
import xpng_amd
from oracle import pyoracle as po
from xpng_amd.synth import synth_raster
r = synth_raster("photo", 600, 520, True, seed=5)
want = po.encode_image(1, r)
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "a.xpng"); open(p,"wb").write(want)
    back = xpng_amd.load(p)
print("equal:", np.array_equal(back, r), "diff px:", int((back != r).any(axis=2).sum()))
d = np.argwhere((back != r).any(axis=2))
print(d[:10])
for y,x in d[:5]: print(back[y,x], r[y,x])
