import sys; sys.path.insert(0,'.')
import numpy as np
from ctucopy_amd import Engine
from oracle.oracle import Oracle
from tests.util import C4_NOVAD, sig
x=sig("CS0")
for cfg in (C4_NOVAD, "-fs 8000 -format_in raw -format_out htk -preset mfcc".split()):
    g=Engine(cfg).extract([x])[0]; r=Oracle(cfg).process(x)
    e=np.abs(g-r)/np.maximum(np.abs(r),1)
    idx=np.unravel_index(np.argmax(e),e.shape)
    print(cfg[-2:], e.max(), idx, g[idx], r[idx], "frames>5e-5:", (e.max(axis=1)>5e-5).sum())
    t=idx[0]; print(" row gpu", g[t]); print(" row ref", r[t]); print(" pcm", x[t*80:t*80+12], np.abs(x[t*80:t*80+200]).max())
