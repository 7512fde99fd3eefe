import sys, time, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
from s2p_amd import _lib as L
from oracle import pyoracle as po
from helpers import load_golden, golden_names, same
po.set_alias_oob(0)
for name in golden_names('sgbm_'):
    g=load_golden(name); dmin,dmax=int(g['params'][0]),int(g['params'][1])
    t=time.time(); r=L.sgbm(g['im1'],g['im2'],dmin,dmax,dump='full'); dt=time.time()-t
    o=po.oracle_sgbm(g['im1'],g['im2'],dmin,dmax,dump='full')
    res={k:same(o[k],r[k]) for k in ('q1','q2','C','S','disp_raw','cost_raw','disp_med','disp_fin','disp','cost')}
    print(name, r['geom'], '%.3fs'%dt, res, 'rminmax', r['rminmax'], o['rminmax'], flush=True)
    for k,v in res.items():
        if not v:
            a,b=o[k],r[k]; bad=np.argwhere(~((a==b)|((a!=a)&(b!=b))))
            print('   ',k,'mismatch count',len(bad),'first',bad[:5].tolist(), [ (a[tuple(i)],b[tuple(i)]) for i in bad[:5]])
