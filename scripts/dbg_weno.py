import sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, cases
from gpu_util import run_case
from oracle import oracle as O
name=sys.argv[1]; dt=np.float64 if sys.argv[2]=='f64' else np.float32
c=[c for c in cases.cases3d()+cases.cases2d() if c['name']==name][0]
r=run_case(c,dt,weno=1)
if c['dim']==3:
    o=O.solve3d(dt,c['ncells'],r['grid'].dx,c['origin'],c['slowness'],c['src'],c['t0'],cell_slowness=c['cell_slowness'],translate=c['translate'],weno=True)
else:
    o=O.solve2d(dt,c['ncells'],r['grid'].dx,r['grid'].dz,c['origin'],c['slowness'],c['src'],c['t0'],cell_slowness=c['cell_slowness'],weno=True)
print('niter',r['niter'],o['niter'],'niterw',r['niterw'],o['niterw'])
d=np.abs(r['tt']-o['tt']); bad=np.nonzero(d>0)[0]
print('nbad',bad.size,'max',d.max())
nn=[v+1 for v in c['ncells']]
if c['dim']==3:
    for n in bad[:10]:
        i=n%nn[0]; j=(n//nn[0])%nn[1]; k=n//(nn[0]*nn[1]); print(i,j,k,r['tt'][n],o['tt'][n])
    if bad.size:
        ii=bad%nn[0]; jj=(bad//nn[0])%nn[1]; kk=bad//(nn[0]*nn[1])
        print('i range',ii.min(),ii.max(),'j',jj.min(),jj.max(),'k',kk.min(),kk.max())
