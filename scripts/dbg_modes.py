import sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
from oracle import oracle as O
n=int(sys.argv[1])
dx=20.0/(n-1); x=np.arange(n)*dx
s=np.ascontiguousarray(np.broadcast_to((1.0/(1.0+0.1*x))[None,None,:],(n,n,n)))
src=cases.mt_sources(1); rcv=np.array([[0.,0,0]])
res={}
for mode in (0,1):
    g=ttcr_amd.Grid3d(x,x,x,cell_slowness=0,method='FSM',tt_from_rp=0,weno=0,dtype=np.float32)
    g.set_option('mode',mode); g.raytrace(src,rcv,slowness=s); res[mode]=g.get_grid_traveltimes().flatten('F'); dxg=g.dx
o=O.solve3d(np.float32,(n-1,)*3,dxg,(0,0,0),s.flatten('F'),src)
for mode in (0,1):
    d=res[mode]!=o['tt']; print('mode',mode,'mismatch vs oracle',int(d.sum()))
d=res[0]!=res[1]; print('mode0 vs mode1 mismatch',int(d.sum()))
