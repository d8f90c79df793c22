import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
n=int(sys.argv[1]) if len(sys.argv)>1 else 200
rng=np.random.default_rng(1)
s=rng.uniform(0.3,1.0,(n,n,n)).astype(np.float32)
x=np.arange(n)*0.1
src=np.array([[3.33,7.1,9.02]]); rcv=np.array([[0.,0,0]])
out={}
for mode in (0,1):
    g=ttcr_amd.Grid3d(x,x,x,cell_slowness=0,method='FSM',tt_from_rp=0,weno=0,dtype=np.float32)
    g.set_option('mode',mode)
    g.set_slowness(s)
    for rep in range(3):
        t=time.time(); g.raytrace(src,rcv); el=time.time()-t
    out[mode]=g.get_grid_traveltimes().copy()
    print('mode',mode,'niter',g.get_niter(),'time %.1f ms'%(el*1e3), g.timing())
print('bit-exact between modes:', np.array_equal(out[0],out[1]), 'maxdiff', float(np.max(np.abs(out[0]-out[1]))))
