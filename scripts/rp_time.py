import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
n=int(sys.argv[1])
dx=20.0/(n-1); x=np.arange(n)*dx
s=np.ascontiguousarray(np.broadcast_to((1.0/(1.0+0.1*x))[None,None,:],(n,n,n)))
src=cases.mt_sources(1); rcv=cases.rcv_lattice3d()
for ttrp in (0,1):
    g=ttcr_amd.Grid3d(x,x,x,cell_slowness=0,method='FSM',tt_from_rp=ttrp,weno=0,dtype=np.float32)
    g.set_slowness(s)
    for rep in range(3):
        t=time.perf_counter(); g.raytrace(src,rcv); el=time.perf_counter()-t
    print('n',n,'tt_from_rp',ttrp,'wall %.2f ms'%(el*1e3),'sweeps %.2f ms'%g.timing()['sweep_ms'])
t=time.perf_counter(); tt,rays=g.raytrace(src,rcv,return_rays=True); print('return_rays wall %.2f ms'%((time.perf_counter()-t)*1e3), sum(len(r) for r in rays),'points')
