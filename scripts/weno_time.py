import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
n=int(sys.argv[1])
dt=np.float64 if len(sys.argv)>2 and sys.argv[2]=='f64' else np.float32
dx=20.0/(n-1); x=np.arange(n)*dx
z=np.arange(n)*dx; s=np.ascontiguousarray(np.broadcast_to((1.0/(1.0+0.1*z))[None,None,:],(n,n,n)))
src=cases.mt_sources(1); rcv=cases.rcv_lattice3d()
for weno in (0,1):
    g=ttcr_amd.Grid3d(x,x,x,cell_slowness=0,method='FSM',tt_from_rp=0,weno=weno,dtype=dt)
    g.set_slowness(s)
    for rep in range(2):
        t=time.time(); g.raytrace(src,rcv); el=time.time()-t
    tm=g.timing()
    print('n',n,dt.__name__,'weno',weno,'niter',g.get_niter(),g.get_niterw(),'time %.1f ms'%(el*1e3),'sweep %.1f ms'%tm['sweep_ms'], 'Mnodes/s/iter %.0f'%(n**3*(g.get_niter()+g.get_niterw())/tm['sweep_ms']/1e3))
