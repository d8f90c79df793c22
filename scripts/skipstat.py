import sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases, time
n=int(sys.argv[1]); ns=int(sys.argv[2]) if len(sys.argv)>2 else 1
dx=20.0/(n-1); x=np.arange(n)*dx
g=ttcr_amd.Grid3d(x,x,x,n_threads=ns,cell_slowness=0,method='FSM',tt_from_rp=0,weno=0,dtype=np.float32)
z=np.arange(n)*dx; s=np.broadcast_to((1.0/(1.0+0.1*z))[None,None,:],(n,n,n))
g.set_slowness(np.ascontiguousarray(s))
src=cases.mt_sources(64)[:ns]; rcv=np.tile(np.array([[0.,0,0]]),(ns,1))
for skip in (0,1):
    g.set_option('skip',skip)
    for maxit in (1,2):
        g.set_option('fixed_iters',maxit)
        g.raytrace(src,rcv); t=g.timing()
        print('skip',skip,'iters',maxit,'evaluated %.3f'%(t['evaluated_updates']/t['node_updates']),'sweep_ms %.2f'%t['sweep_ms'])
