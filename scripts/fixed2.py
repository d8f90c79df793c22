import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, ttcr_amd, cases
n=512; ns=int(sys.argv[1]) if len(sys.argv)>1 else 64
dx=20.0/(n-1); x=np.arange(n)*dx
s=np.ascontiguousarray(np.broadcast_to((1.0/(1.0+0.1*x))[None,None,:],(n,n,n))).astype(np.float32)
g=ttcr_amd.Grid3d(x,x,x,n_threads=ns,cell_slowness=0,method='FSM',tt_from_rp=0,weno=0,dtype=np.float32)
g.set_option('fixed_iters',2); g.set_slowness(s)
src=cases.mt_sources(64)[:ns]; rcv=np.zeros((ns,3))
for r in range(2):
    g.raytrace(src,rcv); print('sweep_ms',g.timing()['sweep_ms'], flush=True)
