import sys, os, ctypes as C, glob
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0]=[R+'/tests',R+'/oracle',R+'/gym-genesis_amd']
import numpy as np, torch
import test_convex_host as H
rows=H.random_pairs(4000,21)
host=np.zeros((4000,8),np.float32)
H.host_lib().convex_host_pairs(rows.ctypes.data_as(C.c_void_p), host.ctypes.data_as(C.c_void_p), 4000)
tin=torch.as_tensor(rows,device='cuda')
for so in sorted(glob.glob(os.path.dirname(os.path.abspath(__file__))+'/probe_*.so')):
    L=C.CDLL(so); L.probe_run.argtypes=[C.c_void_p,C.c_void_p,C.c_int]
    tout=torch.zeros((4000,8),device='cuda')
    rc=L.probe_run(tin.data_ptr(),tout.data_ptr(),4000)
    got=tout.cpu().numpy()
    mm=got[:,0]!=host[:,0]
    both=(got[:,0]==1)&(host[:,0]==1)
    d=np.abs(got[both,4]-host[both,4])
    types=sorted(set(map(tuple,rows[mm][:,[0,11]].astype(int).tolist())))
    print(os.path.basename(so),'rc',rc,'hit mismatches',int(mm.sum()),types,'depth diff >1e-5:',int((d>1e-5).sum()),'max',d.max())
