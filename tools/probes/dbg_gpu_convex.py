import sys, ctypes as C
sys.path[:0]=['tests','oracle','gym-genesis_amd']
import numpy as np, torch
import test_convex_host as H
import test_gpu_convex as G
rows=H.random_pairs(4000,21)
got=G._device_pairs(rows)
host=np.zeros_like(got)
H.host_lib().convex_host_pairs(rows.ctypes.data_as(C.c_void_p), host.ctypes.data_as(C.c_void_p), rows.shape[0])
print('hits gpu',int(got[:,0].sum()),'host',int(host[:,0].sum()))
bad=np.where(got[:,0]!=host[:,0])[0]
print('hit mismatches',len(bad), bad[:20])
for i in bad[:6]:
    print(i, rows[i,[0,11]], 'gpu',got[i],'host',host[i])
both=(got[:,0]==1)&(host[:,0]==1)
d=np.abs(got[both,4]-host[both,4]); print('depth diff median',np.median(d),'max',d.max(), 'n>1e-5', int((d>1e-5).sum()))
