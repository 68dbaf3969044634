import sys, os
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[R+'/tests',R+'/oracle',R+'/gym-genesis_amd']
import numpy as np, torch
import orc
from gym_genesis.backend import spec as S
from gym_genesis.backend.lib import MirScene
sb=S.SceneBuilder()
sb.add_geom(0,S.GEOM_BOX,size=(0.15,0.15,0.05),pos=(0,0,0.05))
sb.add_body('a',0,pos=(0,0,0.5),jtype=S.JNT_FREE,mass=0.3,inertia=S.sphere_inertia(0.3,0.04))
sb.add_geom('a',S.GEOM_SPHERE,size=(0.04,0,0))
sb.task=dict(eef_body=1,obj_body=1,grip_dof=(),reward_z=0.1)
spec=sb.build()
B=8
rng=np.random.default_rng(5)
pos=np.zeros((B,1,3),np.float32); pos[:,0]=rng.uniform(-0.05,0.05,(B,3))+[0.0,0.0,0.16]
quat=np.tile(np.array([1,0,0,0],np.float32),(B,1,1))
sc,o=MirScene(spec,B),orc.Oracle(spec,B)
sc.reset(pos,quat,np.zeros((B,0),np.float32)); o.reset(pos,quat,np.zeros((B,0),np.float32))
bufs=(sc.empty(sc.agent_dim),sc.empty(sc.env_dim),sc.empty(),sc.empty(dtype=torch.uint8))
for t in range(30):
    st=[x.cpu().numpy().copy() for x in sc.get_state()]
    qo_prev=o.state()[0].copy()
    sc.step_fused(None,*bufs); o.step_batch(None)
    nc=sc.get_diag()[0].cpu().numpy(); nco=np.array([o.counts(e)[0] for e in range(B)])
    if (nc!=nco).any():
        print('step',t,'gpu ncon',nc,'oracle ncon',nco)
        print('z before step gpu',st[0][:,2].round(5),'oracle',qo_prev[:,2].round(5))
        sc2=MirScene(spec,B); sc2.set_state(qpos=st[0],qvel=st[1],warmstart=st[3])
        sc2.step_fused(None,*bufs); print('fresh scene, same state, step_fused ncon',sc2.get_diag()[0].cpu().numpy())
        o2=orc.Oracle(spec,B)
        for e in range(B): o2.write(orc.F_QPOS,st[0][e],e); o2.write(orc.F_QVEL,st[1][e],e)
        o2.step_batch(None); print('oracle from the GPU state ncon',[o2.counts(e)[0] for e in range(B)])
        break

# same rollout with the LDS of every CU poisoned (NaN patterns) before each step: an LDS slot read before it is written shows up
import ctypes as C
L=sc.lib
L.mir_debug_poison_lds.argtypes=[C.c_int,C.c_void_p]; L.mir_debug_poison_lds.restype=C.c_int
sc,o=MirScene(spec,B),orc.Oracle(spec,B)
sc.reset(pos,quat,np.zeros((B,0),np.float32)); o.reset(pos,quat,np.zeros((B,0),np.float32))
bad=0
for t in range(30):
    L.mir_debug_poison_lds(0, None); torch.cuda.synchronize()
    sc.step_fused(None,*bufs); o.step_batch(None)
    nc=sc.get_diag()[0].cpu().numpy(); nco=np.array([o.counts(e)[0] for e in range(B)])
    q=sc.get_state()[0].cpu().numpy()
    if (nc!=nco).any() or not np.isfinite(q).all():
        bad+=1
        if bad<4: print('poisoned: step',t,'gpu',nc,'oracle',nco,'finite',np.isfinite(q).all())
print('poisoned run: mismatching steps',bad)
