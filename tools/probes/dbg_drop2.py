import sys, os
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[R+'/tests',R+'/oracle',R+'/gym-genesis_amd']
import numpy as np, torch
import orc
from gym_genesis.backend import spec as S
from gym_genesis.backend.lib import MirScene
def scene(with_plane, with_caps, static_first=True):
    sb=S.SceneBuilder()
    if with_plane: sb.add_geom(0,S.GEOM_PLANE)
    sb.add_geom(0,S.GEOM_BOX,size=(0.15,0.15,0.05),pos=(0,0,0.05))
    sb.add_body('a',0,pos=(0,0,0.5),jtype=S.JNT_FREE,mass=0.3,inertia=S.sphere_inertia(0.3,0.04))
    sb.add_geom('a',S.GEOM_SPHERE,size=(0.04,0,0))
    if with_caps:
        sb.add_body('b',0,pos=(0.5,0,0.5),jtype=S.JNT_FREE,mass=0.3,inertia=S.capsule_inertia(0.3,0.03,0.07))
        sb.add_geom('b',S.GEOM_CAPSULE,size=(0.03,0.07,0))
    sb.task=dict(eef_body=1,obj_body=1,grip_dof=(),reward_z=0.1)
    return sb.build()
for wp,wc in ((False,False),(True,False),(True,True)):
    spec=scene(wp,wc)
    B=8
    rng=np.random.default_rng(5)
    nf=2 if wc else 1
    pos=np.zeros((B,nf,3),np.float32)
    pos[:,0]=rng.uniform(-0.05,0.05,(B,3))+[0.0,0.0,0.16]
    if wc: pos[:,1]=rng.uniform(-0.05,0.05,(B,3))+[0.5,0.0,0.3]
    quat=np.tile(np.array([1,0,0,0],np.float32),(B,nf,1))
    sc,o=MirScene(spec,B),orc.Oracle(spec,B)
    arm=np.zeros((B,0),np.float32)
    sc.reset(pos,quat,arm); o.reset(pos,quat,arm)
    bufs=(sc.empty(sc.agent_dim),sc.empty(sc.env_dim),sc.empty(),sc.empty(dtype=torch.uint8))
    first=None
    for t in range(30):
        sc.step_fused(None,*bufs); o.step_batch(None)
        nc=sc.get_diag()[0].cpu().numpy(); nco=np.array([o.counts(e)[0] for e in range(B)])
        if (nc!=nco).any() and first is None: first=(t,nc.copy(),nco.copy())
    print('plane',wp,'capsule',wc,'npair',sc.npair,'first ncon mismatch',first)

# exact state replay of the failing step through the pair-by-pair debug kernel
import test_gpu_convex as G
spec=scene(False,False)
B=8
rng=np.random.default_rng(5)
pos=np.zeros((B,1,3),np.float32); pos[:,0]=rng.uniform(-0.05,0.05,(B,3))+[0.0,0.0,0.16]
quat=np.tile(np.array([1,0,0,0],np.float32),(B,1,1))
sc,o=MirScene(spec,B),orc.Oracle(spec,B)
sc.reset(pos,quat,np.zeros((B,0),np.float32)); o.reset(pos,quat,np.zeros((B,0),np.float32))
bufs=(sc.empty(sc.agent_dim),sc.empty(sc.env_dim),sc.empty(),sc.empty(dtype=torch.uint8))
for t in range(30):
    qprev=sc.get_state()[0].cpu().numpy().copy(); vprev=sc.get_state()[1].cpu().numpy().copy()
    sc.step_fused(None,*bufs); o.step_batch(None)
    nc=sc.get_diag()[0].cpu().numpy(); nco=np.array([o.counts(e)[0] for e in range(B)])
    if (nc!=nco).any():
        e=int(np.where(nc!=nco)[0][0])
        print('step',t,'env',e,'state before step q',qprev[e],'v',vprev[e])
        rows=np.zeros((1,22),np.float32)
        rows[0,0]=1; rows[0,1:4]=[0.15,0.15,0.05]; rows[0,4:7]=[0,0,0.05]; rows[0,7]=1
        rows[0,11]=2; rows[0,12]=0.04; rows[0,15:18]=qprev[e,:3]; rows[0,18:22]=qprev[e,3:7]
        print('debug kernel on that state:', G._device_pairs(rows))
        # forward at that state through mir_forward (mode 1: collision of the CURRENT state)
        sc2=MirScene(spec,B)
        sc2.set_state(qpos=np.tile(qprev[e],(B,1)),qvel=np.tile(vprev[e],(B,1)))
        sc2.forward(); print('mir_forward ncon at that state (all envs same):', sc2.get_diag()[0].cpu().numpy())
        break
