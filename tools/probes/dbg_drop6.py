import sys, os, ctypes as C
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
for t in range(14):
    z=sc.get_state()[0].cpu().numpy()[:,2]
    sc.step_fused(None,*bufs); o.step_batch(None)
    d=sc.get_diag(); nc=d[0].cpu().numpy(); w=d[2].cpu().numpy()
    nco=np.array([o.counts(e)[0] for e in range(B)])
    print(t,'z',z.round(4),'gpu ncon',nc,'ncand',(w>>8)&255,'mycount lane0',(w>>16)&255,'oracle',nco)
    ap=bufs[0].cpu().numpy()
    bad=np.where(nc!=nco)[0]
    for e in bad: print('   env',e,'[overlap, dist, pa.z, pb.z, A.z, B.z, B.r] =',ap[e])
