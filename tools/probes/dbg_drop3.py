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
q=np.zeros((B,7),np.float32); q[:,3]=1
rng=np.random.default_rng(1)
q[:,0:2]=rng.uniform(-0.05,0.05,(B,2)); q[:,2]=np.linspace(0.128,0.1395,B)
v=np.zeros((B,6),np.float32); v[:,2]=rng.uniform(-0.5,0.5,B)
o=orc.Oracle(spec,B)
for e in range(B):
    o.write(orc.F_QPOS,q[e],e); o.write(orc.F_QVEL,v[e],e)
o.step_batch(None)
nco=np.array([o.counts(e)[0] for e in range(B)])
res={}
for name in ('fused','step','forward','packed'):
    sc=MirScene(spec,B)
    sc.set_state(qpos=q,qvel=v,warmstart=np.zeros((B,6),np.float32))
    if name=='fused':
        sc.step_fused(None,sc.empty(sc.agent_dim),sc.empty(sc.env_dim),sc.empty(),sc.empty(dtype=torch.uint8))
    elif name=='step': sc.step(1)
    elif name=='packed': sc.step_packed(None, torch.zeros((B,24),device=sc.device))
    else: sc.forward()
    res[name]=sc.get_diag()[0].cpu().numpy()
print('oracle',nco)
for k,vv in res.items(): print(k,vv)
