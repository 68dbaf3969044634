import sys, os, ctypes as C
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[R+'/tests',R+'/oracle',R+'/gym-genesis_amd']
import numpy as np, torch
import orc
from gym_genesis.backend import spec as S
from gym_genesis.backend.lib import MirScene
def run(kind, poison):
    sb=S.SceneBuilder()
    sb.add_geom(0,S.GEOM_BOX,size=(0.15,0.15,0.05),pos=(0,0,0.05))
    sb.add_body('a',0,pos=(0,0,0.5),jtype=S.JNT_FREE,mass=0.3,inertia=S.sphere_inertia(0.3,0.04))
    if kind=='sphere': sb.add_geom('a',S.GEOM_SPHERE,size=(0.04,0,0))
    elif kind=='capsule': sb.add_geom('a',S.GEOM_CAPSULE,size=(0.03,0.05,0))
    else: sb.add_geom('a',S.GEOM_BOX,size=(0.04,0.04,0.04))
    sb.task=dict(eef_body=1,obj_body=1,grip_dof=(),reward_z=0.1)
    spec=sb.build()
    B=8
    rng=np.random.default_rng(5)
    pos=np.zeros((B,1,3),np.float32); pos[:,0]=rng.uniform(-0.05,0.05,(B,3))+[0.0,0.0,0.16]
    quat=np.tile(np.array([1,0,0,0],np.float32),(B,1,1))
    sc,o=MirScene(spec,B),orc.Oracle(spec,B)
    L=sc.lib; L.mir_debug_poison_lds.argtypes=[C.c_int,C.c_void_p]; L.mir_debug_poison_lds.restype=C.c_int
    sc.reset(pos,quat,np.zeros((B,0),np.float32)); o.reset(pos,quat,np.zeros((B,0),np.float32))
    bufs=(sc.empty(sc.agent_dim),sc.empty(sc.env_dim),sc.empty(),sc.empty(dtype=torch.uint8))
    bad=0
    for t in range(30):
        if poison: L.mir_debug_poison_lds(0,None); torch.cuda.synchronize()
        sc.step_fused(None,*bufs); o.step_batch(None)
        nc=sc.get_diag()[0].cpu().numpy(); nco=np.array([o.counts(e)[0] for e in range(B)])
        if (nc!=nco).any(): bad+=1
    q=sc.get_state()[0].cpu().numpy()
    print(kind,'poison',poison,'mismatching steps',bad,'final err',np.abs(q-o.state()[0]).max())
for kind in ('box','sphere','capsule'):
    for p in (False,True): run(kind,p)
