import sys, os
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[R+'/oracle',R+'/gym-genesis_amd']
import numpy as np, torch, time
from gym_genesis.backend import models
from gym_genesis.backend.lib import MirScene
B=4096
def run(iters):
    sb=models.franka_cube_pick_scene(); sb.opt["iterations"]=iters
    sc=MirScene(sb.build(),B)
    rng=np.random.RandomState(0)
    pos=np.stack([rng.uniform(.45,.8,B),rng.uniform(-.25,.25,B),np.full(B,.02)],1).astype(np.float32)
    sc.reset(pos,np.tile(np.array([0,0,0,1],np.float32),(B,1)),np.tile(np.array(models.FRANKA_HOME,np.float32),(B,1)))
    g=torch.Generator(device=sc.device).manual_seed(1234)
    acts=torch.empty((256,B,9),device=sc.device).uniform_(-1,1,generator=g)
    bufs=(sc.empty(9),sc.empty(11),sc.empty(),sc.empty(dtype=torch.uint8))
    hist=np.zeros(8); blockmax=np.zeros(8)
    for t in range(100):
        sc.step_fused(acts[t],*bufs)
        if t>=20:
            ni=sc.get_diag()[2].cpu().numpy()
            hist+=np.bincount(np.minimum(ni,7),minlength=8)
            blockmax+=np.bincount(np.minimum(ni.reshape(-1,4).max(1),7),minlength=8)
    sc.set_diag(False)
    torch.cuda.synchronize(); t0=time.perf_counter()
    for t in range(500): sc.step_fused(acts[t%256],*bufs)
    torch.cuda.synchronize(); us=(time.perf_counter()-t0)/500*1e6
    return hist/hist.sum(), blockmax/blockmax.sum(), us, sc.get_state()[0].cpu().numpy()
h50,b50,us50,q50=run(50)
print('iterations=50: niter hist',h50.round(4),'block-max hist',b50.round(4),'us/step',round(us50,2))
for it in (1,2,3):
    h,b,us,q=run(it)
    print(f'iterations={it}: us/step {us:.2f}; state diff vs 50 after 600 steps: median {np.median(np.abs(q-q50).max(1)):.2e}')
