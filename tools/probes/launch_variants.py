import sys, os
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[R+'/gym-genesis_amd']
from gym_genesis.backend import models
from gym_genesis.backend.lib import MirScene
sc=MirScene(models.franka_cube_pick_scene().build(),64)
for name,it in (('hipLaunchKernelGGL',2000),('hipExtLaunchKernelGGL any-order',-1),('graph relaunch',-2),('hipLaunchKernelGGL',2000)):
    print(f'{name:34s} {sc.null_roundtrip_us(it):.2f} us per launch + host-visible completion')
