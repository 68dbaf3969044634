import sys, os, time
R=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[R+'/gym-genesis_amd']
import torch
from gym_genesis.env import GenesisEnv
B=4096
for shape in ('capsule','box','capsule','box'):
    env=GenesisEnv(task='cube_pick',robot='franka',num_envs=B,link_shape=shape)
    env.reset(seed=0)
    dev=env._env.device
    acts=list(torch.empty((64,B,9),device=dev).uniform_(-1,1).unbind(0))
    for i in range(100): env.step(acts[i&63])
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(3000): env.step(acts[i&63])
    torch.cuda.synchronize(); api=(time.perf_counter()-t0)/3000*1e6
    t=env._env
    for i in range(100): t.step_raw(acts[i&63])
    torch.cuda.synchronize(); t0=time.perf_counter()
    for i in range(3000): t.step_raw(acts[i&63])
    torch.cuda.synchronize(); raw=(time.perf_counter()-t0)/3000*1e6
    print(f'{shape:8s} env.step {api:.2f} us  raw {raw:.2f} us  overhead {api-raw:.2f} us')
torch.cuda.synchronize(); t0=time.perf_counter()
for i in range(200): env.reset()
torch.cuda.synchronize(); print(f'env.reset() {(time.perf_counter()-t0)/200*1e6:.1f} us')
import numpy as np
t0=time.perf_counter()
for i in range(200): p=env._env.sample_spawn()
print(f'  sample_spawn {(time.perf_counter()-t0)/200*1e6:.1f} us')
t0=time.perf_counter()
for i in range(200): x=torch.from_numpy(p).to(dev)
torch.cuda.synchronize(); print(f'  from_numpy.to(device) {(time.perf_counter()-t0)/200*1e6:.1f} us')
t0=time.perf_counter()
for i in range(200): env._env.get_obs()
torch.cuda.synchronize(); print(f'  get_obs {(time.perf_counter()-t0)/200*1e6:.1f} us')
