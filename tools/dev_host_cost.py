import os, sys, time, torch
ROOT=os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, ROOT)
from smarts_amd.sumo_map import load_net
from smarts_amd.map_compiler import compile_map
from smarts_amd.engine import BatchedSim, SimConfig, make_spawns
from smarts_amd import sharding
cm=compile_map(load_net(os.path.join(ROOT,'smarts_amd/scenarios/loop')))
E,N=1024,8
cfg=SimConfig(num_envs=E,num_vehicles=N,neighbors=True,nb_radius=50.0,auto_reset=True)
sim=BatchedSim(cm,cfg,spawns=make_spawns(cm,E,N,episodes=4,seed=42))
acts=torch.zeros((E,N),dtype=torch.int8,device='cuda')
g=sharding.RewardDoneGather(E,N,'cuda',1)
sim.reset()
for _ in range(50): sim.step(acts)
torch.cuda.synchronize()
# host-only cost: enqueue a small number of steps while the GPU is idle at start
K=20
t0=time.perf_counter()
for _ in range(K):
    g.release(sim.next_learner_block); o=sim.step(acts); g.start_packed(o["learner"])
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("enqueue-only per step (us):", (t1-t0)/K*1e6, " incl. drain:", (t2-t0)/K*1e6)
K=2000
t0=time.perf_counter()
for _ in range(K):
    g.release(sim.next_learner_block); o=sim.step(acts); g.start_packed(o["learner"])
t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
print("steady per step (us): host loop", (t1-t0)/K*1e6, " total", (t2-t0)/K*1e6)
