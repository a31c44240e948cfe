"""what a 20-step bracket reads after different preludes (scratch tool): 20 steps = 5 replays of the 4-step graph, bracketed by synchronize"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import bench
w = bench.NetflixShaped("nf", 0, torch.device("cuda:0"))
w.step(); w.step()
torch.cuda.synchronize()
f = w.fused
def bracket(n_steps):
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); w.run_steps(n_steps); e.record(); torch.cuda.synchronize()
    return round(s.elapsed_time(e) / n_steps, 4)
print("first 20 after capture:", bracket(20), "next:", bracket(20), bracket(20))
time.sleep(0.5)
print("20 after 0.5 s idle:", bracket(20), bracket(20))
w.eval_once(); torch.cuda.synchronize()
for _ in range(5): w.eval_once()
w.run_steps(5); torch.cuda.synchronize()
print("20 after 5 evals + 5 steps:", bracket(20), bracket(20))
print("200:", bracket(200), "then 20:", bracket(20), bracket(20), bracket(20))
time.sleep(10.0)
w.run_steps(5)
print("20 after 10 s idle + 5 steps:", bracket(20), bracket(20), "200:", bracket(200))
