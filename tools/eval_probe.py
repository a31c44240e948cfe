"""Where do the 80 ms of Trainer.test() in graph mode go (tools/e2e_main.py: eval 0.082 s with the evaluation graph, 0.0014 s without)?
Times the pieces of main.Trainer.test() after two epochs of training on the tools/e2e_main.py dataset."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import e2e_main
data = "/tmp/llmrec_e2e"
e2e_main.write_dataset(data)
sys.argv = ["main.py", "--dataset", "netflix_valid_item", "--data_path", data + "/", "--epoch", "2", "--debug"]
import main as M
import torch
M._progress = lambda it: it
M.set_seed(2022)
tr = M.Trainer(data_config={})
tr.train()
sync = torch.cuda.synchronize
users = list(M.data_generator.test_set.keys())
fused = tr._fused_step()
def T(fn, n=5):
    sync(); t = time.perf_counter()
    for _ in range(n): r = fn()
    sync(); return (time.perf_counter() - t) / n * 1e3
print("whole test() ms", T(lambda: tr.test(users, False)))
print("tuple build ms", T(lambda: tuple(int(u) for u in users)))
import numpy as np
q = tr._eval_queries[np.asarray(users, dtype=np.int64).tobytes()]
st = M.data_generator.device_state(M.device)
print("eval_topk graph replay ms", T(lambda: fused.eval_topk(q, st["train"], 50, use_graph=True)))
idx, _ = fused.eval_topk(q, st["train"], 50, use_graph=True)
print("test_torch(topk given) ms", T(lambda: M.test_torch(fused.E_u, fused.E_i, users, False, topk=(q, idx))))
print("eval_topk eager ms", T(lambda: fused.eval_topk(q, st["train"], 50, use_graph=False)))
print("test_torch(no topk) ms", T(lambda: M.test_torch(fused.E_u, fused.E_i, users, False)))
# alternate a training step and an evaluation
b = tr.sample_batch()
def alt():
    tr.train_step(*b, clone=False)
    return fused.eval_topk(q, st["train"], 50, use_graph=True)
print("train step + eval replay ms", T(alt))
print("train step alone ms", T(lambda: tr.train_step(*b, clone=False)))
print("sample_batch ms", T(lambda: tr.sample_batch(), 20))
print("Data.sample ms", T(lambda: M.data_generator.sample(), 20))

import cProfile, pstats
pr = cProfile.Profile(); sync(); pr.enable()
for _ in range(3): tr.test(users, False)
sync(); pr.disable()
pstats.Stats(pr).sort_stats("cumtime").print_stats(18)

# round 6: the evaluation as a graph / eagerly, with the forward's side streams or on one stream
print("--- eval_topk variants (ms per evaluation, 20 back to back) ---")
for ms_flag in (True, False):
    fused.multi_stream = ms_flag
    fused._eval_graphs.clear()
    for g in (True, False):
        fused.eval_topk(q, st["train"], 50, use_graph=g)
        print("multi_stream=%s graph=%s: %.4f" % (ms_flag, g, T(lambda: fused.eval_topk(q, st["train"], 50, use_graph=g), 20)))
fused.multi_stream = True
fused._eval_graphs.clear()
print("--- evaluation graph with two branches (ID chain beside the projection, the profile chain on the main stream) ---")
fused.profile_on_main = True
for g in (True, False):
    fused._eval_graphs.clear()
    fused.eval_topk(q, st["train"], 50, use_graph=g)
    print("profile_on_main graph=%s: %.4f" % (g, T(lambda: fused.eval_topk(q, st["train"], 50, use_graph=g), 20)))
    print("   isolated (sync per call): %.4f" % T(lambda: (fused.eval_topk(q, st["train"], 50, use_graph=g), sync()), 20))
fused.profile_on_main = False
for ms_flag, g in ((True, True), (False, True), (True, False)):
    fused.multi_stream = ms_flag
    fused._eval_graphs.clear()
    fused.eval_topk(q, st["train"], 50, use_graph=g)
    print("multi_stream=%s graph=%s isolated (sync per call): %.4f" % (ms_flag, g, T(lambda: (fused.eval_topk(q, st["train"], 50, use_graph=g), sync()), 20)))
fused.multi_stream = True
fused._eval_graphs.clear()
