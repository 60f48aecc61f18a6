"""cProfile of the eager step's enqueueing thread (and the autograd thread's Python frames): python tools/exp/host_profile.py [bench flags]"""
import cProfile, os, pstats, sys
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import torch
import bench
sys.argv = ["bench.py"] + sys.argv[1:]
args = bench.parse()
dev = torch.device("cuda", 0)
net, crit, opt, batches, train = bench.build_step(args, dev)
params = list(net.parameters())
def step():
    for p in params: p.grad = None
    out = train.training_step(net, batches, 0, crit)
    out["loss"].backward()
    opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(60)
