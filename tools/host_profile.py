"""Where does the host time of one step go? (GPU box)"""
import cProfile, pstats, io, os, sys, time
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "neuralnet-tracker-traincode_amd"))
import torch
import bench
args = bench.parse(); args.batch = int(os.environ.get("B", 512))
dev = torch.device("cuda", 0)
net, crit, opt, batches, train = bench.build_step(args, dev)
params = list(net.parameters())
def step():
    for p in params: p.grad = None
    out = train.training_step(net, batches, 0, crit)
    out["loss"].backward()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"host enqueue {1e2*(t1-t0):.2f} ms/step, total {1e2*(t2-t0):.2f} ms/step")
x = batches[0]["image"]
def bb():
    f = net.convnet.forward_features(x); f.backward(torch.ones_like(f))
for _ in range(2): bb()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): bb()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"backbone-only host enqueue {1e2*(t1-t0):.2f} ms/step (batch {x.shape[0]})")
pr = cProfile.Profile(); pr.enable()
for _ in range(5): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28); print(s.getvalue()[:6000])
