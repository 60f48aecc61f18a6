"""Repeats the B=8 backbone gradient comparison (tests/test_backbone_gpu.py) to look for run-to-run differences of the HIP path."""
import sys, os, hashlib
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "neuralnet-tracker-traincode_amd"))
import numpy as np, torch
import test_backbone_gpu as T
from oracle.synth import make_inputs
from trackertraincode.backbones.mobilenet_v1 import MobileNet

B = 8
for blur in (True, False):
    sd = T._backbone_state(blur=blur)
    image, _ = make_inputs(B, seed=7)
    G = np.random.default_rng(5).standard_normal((B, 1024)).astype(np.float32)
    f64, st64 = T._run_oracle(sd, image, G, torch.float64)
    f32, st32 = T._run_oracle(sd, image, G, torch.float32)
    ref = None
    for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
        net = MobileNet(num_classes=None, use_blurpool=blur).cuda(); T._load_into(net, sd); net.train()
        # disturb the allocator / leave other data behind between repetitions
        junk = [torch.randn(np.random.randint(1, 50) * 100000, device="cuda") for _ in range(np.random.randint(0, 6))]
        feat = net.forward_features(torch.from_numpy(image).cuda())
        (feat * torch.from_numpy(G).cuda()).sum().backward()
        torch.cuda.synchronize()
        worst = (0, None)
        h = hashlib.sha256(feat.detach().cpu().numpy().tobytes()).hexdigest()[:8]
        for k, p_ in net.named_parameters():
            g64 = st64["convnet." + k].grad
            e_hip, e_cpu = T._rel(p_.grad.cpu(), g64), T._rel(st32["convnet." + k].grad, g64)
            r = e_hip / (3 * e_cpu + 2e-5)
            if r > worst[0]:
                worst = (r, k, e_hip, e_cpu)
        gh = hashlib.sha256(net.conv1.weight.grad.cpu().numpy().tobytes()).hexdigest()[:8]
        print(f"blur={blur} rep={rep} feat={h} conv1grad={gh} worst={worst[0]:.2f} {worst[1]} e_hip={worst[2]:.2e} e_cpu={worst[3]:.2e}", flush=True)
        del junk
