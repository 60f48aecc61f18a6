"""B=3 gradient error of the HIP backbone vs the fp64 oracle over input seeds, with and without BlurPool blocks
(diagnostic for tests/test_backbone_gpu.py: is a miss the ReLU-flip noise of tiny batches or a real deviation?)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "neuralnet-tracker-traincode_amd"))
import numpy as np, torch
import test_backbone_gpu as T
from oracle.synth import make_inputs
from trackertraincode.backbones.mobilenet_v1 import MobileNet

for blur in (False, True):
    for seed in (7, 7, 7, 8, 9, 10):
        B = int(sys.argv[1]) if len(sys.argv) > 1 else 3
        sd = T._backbone_state(blur=blur)
        image, _ = make_inputs(B, seed=seed)
        G = np.random.default_rng(5).standard_normal((B, 1024)).astype(np.float32)
        f64, st64 = T._run_oracle(sd, image, G, torch.float64)
        f32, st32 = T._run_oracle(sd, image, G, torch.float32)
        net = MobileNet(num_classes=None, use_blurpool=blur).cuda(); T._load_into(net, sd); net.train()
        feat = net.forward_features(torch.from_numpy(image).cuda())
        (feat * torch.from_numpy(G).cuda()).sum().backward()
        worst = (0, None)
        for k, p_ in net.named_parameters():
            g64 = st64["convnet." + k].grad
            e_hip, e_cpu = T._rel(p_.grad.cpu(), g64), T._rel(st32["convnet." + k].grad, g64)
            r = e_hip / (3 * e_cpu + 2e-5)
            if r > worst[0]:
                worst = (r, k, e_hip, e_cpu)
        c1 = "conv1.weight"
        print(f"blur={blur} seed={seed} feat_hip={T._rel(feat.detach().cpu(), f64):.2e} feat_cpu={T._rel(f32, f64):.2e} worst={worst[0]:.2f} {worst[1]} e_hip={worst[2]:.2e} e_cpu={worst[3]:.2e}", flush=True)
