"""dev: extract_vectors on a list of images with MANY distinct sizes (real collections: longer side 1024, arbitrary aspect), multi-scale + whitening network"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from bench_configs import _c3_network
from gandtr_amd.stages.validate import extract_vectors
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
with tempfile.TemporaryDirectory() as tmp:
    net = _c3_network(dev, True, tmp)
    rng = np.random.RandomState(0)
    shorts = [683, 768, 682, 576, 1024, 685, 680, 765, 700, 640, 819, 724]
    sizes = []
    for i in range(96):
        s = shorts[rng.randint(len(shorts))]
        sizes.append((1024, s) if rng.rand() < 0.7 else (s, 1024))
    imgs = [synth.synth_input(500 + i, (3,) + sz).to(dev) for i, sz in enumerate(sizes)]
    print("distinct sizes:", len(set(sizes)))
    for label, kw in (("batch-1 loop", {"batched": False}), ("equal sizes batched, one group at a time", {"concurrent": 1}), ("equal sizes batched, small groups together (8 images in flight)", {})):
        extract_vectors(net, imgs, dev, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        v = extract_vectors(net, imgs, dev, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(label, "%.1f desc/s (%.1f ms per image)" % (len(imgs) / dt, dt / len(imgs) * 1e3))
    # every size different: nothing to batch
    uniq = [synth.synth_input(900 + i, (3, 1024, 600 + 8 * i)).to(dev) for i in range(48)]
    for label, kw in (("all sizes distinct: batch-1 loop", {"batched": False}), ("all sizes distinct: two forwards in flight", {"concurrent": 2}), ("all sizes distinct: four forwards in flight", {"concurrent": 4}),
                      ("all sizes distinct: eight forwards in flight (default)", {})):
        extract_vectors(net, uniq, dev, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        v = extract_vectors(net, uniq, dev, **kw)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print(label, "%.1f desc/s (%.1f ms per image)" % (len(uniq) / dt, dt / len(uniq) * 1e3))
