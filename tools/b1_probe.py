"""dev: the reference's own operating point -- one image per call through the multi-scale + whitening ResNet-101 (extract_vectors' batch-1 loop,
imageretrievalnet.py:319-333): where a call's time goes"""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bench_configs import _c3_network
from gandtr_amd.stages.validate import extract_vectors
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")


def loop(fn, k):
    fn(); fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    return (t1 - t0) / k * 1e3, (time.perf_counter() - t0) / k * 1e3


with torch.no_grad(), tempfile.TemporaryDirectory() as tmp:
    hub = _c3_network(dev, True, tmp)
    x = synth.synth_input(5, (1, 3, 1024, 1024)).to(dev)
    h, t = loop(lambda: hub(x), 32)
    print("hub(x) one 1024^2 image per call, 32 calls back to back: host %.2f ms, total %.2f ms per call" % (h, t))
    sizes = [(768, 1024), (1024, 768), (1024, 1024)]
    imgs = [synth.synth_input(300 + i, (3,) + sizes[i % 3]).to(dev) for i in range(48)]
    same = [synth.synth_input(300 + i, (3, 1024, 1024)).to(dev) for i in range(48)]
    for name, lst in (("3 sizes alternating", imgs), ("one size", same)):
        h, t = loop(lambda: extract_vectors(hub, lst, dev, batched=False), 2)
        print("extract_vectors batch-1 loop, 48 images, %s: host %.2f ms, total %.2f ms per image" % (name, h / 48, t / 48))
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    extract_vectors(hub, imgs, dev, batched=False)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
