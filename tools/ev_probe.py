"""dev: bench_configs.py's batch-1 extract_vectors row reads 205 desc/s after its c3 section and 360 without it: profile the loop in that state"""
import argparse, cProfile, os, pstats, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench_configs as B
from gandtr_amd.stages.validate import extract_vectors
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
B.main(argparse.Namespace(only="c3", no_cpu_baseline=True))
with torch.no_grad(), tempfile.TemporaryDirectory() as tmp:
    sizes = [(768, 1024), (1024, 768), (1024, 1024)]
    imgs = [synth.synth_input(300 + i, (3,) + sizes[i % 3]).to(dev) for i in range(64)]
    net = B._c3_network(dev, True, tmp)
    print("batch-1 loop:", B.rate(lambda: extract_vectors(net, imgs, dev, batched=False), 64, steps=2, warmup=1))
    print("env:", {k: v for k, v in os.environ.items() if k.startswith("GANDTR") or k.startswith("GDT")})
    pr = cProfile.Profile(); pr.enable()
    extract_vectors(net, imgs, dev, batched=False)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
