"""dev tool: average launch time of the MODE-1 resblock conv (IN+ReLU folded, 64x256x64x64) under GDT_RB_ABL ablations"""
import os, sys, json
sys.path.insert(0, os.getcwd())
import torch
from gandtr_amd import engine
from gandtr_amd.tools import synth
dev = torch.device("cuda:0")
gen = engine.build_generator(synth.generator_state(0, "instance", gain=0.02), dev)
x = synth.synth_input(1, (64, 3, 256, 256), 1.0).to(dev)
for _ in range(5): gen.forward(x)
gen.set_profiling(True)
tot = {}
for _ in range(10):
    gen.forward(x); torch.cuda.synchronize()
    for kind, variant, ms, fl in gen.profile():
        if kind == 1 and variant == 910256:
            tot.setdefault("rb", []).append(ms)
ms = sorted(tot["rb"])
# 18 launches per forward: 9 MODE-1 (conv2 of each block) + 9 MODE-7/5; ablations only touch MODE 1 -> report both halves
print(json.dumps({"abl": os.environ.get("GDT_RB_ABL", "0"), "fastest_half_avg_ms": round(sum(ms[:len(ms)//2]) / (len(ms)//2), 4),
                  "slowest_half_avg_ms": round(sum(ms[len(ms)//2:]) / (len(ms) - len(ms)//2), 4)}))
