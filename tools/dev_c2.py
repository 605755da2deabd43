"""dev tool: BASELINE config 3 (hedngan generator + HED edge branch, 64x3x256x256) split into its legs, through the hub / wrapper API
and through the engine, to find where the HED leg's time goes."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
import hubconf
from gandtr_amd import engine
from gandtr_amd.learning import network as N
from gandtr_amd.tools import synth

dev = torch.device("cuda:0")


def rate(fn, steps=10, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


with torch.no_grad():
    gen = hubconf.hedngan(pretrained=False, device=dev)
    gen.model.load_state_dict(synth.generator_state(0, "batch"))
    hed = N.initialize_network({"type": "SingleNetwork", "model": {"architecture": "hed_interpolation"}, "initialize": False,
                                "runtime": {"wrappers": "rgb2bgr_pre, meanstd_pre:[[0.5,0.5,0.5],[0.5,0.5,0.5]]:"
                                                        "[[0.40787054,0.45752458,0.48109378],[1,1,1]]"}}, dev).eval()
    hed.model.load_state_dict(synth.hed_state(0))
    x = synth.synth_input(3, (64, 3, 256, 256), 1.0).to(dev)
    for prec in ("f16", "f16c"):
        gen.model.hip_precision = prec
        y = gen(x)
        print("generator (%s) via hub: %.2f ms" % (prec, rate(lambda: gen(x))))
        print("hed(gen(x)) via hub (%s generator): %.2f ms" % (prec, rate(lambda: hed(gen(x)))))
    print("HED leg alone via hub (wrappers folded into the input pack): %.2f ms" % rate(lambda: hed(y)))
    # the same with the wrappers as torch ops (the round-1 hub path)
    ws = hed.wrappers["eval"].wrappers
    def torch_wrappers():
        t = y
        for w in ws:
            t, _ = w.preprocess(t, None)
        return hed.model(t)
    print("HED leg alone, wrappers as torch ops + plain HED net: %.2f ms" % rate(torch_wrappers))
    net = engine.build_hed(synth.hed_state(0), dev, perm=[2, 1, 0], in_affine=([0.5] * 3, [0.09212946, 0.04247542, 0.01890622]))
    print("HED leg alone via engine: %.2f ms" % rate(lambda: net.forward(y)))
    net.set_profiling(True)
    net.forward(y); torch.cuda.synchronize()
    names = {0: "input", 1: "conv", 2: "inorm", 3: "maxpool", 4: "gem", 5: "tap", 6: "hed"}
    tot = 0.0
    for i, (k, t, ms, fl) in enumerate(net.profile()):
        tot += ms
        print("  %2d %-8s var %6d %7.3f ms %7.1f TFLOP/s" % (i, names[k], t, ms, fl / ms / 1e9 if ms > 0 else 0))
    print("  sum of ops %.3f ms" % tot)
