"""dev tool (GPU box): checks and timings of the "f16c" precision mode (fp16 product + block-scaled fp4 x fp6 correction product).
usage: python tools/dev_f16c.py [layer] [gen] [time]"""
import os
import sys
import time

sys.path.insert(0, os.getcwd())
import torch
import torch.nn.functional as F

from gandtr_amd.engine import HipNet, build_generator
from gandtr_amd.tools import synth

dev = torch.device("cuda:0")
what = set(sys.argv[1:]) or {"layer", "gen", "time"}


def rel(a, b):
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-12))


def single_layer(precision, cin=256, cout=256, n=8, h=64, w=64, reflect=True, norm=False, seed=0):
    g = lambda name, shape, std=1.0: synth._normal(seed, name, shape, std)
    net = HipNet(dev, precision)
    t = net.input(3)
    t = net.conv(t, g("w0", (cin, 3, 1, 1), 0.7))
    res_t = t
    if norm:
        t = net.instance_norm(t, relu=True)
    wt = g("w", (cout, cin, 3, 3), 0.05)
    bias = g("b", (cout,), 0.2)
    out = net.conv(t, wt, bias, pad=1, reflect=reflect)
    tap_in = net.output_nchw(t)
    tap_out = net.output_nchw(out)
    net.finalize()
    x = synth.synth_input(seed + 1, (n, 3, h, w))
    outs = net.forward(x.to(dev))
    xin = outs[tap_in].double().cpu()
    xi = F.pad(xin, (1,) * 4, mode="reflect") if reflect else xin
    ref = F.conv2d(xi, wt.double(), bias.double(), padding=0 if reflect else 1)
    return rel(outs[tap_out].double().cpu(), ref)


if "layer" in what:
    for prec in ("f16", "f16x3", "f16c"):
        for norm in (False, True):
            for reflect in (True, False):
                print("single 3x3 256->256 %-6s norm-fold %d reflect %d: rel err vs fp64 conv %.3e" % (prec, norm, reflect, single_layer(prec, norm=norm, reflect=reflect)), flush=True)

if "gen" in what:
    from oracle import gandtr_oracle as O
    for norm, gain in (("instance", 0.2), ("instance", 0.02), ("batch", None)):
        sd = synth.generator_state(0, norm, gain=gain or 0.02)
        x = synth.synth_input(2, (2, 3, 256, 256), 1.0)
        taps = (1, 3, 9, 10, 14, 18, 21, 24, 26)
        ref, feats = O.resnet_generator(x, sd, norm, 9, taps=taps)
        for prec in ("f16c",):
            net = build_generator(sd, dev, taps=taps, precision=prec)
            outs = net.forward(x.to(dev))
            print(norm, gain, prec, " ".join("%d:%.2e" % (t, rel(outs[net.tap_slots[t]].cpu(), feats[t])) for t in taps), flush=True)
    # production geometry (batch 64: folded norms, persistent kernels) on two images
    sd = synth.generator_state(0, "instance", gain=0.2)
    x = synth.synth_input(40, (64, 3, 256, 256), 1.0)
    net = build_generator(sd, dev, pre_tanh=True, precision="f16c")
    full = net.forward(x.to(dev))[net.out_slot]
    for i in (5, 63):
        ref = O.resnet_generator(x[i:i + 1], sd, "instance", 9, pre_tanh=True)
        print("batch-64 f16c image %d: pre-tanh rel err %.3e" % (i, rel(full[i:i + 1].cpu(), ref)), flush=True)
    again = net.forward(x.to(dev))[net.out_slot]
    print("deterministic:", bool(torch.equal(full, again)))

if "time" in what:
    sd = synth.generator_state(0, "instance")
    x = synth.synth_input(40, (64, 3, 256, 256), 1.0).to(dev)
    names = {0: "input", 1: "conv", 2: "inorm", 3: "maxpool", 4: "gem", 5: "out", 6: "hed"}
    for prec in ("f16", "f16c", "f16x3"):
        net = build_generator(sd, dev, precision=prec)
        for _ in range(3):
            net.forward(x)
        torch.cuda.synchronize()
        iters = 20 if prec != "f16x3" else 5
        t0 = time.time()
        for _ in range(iters):
            net.forward(x)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / iters
        print("%s: %.3f ms per 64-image step = %.0f images/s" % (prec, dt * 1e3, 64 / dt), flush=True)
        net.set_profiling(True)
        net.forward(x)
        agg = {}
        for kind, var, ms, fl in net.profile():
            k = (names.get(kind, kind), var)
            a = agg.setdefault(k, [0, 0.0, 0.0])
            a[0] += 1; a[1] += ms; a[2] += fl
        for k, (cnt, ms, fl) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            if ms > 0.02:
                print("    %-8s variant %-7d x%-3d %.3f ms  %s" % (k[0], k[1], cnt, ms, ("%.0f TFLOP/s" % (fl / ms / 1e9)) if fl else ""))
        net.set_profiling(False)

if "bench" in what:
    # sustained timing of the resblock conv (256 -> 256, 64 x 64, batch 64) repeated 8 times on the same input
    prec = os.environ.get("PREC", "f16c")
    net = HipNet(dev, prec)
    t = net.input(3)
    a = net.conv(t, synth._normal(0, "w0", (256, 3, 1, 1), 0.5))
    outs = [net.conv(a, synth._normal(0, "w", (256, 256, 3, 3), 0.02), pad=1, reflect=True) for _ in range(8)]
    net.gem_l2n(outs[-1], 3.0)
    net.finalize()
    x = synth.synth_input(1, (64, 3, 64, 64)).to(dev)
    for _ in range(3):
        net.forward(x)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(30):
        net.forward(x)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 30
    net.set_profiling(True); net.forward(x); p = net.profile()
    convs = [(var, ms, fl) for kind, var, ms, fl in p if kind == 1][-8:]
    other = sum(ms for kind, var, ms, fl in p) - sum(c[1] for c in convs)
    print("%s lib %s: sustained %.3f ms per forward; per conv (sustained - other) %.3f ms, profiled %.3f ms, variant %d, %.0f TFLOP/s"
          % (prec, os.path.basename(os.environ.get("GANDTR_HIP_LIB", "default")), dt * 1e3, (dt * 1e3 - other) / 8, sum(c[1] for c in convs) / 8, convs[-1][0],
             convs[-1][2] / ((dt * 1e3 - other) / 8) / 1e9), flush=True)

if "ct" in what:
    def single_ct(precision, cin, cout, n=8, h=64, w=64, norm=False, res=False, seed=0):
        g = lambda name, shape, std=1.0: synth._normal(seed, name, shape, std)
        net = HipNet(dev, precision)
        t = net.input(3)
        t0 = net.conv(t, g("w0", (cin, 3, 1, 1), 0.7))
        t = t0
        if norm:
            r = net.conv(net.input_tensor if False else t0, g("w1", (cin, cin, 1, 1), 0.1)) if res else -1
            t = net.instance_norm(t0, relu=not res, residual=r)
        wt = g("w", (cin, cout, 3, 3), 0.05)
        bias = g("b", (cout,), 0.2)
        out = net.conv(t, wt, bias, stride=2, pad=1, transposed=True)
        o2 = net.instance_norm(out, relu=True)          # makes the conv take statistics
        tap_out = net.output_nchw(out)
        tap_n = net.output_nchw(o2)
        net.finalize()
        x = synth.synth_input(seed + 1, (n, 3, h, w))
        outs = net.forward(x.to(dev))
        # reference from the same (exact) 1x1 lifts, computed in fp64 on the host
        xd = x.double()
        a0 = F.conv2d(xd, g("w0", (cin, 3, 1, 1), 0.7).double())
        a = a0
        if norm:
            a = F.instance_norm(a0, eps=1e-5)
            if res:
                a = a + F.conv2d(a0, g("w1", (cin, cin, 1, 1), 0.1).double())
            else:
                a = F.relu(a)
        ref = F.conv_transpose2d(a, wt.double(), bias.double(), stride=2, padding=1, output_padding=1)
        refn = F.relu(F.instance_norm(ref, eps=1e-5))
        return rel(outs[tap_out].double().cpu(), ref), rel(outs[tap_n].double().cpu(), refn)
    for prec in ("f16x3", "f16c"):
        for (cin, cout) in ((256, 128), (128, 64)):
            for norm, res in ((False, False), (True, False), (True, True)):
                if res and cin != 256:
                    continue
                print("convT %d->%d %-6s norm %d res %d: rel err (conv, IN+ReLU of it) %s" % (cin, cout, prec, norm, res, "%.3e %.3e" % single_ct(prec, cin, cout, norm=norm, res=res)), flush=True)
