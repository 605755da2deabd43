"""CPU emulation: how much of the f16c error budget does the WEIGHT correction term buy?  (profiles/experiments: not product code)

Every conv of the generator is evaluated in fp64 on perturbed operands:
    a -> fp16(a) [+ fp4 e2m1 of (a - fp16(a)) at scale 2^-12]      (activation side)
    w -> fp16(w) [+ e2m3 block-scaled (w - fp16(w))]                (weight side)
and the taps are compared with the unperturbed fp64 forward.  Prints max|delta|/max|ref| at the pre-tanh output.
usage: python profiles/experiments/emulate_f16c_lite.py [size] [batch]"""
import sys, os
sys.path.insert(0, os.getcwd())
import torch
import torch.nn.functional as F
from gandtr_amd.tools import synth

torch.set_num_threads(8)
size = int(sys.argv[1]) if len(sys.argv) > 1 else 128
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 2


def fp4(x, scale_exp):                     # e2m1: values {0, .5, 1, 1.5, 2, 3, 4, 6} * 2^scale_exp, round to nearest, saturate
    s = 2.0 ** scale_exp
    y = (x / s).clamp(-6, 6)
    grid = torch.tensor([0, .5, 1, 1.5, 2, 3, 4, 6], dtype=x.dtype)
    a = y.abs()
    idx = (a.unsqueeze(-1) - grid).abs().argmin(-1)
    return torch.sign(y) * grid[idx] * s


def e2m3_block(w, kdim_block=32):           # per (cout, 32 k) block scale = 2^ceil(log2(max/7.5)); e2m3 grid
    co = w.shape[0]
    flat = w.reshape(co, -1)
    K = flat.shape[1]
    pad = (-K) % kdim_block
    f = F.pad(flat, (0, pad)).reshape(co, -1, kdim_block)
    mx = f.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    sc = 2.0 ** torch.ceil(torch.log2(mx / 7.5))
    y = (f / sc).clamp(-7.5, 7.5)
    a = y.abs()
    e = torch.floor(torch.log2(a.clamp_min(2.0 ** -20))).clamp(0, 2)        # normal exponents 0..2 (bias 1), subnormal step 0.125
    step = torch.where(a < 1, torch.full_like(a, 0.125), 2.0 ** e / 8)
    q = torch.round(a / step) * step
    out = (torch.sign(y) * q * sc).reshape(co, -1)[:, :K]
    return out.reshape(w.shape)


def q_act(a, corr):
    hi = a.half().double()
    return hi + fp4(a - hi, -12 - 1) * 1.0 if corr else hi          # (stored * 2^12, fp4 grid step .5 -> 2^-13 resolution)


def q_w(w, corr):
    hi = w.half().double()
    return hi + e2m3_block(w - hi) if corr else hi


def forward(x, sd, acorr, wcorr, exact=False):
    qa = (lambda t: t) if exact else (lambda t: q_act(t, acorr))
    qw = (lambda t: t) if exact else (lambda t: q_w(t, wcorr))
    IN = lambda t: F.instance_norm(t, eps=1e-5)
    h = F.conv2d(F.pad(x, (3,) * 4, mode="reflect"), sd["model.1.weight"], sd["model.1.bias"])     # stem: fp32-class in every mode
    h = F.relu(IN(h))
    i = 4
    for _ in range(2):
        h = F.conv2d(qa(h), qw(sd["model.%d.weight" % i]), sd["model.%d.bias" % i], stride=2, padding=1)
        h = F.relu(IN(h)); i += 3
    for _ in range(9):
        p = "model.%d.conv_block." % i
        r = F.conv2d(F.pad(qa(h), (1,) * 4, mode="reflect"), qw(sd[p + "1.weight"]), sd[p + "1.bias"])
        r = F.relu(IN(r))
        r = F.conv2d(F.pad(qa(r), (1,) * 4, mode="reflect"), qw(sd[p + "5.weight"]), sd[p + "5.bias"])
        h = h + IN(r); i += 1
    for _ in range(2):
        h = F.conv_transpose2d(qa(h), qw(sd["model.%d.weight" % i]), sd["model.%d.bias" % i], stride=2, padding=1, output_padding=1)
        h = F.relu(IN(h)); i += 3
    hq = h if exact else h.half().double()                                                       # head: single fp16 pass
    wq = sd["model.%d.weight" % (i + 1)] if exact else sd["model.%d.weight" % (i + 1)].half().double()
    return F.conv2d(F.pad(hq, (3,) * 4, mode="reflect"), wq, sd["model.%d.bias" % (i + 1)])


for seed in (0, 1):
    sd = {k: v.double() for k, v in synth.generator_state(seed, "instance").items()}
    x = synth.synth_input(40 + seed, (batch, 3, size, size), 1.0).double()
    ref = forward(x, sd, False, False, exact=True)
    for name, ac, wc in (("a + w corrected (f16c)", True, True), ("a corrected only", True, False), ("w corrected only", False, True), ("none (f16 operands, fp32 storage)", False, False)):
        out = forward(x, sd, ac, wc)
        print("seed %d  %-36s pre-tanh max|d|/max|ref| = %.2e" % (seed, name, float((out - ref).abs().max() / ref.abs().max())), flush=True)
