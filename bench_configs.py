"""Throughput of every BASELINE.json configuration on ONE MI355X (development tool; bench.py is the driver's contract).

  c0  cyclegan forward, 4x3x256x256, hub entrypoint on the CPU (plumbing)             -> images/s
  c1  gem_vgg16_cyclegan descriptors, batch 32x3x1024x1024, single scale              -> descriptors/s
  c2  hedngan (BatchNorm) generator + HED edge branch, batch 64x3x256x256             -> images/s
  c3  gem_resnet101_hedngan multi-scale descriptors + lw whitening, 3x1024x1024       -> descriptors/s
      (scales hub default {1, 1/sqrt2, 1/2} and 'sms' {1, 1/sqrt2, sqrt2}; one rank's share of the 8-GPU job)
  c4  augment-then-embed: cyclegan on 128x3x256x256 -> meanstd_post -> GeM-ResNet101  -> images/s
      (+ the same chain with the reference's clahepost step, and the CLAHE / retrieval 'next' rows alone)
Everything runs through the hub / wrapper / network-container API of the host mirror (the drop-in surface).

`python bench_configs.py --gpus N` runs BASELINE configs 4 and 5 as north_star states them -- one process per GPU (the script starts its own
`torch.distributed.run` child, as bench.py does), weights replicated, the batch sharded in contiguous chunks, ONE RCCL all-gather of the
descriptor block at the end (generator outputs are not gathered):
  c3  per-rank chunk (8 x 3x1024x1024) -> 3-scale pyramid -> GeM-ResNet101 -> multi-scale aggregation -> lw whitening -> all-gather  -> descriptors/s
  c4  per-rank chunk (128 x 3x256x256) -> cyclegan -> meanstd_post -> GeM-ResNet101 -> all-gather                                     -> images/s
`--dry-run` rehearses both on the CPU with gloo (tiny images, the same code path through wrappers / containers / sharding) and checks on
every rank that the gathered D x N matrix equals, bit for bit, the single-process result over the same chunks (tests/test_bench_launcher.py).
"""
import json
import math
import os
import pickle
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import hubconf                                         # noqa: E402
from gandtr_amd.learning import network as N          # noqa: E402
from gandtr_amd.learning.checkpoints import Checkpoints  # noqa: E402
from gandtr_amd.tools import synth                    # noqa: E402


def rate(fn, units, steps=8, warmup=2):
    """(units per second, ms per call).  Each call is timed on its own and the MEDIAN is reported: a single stall inside a short window -- a hipMalloc when the
    caching allocator regrows after the previous configuration's empty_cache() -- moved an 8-step mean by 3-4 ms ("the config-3 regression")."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    times = []
    for _ in range(steps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    times.sort()
    dt = times[len(times) // 2] if steps >= 3 else sum(times) / len(times)
    return round(units / dt, 1), round(dt * 1e3, 2)


def rate_back_to_back(fn, units, steps=8, warmup=2):
    """(units per second, ms per call) of `steps` calls issued back to back between two synchronisations -- bench.py's protocol (the driver's contract): the host
    issues call k + 1 while the device works on call k.  `rate` above synchronises after every call and so measures a call's LATENCY, host issue time included."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return round(units / dt, 1), round(dt * 1e3, 2)


def _c3_network(dev, scales, tmp):
    """gem_resnet101 as the hub's pretrained call path builds it: checkpoint + lw.pkl -> whitening + multi-scale wrappers"""
    base = hubconf.gem_resnet101_hedngan(pretrained=False, device="cpu")
    base.model.load_state_dict(synth.resnet101_state(0))
    sd = base.state_dict()["net"]
    sd["network_params"]["runtime"]["data"] = {"transforms": "pil2np | totensor | normalize",
                                               "mean_std": [[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]]}
    ck, lw = os.path.join(tmp, "r101.pth"), os.path.join(tmp, "lw.pkl")
    torch.save(sd, ck)
    with open(lw, "wb") as f:
        pickle.dump(synth.whitening_state(0, 2048), f)
    runtime = {"wrappers": {"train": None, "eval": {"0_cirwhiten": {"whitening": lw, "dimensions": None}, "1_cirmultiscale": {"scales": scales}}}}
    return N.initialize_network(None, dev, Checkpoints.load_network(ck), runtime).eval()


def _c4_chain(dev):
    gen_p = {"type": "SingleNetwork",
             "model": {"architecture": "official_resnet_generator", "input_nc": 3, "output_nc": 3, "n_blocks": 9,
                       "norm_layer": "instance", "no_antialias": True, "no_antialias_up": True},
             "initialize": False,
             "runtime": {"wrappers": "meanstd_post:[[0.5,0.5,0.5],[0.5,0.5,0.5]]:[[0.485,0.456,0.406],[0.229,0.224,0.225]]",
                         "data": {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.5] * 3, [0.5] * 3]}}}
    emb_p = {"type": "SingleNetwork",
             "model": {"architecture": "cirnet", "cir_architecture": "resnet101", "local_whitening": False, "pooling": "gem",
                       "pretrained": False, "regional": False, "whitening": False},
             "initialize": False,
             "runtime": {"wrappers": "cirfaketuplebatch",
                         "data": {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.5] * 3, [0.5] * 3]}}}
    chain = N.initialize_network({"type": "CirSequentialNetwork", "sequence": "augment,embed", "augment": gen_p, "embed": emb_p}, dev).eval()
    chain.networks["augment"].model.load_state_dict(synth.generator_state(0, "instance"))
    chain.networks["embed"].model.load_state_dict(synth.resnet101_state(0))
    return chain


def sharded_main(args):
    """configs 4 / 5 over the ranks of one node (module docstring).  Launched by `torch.distributed.run`: RANK / LOCAL_RANK / WORLD_SIZE."""
    import torch.distributed as dist
    from gandtr_amd import sharding
    world, rank, local = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry_run
    dist.init_process_group("gloo" if dry else "nccl")
    if dry:
        dev = torch.device("cpu")
        torch.set_num_threads(2)
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)

    def sync():
        if not dry:
            torch.cuda.synchronize()
        dist.barrier()

    def timed(fn, steps, warmup):
        for _ in range(warmup):
            fn()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        sync()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if not dry else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / steps

    out = {"n_gpus": world, "rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "dry_run": dry}
    tmp = tempfile.mkdtemp()
    steps, warmup = (1, 0) if dry else (args.steps, 1)
    small = dry or args.small          # tiny images: the CPU rehearsal, or the GPU test of the RCCL path (tests/test_hip_rccl_world1.py)
    check = dry or args.check          # gathered == single-process result over the same chunks, bit for bit, on EVERY rank
    with torch.no_grad():
        # ---- c3: multi-scale + whitening, batch-sharded, one all-gather
        per_rank, side = ((2, 64) if dry else (2, 256)) if small else (8, 1024)
        # (--global-batch: a global batch that the ranks do not divide -- contiguous chunks of ceil(N / world) images, the last chunk short or
        #  empty, zero-padded for the collective and trimmed afterwards: sharding.chunk_bounds / all_gather_descriptors)
        n_glob = args.global_batch if args.global_batch else per_rank * world
        per_rank = (n_glob + world - 1) // world
        x = synth.synth_input(4, (n_glob, 3, side, side)).to(dev)          # every rank synthesises the same global batch
        for tag, scales in (("hub_default", True), ("sms", "sms")):
            net = _c3_network(dev, scales, tmp)
            dt = timed(lambda: sharding.embed_sharded(net, x), steps, warmup)
            got = sharding.embed_sharded(net, x)
            rec = {"descriptors_per_s": round(n_glob / dt, 1), "ms_per_step": round(dt * 1e3, 2), "global_batch": n_glob,
                   "image": "%dx%d" % (side, side), "gathered": list(got.shape)}
            if check:          # the multi-GPU contract: gathered == single-process result over the same chunks, bit for bit, on EVERY rank
                ref = sharding.descriptors_in_chunks(net, x, per_rank)
                ok = torch.tensor([int(torch.equal(got, ref))], device=dev)
                dist.all_reduce(ok, op=dist.ReduceOp.MIN)
                rec["sharded_equals_single_process_bitwise"] = bool(ok.item())
            out["c3_gem_resnet101_ms_%s" % tag] = rec
            del net
        # ---- c4: augment -> embed, 128 images per rank, descriptors gathered, generator outputs not
        per_rank, side = ((2, 32) if dry else (8, 64)) if small else (128, 256)
        n_glob = args.global_batch if args.global_batch else per_rank * world
        per_rank = (n_glob + world - 1) // world
        x = synth.synth_input(5, (n_glob, 3, side, side), 1.0).to(dev)
        chain = _c4_chain(dev)
        dt = timed(lambda: sharding.embed_sharded(chain, x), steps, warmup)
        got = sharding.embed_sharded(chain, x)
        rec = {"images_per_s": round(n_glob / dt, 1), "ms_per_step": round(dt * 1e3, 2), "global_batch": n_glob,
               "image": "%dx%d" % (side, side), "gathered": list(got.shape)}
        if check:
            ref = sharding.descriptors_in_chunks(chain, x, per_rank)
            ok = torch.tensor([int(torch.equal(got, ref))], device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            rec["sharded_equals_single_process_bitwise"] = bool(ok.item())
        out["c4_augment_then_embed"] = rec
    if rank == 0:
        print(json.dumps(out, indent=1), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def self_launch(argv, gpus):
    """the parent never touches the GPU: it starts `python -m torch.distributed.run --nproc-per-node N bench_configs.py ...` as a CHILD
    process and relays its output and exit code (no exec of a process that has initialised the GPU)"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env, check=False).returncode


def _cpu_baseline(fn, units, unit, sample, seconds=8.0):
    """The CPU oracle (oracle/gandtr_oracle.py: the reference's torch ops restated) timed on a bounded sample of the config's workload, on the threads bench.py
    uses (the cores of this process, at most 16), for about `seconds`: {"value", "unit", "cores", "kind": "port", "sample"}"""
    import bench
    cores = bench.host_cores()
    torch.set_num_threads(cores)
    with torch.no_grad():
        fn()
        n, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < seconds:
            fn()
            n += units
        dt = time.perf_counter() - t0
    return {"value": round(n / dt, 3), "unit": unit, "cores": cores, "kind": "port",
            "sample": "%s: %d in %.1f s, torch CPU fp32 oracle, %d threads" % (sample, n, dt, cores)}


def _roofline_of(model, x):
    """bench.py's live per-kernel roofline object (HIP events around every op inside the library) for the engine net a hub model has built for its last forward"""
    import bench
    nets = [v for k, v in getattr(model, "_hip_cache", {}).items() if k != "__stamp__"]
    if len(nets) > 1:                                   # (the generator of c2 also ran in the opt-in fp16 mode: report the default mode's net)
        nets = [n for n in nets if getattr(n, "precision", None) == "f16c"]
    if len(nets) != 1:
        return None
    return bench.conv_roofline(nets[0], x, steps=2)


def main(args):
    dev = torch.device("cuda:0")
    out = {}
    only = [k for k in (args.only or "").split(",") if k]

    def want(name):                     # --only c3,extract: run some of the sections (dev: bisecting, quick re-measurements)
        return not only or name in only
    tmp = tempfile.mkdtemp()
    with torch.no_grad():
        if want("c0"):
            # c0 (CPU plumbing)
            net = hubconf.cyclegan(pretrained=False, device="cpu")
            x = synth.synth_input(1, (4, 3, 256, 256), 1.0)
            torch.set_num_threads(min(16, os.cpu_count() or 1))
            net(x)
            t0 = time.perf_counter(); net(x); dt = time.perf_counter() - t0
            out["c0_cyclegan_cpu_4x256"] = {"images_per_s": round(4 / dt, 2), "threads": torch.get_num_threads()}

        if want("c1"):
            # c1
            net = hubconf.gem_vgg16_cyclegan(pretrained=False, device=dev)
            net.model.load_state_dict(synth.vgg16_state(0))
            x = synth.synth_input(2, (32, 3, 1024, 1024)).to(dev)
            r, ms = rate(lambda: net(x), 32, steps=4, warmup=1)
            out["c1_gem_vgg16_32x1024"] = {"descriptors_per_s": r, "ms_per_batch": ms, "tflops": round(r * 641.4 / 1e3, 1),
                                           "roofline": _roofline_of(net.model, x)}
            if not args.no_cpu_baseline:
                from oracle import gandtr_oracle as O
                sdv, xc = synth.vgg16_state(0), synth.synth_input(2, (1, 3, 1024, 1024))
                out["c1_gem_vgg16_32x1024"]["cpu_baseline"] = _cpu_baseline(lambda: O.image_retrieval_forward(xc, sdv, "vgg16"), 1, "descriptors/s", "1x3x1024x1024 images")
            del net, x
            torch.cuda.empty_cache()

        if want("c2"):
            # c2: hedngan generator (BN) + HED branch with its wrappers (edges_epochs.py:87, forward only)
            gen = hubconf.hedngan(pretrained=False, device=dev)
            gen.model.load_state_dict(synth.generator_state(0, "batch"))
            hed = N.initialize_network({"type": "SingleNetwork", "model": {"architecture": "hed_interpolation"}, "initialize": False,
                                        "runtime": {"wrappers": "rgb2bgr_pre, meanstd_pre:[[0.5,0.5,0.5],[0.5,0.5,0.5]]:"
                                                                "[[0.40787054,0.45752458,0.48109378],[1,1,1]]"}}, dev).eval()
            hed.model.load_state_dict(synth.hed_state(0))
            x = synth.synth_input(3, (64, 3, 256, 256), 1.0).to(dev)
            # the generator's default precision is "f16c" (north_star's 1e-3); the single-pass fp16 numbers are the opt-in fast mode
            r, ms = rate(lambda: hed(gen(x)), 64)
            rg, msg = rate(lambda: gen(x), 64)
            y = gen(x)
            rh, msh = rate(lambda: hed(y), 64)
            gen.model.hip_precision = "f16"
            rf, msf = rate(lambda: hed(gen(x)), 64)
            rgf, _ = rate(lambda: gen(x), 64)
            gen.model.hip_precision = None
            out["c2_hedngan_plus_hed_64x256"] = {"images_per_s": r, "ms_per_batch": ms, "generator_only_images_per_s": rg,
                                                 "hed_leg_ms": msh, "tflops": round(r * 139.2 / 1e3, 1),
                                                 "fast_mode_f16": {"images_per_s": rf, "ms_per_batch": msf, "generator_only_images_per_s": rgf}}
            # regression guard (round 1 saw this leg at 4.2 -> 6.5 ms with no kernel change; root cause in DESIGN.md section 6): the HED leg
            # (40.1 GFLOP / image, wrappers folded into its input pack) must stay under 4.5 ms per 64-image batch
            out["c2_hedngan_plus_hed_64x256"]["hed_leg_within_4p5_ms"] = bool(msh < 4.5)
            out["c2_hedngan_plus_hed_64x256"]["roofline"] = _roofline_of(gen.model, x)          # (the generator leg: 99.1 of the 139.2 GFLOP per image)
            out["c2_hedngan_plus_hed_64x256"]["roofline_hed_leg"] = _roofline_of(hed.model, gen(x))
            if not args.no_cpu_baseline:
                from oracle import gandtr_oracle as O
                sdg, sdh, xc = synth.generator_state(0, "batch"), synth.hed_state(0), synth.synth_input(3, (4, 3, 256, 256), 1.0)
                out["c2_hedngan_plus_hed_64x256"]["cpu_baseline"] = _cpu_baseline(
                    lambda: O.hed_on_generator_output(O.resnet_generator(xc, sdg, "batch", 9), sdh), 4, "images/s", "4x3x256x256 batches (BatchNorm generator + HED)")
            del y
            del gen, hed, x
            torch.cuda.empty_cache()

        if want("c3"):
            # c3: pretrained-style path (checkpoint + lw.pkl -> whiten + multiscale wrappers), batch of 8 per call
            base = hubconf.gem_resnet101_hedngan(pretrained=False, device="cpu")
            base.model.load_state_dict(synth.resnet101_state(0))
            sd = base.state_dict()["net"]
            sd["network_params"]["runtime"]["data"] = {"transforms": "pil2np | totensor | normalize",
                                                       "mean_std": [[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]]}
            torch.save(sd, os.path.join(tmp, "r101.pth"))
            with open(os.path.join(tmp, "lw.pkl"), "wb") as f:
                pickle.dump(synth.whitening_state(0, 2048), f)
            x = synth.synth_input(4, (8, 3, 1024, 1024)).to(dev)
            for tag, scales in (("hub_default", True), ("sms", "sms")):
                runtime = {"wrappers": {"train": None, "eval": {"0_cirwhiten": {"whitening": os.path.join(tmp, "lw.pkl"), "dimensions": None},
                                                                "1_cirmultiscale": {"scales": scales}}}}
                net = N.initialize_network(None, dev, Checkpoints.load_network(os.path.join(tmp, "r101.pth")), runtime).eval()
                rl, msl = rate(lambda: net(x), 8, steps=4, warmup=1)
                r, ms = rate_back_to_back(lambda: net(x), 8, steps=8, warmup=2)
                gf = 574.8 if scales is True else 1151.9
                out["c3_gem_resnet101_ms_%s_8x1024" % tag] = {"descriptors_per_s": r, "ms_per_batch": ms, "tflops": round(r * gf / 1e3, 1),
                                                              "synchronised_per_call": {"descriptors_per_s": rl, "ms_per_call": msl,
                                                                                        "note": "a call's latency: ~300 launches issued by the host before the device can finish"},
                                                              "roofline": _roofline_of(net.model, x),      # (the full-size level of the pyramid, 8 x 1024^2)
                                                              "roofline_note": "per-kernel figures of the scale-1 level alone (8 x 3 x 1024 x 1024); the three levels run concurrently on side streams"}
                if not args.no_cpu_baseline:
                    from oracle import gandtr_oracle as O
                    sdr, lw, xc = synth.resnet101_state(0), synth.whitening_state(0, 2048), synth.synth_input(4, (1, 3, 1024, 1024))
                    Pm = (torch.from_numpy(lw["P"]), torch.from_numpy(lw["m"]))
                    sc = O.SCALE_PRESETS[scales]
                    out["c3_gem_resnet101_ms_%s_8x1024" % tag]["cpu_baseline"] = _cpu_baseline(
                        lambda: O.embed_ms_whiten(xc, sdr, "resnet101", sc, Pm[0], Pm[1]), 1, "descriptors/s", "1x3x1024x1024 images, %d-level pyramid + whitening" % len(sc), seconds=6.0)
                if scales is True:                      # the same network on a batch of 32 (what a rank holds when the global batch is 256): the large-geometry kernels apply
                    x32 = synth.synth_input(5, (32, 3, 1024, 1024)).to(dev)
                    r32, ms32 = rate_back_to_back(lambda: net(x32), 32, steps=4, warmup=1)
                    out["c3_gem_resnet101_ms_%s_32x1024" % tag] = {"descriptors_per_s": r32, "ms_per_batch": ms32, "tflops": round(r32 * gf / 1e3, 1)}
                    del x32
                del net
            del x
            torch.cuda.empty_cache()

        if want("extract"):
            # the validate-stage caller (mdir/external/cirtorch/networks/imageretrievalnet.py:312-339): 64 database images of 3 sizes through the
            # multi-scale + whitening network, image by image (the reference's batch-1 loop) vs equal sizes grouped into one forward per group
            from gandtr_amd.stages.validate import extract_vectors
            net = _c3_network(dev, True, tmp)
            sizes = [(768, 1024), (1024, 768), (1024, 1024)]
            imgs = [synth.synth_input(300 + i, (3,) + sizes[i % 3]).to(dev) for i in range(64)]
            r1, ms1 = rate(lambda: extract_vectors(net, imgs, dev, batched=False), 64, steps=2, warmup=1)
            r2, ms2 = rate(lambda: extract_vectors(net, imgs, dev, batched=True), 64, steps=2, warmup=1)
            out["extract_vectors_64_images_3_sizes_ms_whiten"] = {"batch1_loop_descriptors_per_s": r1, "equal_sizes_batched_descriptors_per_s": r2,
                                                                   "speedup": round(r2 / r1, 2)}
            del net, imgs
            torch.cuda.empty_cache()

        if want("c4"):
            # c4: augment -> embed chain through CirSequentialNetwork
            gen_p = {"type": "SingleNetwork",
                     "model": {"architecture": "official_resnet_generator", "input_nc": 3, "output_nc": 3, "n_blocks": 9,
                               "norm_layer": "instance", "no_antialias": True, "no_antialias_up": True},
                     "initialize": False,
                     "runtime": {"wrappers": "meanstd_post:[[0.5,0.5,0.5],[0.5,0.5,0.5]]:[[0.485,0.456,0.406],[0.229,0.224,0.225]]",
                                 "data": {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.5] * 3, [0.5] * 3]}}}
            emb_p = {"type": "SingleNetwork",
                     "model": {"architecture": "cirnet", "cir_architecture": "resnet101", "local_whitening": False, "pooling": "gem",
                               "pretrained": False, "regional": False, "whitening": False},
                     "initialize": False,
                     "runtime": {"wrappers": "cirfaketuplebatch",
                                 "data": {"transforms": "pil2np | totensor | normalize", "mean_std": [[0.5] * 3, [0.5] * 3]}}}
            gen_c, emb_c = json.loads(json.dumps(gen_p)), json.loads(json.dumps(emb_p))      # initialize_network consumes its params
            chain = N.initialize_network({"type": "CirSequentialNetwork", "sequence": "augment,embed", "augment": gen_p, "embed": emb_p},
                                         dev).eval()
            chain.networks["augment"].model.load_state_dict(synth.generator_state(0, "instance"))
            chain.networks["embed"].model.load_state_dict(synth.resnet101_state(0))
            x = synth.synth_input(5, (128, 3, 256, 256), 1.0).to(dev)
            r, ms = rate(lambda: chain(x), 128)
            chain.networks["augment"].model.hip_precision = "f16"
            rf, msf = rate(lambda: chain(x), 128)
            chain.networks["augment"].model.hip_precision = None
            out["c4_augment_then_embed_128x256"] = {"images_per_s": r, "ms_per_batch": ms, "tflops": round(r * 119.5 / 1e3, 1),
                                                    "fast_mode_f16_generator": {"images_per_s": rf, "ms_per_batch": msf},
                                                    "roofline": _roofline_of(chain.networks["augment"].model, x)}       # (the generator leg: 99.1 of the 119.5 GFLOP per image)
            if not args.no_cpu_baseline:
                from oracle import gandtr_oracle as O
                sdg, sdr, xc = synth.generator_state(0, "instance"), synth.resnet101_state(0), synth.synth_input(5, (4, 3, 256, 256), 1.0)
                ms_gen, ms_emb = [[0.5] * 3, [0.5] * 3], [[0.485, 0.456, 0.406], [0.229, 0.224, 0.225]]
                out["c4_augment_then_embed_128x256"]["cpu_baseline"] = _cpu_baseline(
                    lambda: O.image_retrieval_forward(O.meanstd_adapt(O.resnet_generator(xc, sdg, "instance", 9), ms_gen, ms_emb), sdr, "resnet101"), 4, "images/s",
                    "4x3x256x256 batches (generator -> meanstd_post -> GeM-ResNet-101)")
            # the same chain with the reference's CLAHE step between generator and embedder (finetune.yml:13: wrappers
            # meanstd_post, clahepost -- post-processing runs in reverse order: CLAHE first, then the ImageNet mean / std)
            gen_c["runtime"]["wrappers"] += ",clahepost:[[0.5,0.5,0.5],[0.5,0.5,0.5]]:1.0"
            chain_c = N.initialize_network({"type": "CirSequentialNetwork", "sequence": "augment,embed", "augment": gen_c, "embed": emb_c},
                                           dev).eval()
            chain_c.networks["augment"].model.load_state_dict(synth.generator_state(0, "instance"))
            chain_c.networks["embed"].model.load_state_dict(synth.resnet101_state(0))
            r, ms = rate(lambda: chain_c(x), 128)
            out["c4_with_clahepost_128x256"] = {"images_per_s": r, "ms_per_batch": ms}
            # "next" row: CLAHE post-processing alone, 38 algorithmic bytes per pixel
            from gandtr_amd import clahe
            y = chain.networks["augment"].model(x)
            pair = ([0.5] * 3, [0.5] * 3)
            r, ms = rate(lambda: clahe.clahe_lab(y, 1.0, 8, pair, pair), 128, steps=50, warmup=5)
            out["next_clahe_post_128x256"] = {"images_per_s": r, "us": round(128 / r * 1e6, 1), "hbm_GBps_algorithmic": round(38 * 65536 * r / 1e9, 1)}
            del chain_c, y
            # "next" row: ingest of one decoded 1200x1600 photo -> thumbnail 1024 -> CLAHE -> normalised CHW
            from gandtr_amd import ingest
            import numpy as np
            photo = torch.from_numpy(np.random.default_rng(0).integers(0, 256, (1200, 1600, 3)).astype(np.uint8)).to(dev)
            mean, std = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
            r, ms = rate(lambda: ingest.ingest(photo, 1024, mean, std), 1, steps=200, warmup=10)
            r2, _ = rate(lambda: ingest.ingest(photo, 1024, mean, std, clahe_clip=1.0), 1, steps=200, warmup=10)
            out["next_ingest_1200x1600_to_1024"] = {"images_per_s": r, "us": round(1e6 / r, 1), "with_clahe_images_per_s": r2,
                                                    "with_clahe_us": round(1e6 / r2, 1),
                                                    "note": "wall time per Python call of ingest.ingest, calls back to back: host-bound (wrapper + output allocation + "
                                                            "launches); the two resampling kernels themselves take 16.9 + 7.6 us (profiles/r01_ingest_1200x1600_kernel_stats.csv)"}
            rng = np.random.default_rng(1)                      # a list of 64 photos of mixed sizes through the batched entry point (one call)
            mixed = [torch.from_numpy(rng.integers(0, 256, (int(h), int(w), 3)).astype(np.uint8)).to(dev)
                     for h, w in zip(rng.integers(700, 1500, 64), rng.integers(900, 2000, 64))]
            r3, _ = rate(lambda: ingest.ingest_many(mixed, 1024, mean, std), 64, steps=20, warmup=3)
            out["next_ingest_1200x1600_to_1024"]["mixed_64_batched_images_per_s"] = r3
            del mixed
            # "next" row (first half): JPEG files -> decoded pixels on the device (tools/jpeg_bench.py: 64 photo-like
            # 1024x768 4:2:0 files, quality 90) against the reference's loader (Pillow, one host thread), and files -> normalised 362-pixel tensors
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import jpeg_bench
            out["next_jpeg_decode_64_files_1024x768"] = jpeg_bench.run()
            # "next" row: learned whitening, D = 2048, 20 k vectors, 8 k pairs (float64)
            from gandtr_amd import whiten_learn
            gq = torch.Generator(device=dev).manual_seed(0)
            desc = torch.nn.functional.normalize(torch.randn(20000, 2048, generator=gq, device=dev) *
                                                 torch.linspace(1.5, 0.2, 2048, device=dev)[None, :], dim=1)
            qi = torch.randint(0, 10000, (8000,), generator=torch.Generator().manual_seed(1))
            torch.cuda.synchronize(); t0 = time.perf_counter()
            whiten_learn.whitenlearn(desc.t(), qi, qi + 10000)
            torch.cuda.synchronize()
            out["next_whiten_learn_d2048_20k_vectors"] = {"seconds": round(time.perf_counter() - t0, 3)}
            del desc
            # "next" row: retrieval scoring, revisitop-style: 200k database x 70 queries, D = 2048
            from gandtr_amd import retrieval
            import numpy as np
            d, ndb, nq = 2048, 200000, 70
            vecs = torch.nn.functional.normalize(torch.randn(ndb, d, device=dev), dim=1).t()      # D x Ndb view of [Ndb][D]
            qv = torch.nn.functional.normalize(torch.randn(nq, d, device=dev), dim=1).t()
            r, ms = rate(lambda: retrieval.scores_and_ranks(vecs, qv), nq, steps=5, warmup=2)
            vc, qc = vecs[:, :20000].t().contiguous().cpu().numpy(), qv.cpu().numpy()
            t0 = time.perf_counter(); s = np.dot(vc, qc); np.argsort(-s, axis=0); cpu = (time.perf_counter() - t0) * (ndb / 20000)
            out["next_retrieval_200k_x_70_d2048"] = {"queries_per_s": r, "ms": ms, "gemm_gflop": round(2.0 * ndb * nq * d / 1e9, 1),
                                                     "cpu_numpy_ms_extrapolated_from_20k": round(cpu * 1e3, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--dry-run", action="store_true", help="configs 4 / 5 sharded over N gloo ranks on the CPU, tiny images, bitwise check against one process")
    ap.add_argument("--global-batch", type=int, default=0, help="sharded configs: images in the global batch (default: 8 / 128 per rank; any N, the ranks need not divide it)")
    ap.add_argument("--small", action="store_true", help="sharded configs on the GPU with small images (2 x 256^2 / 8 x 64^2 per rank): the RCCL path's functional test")
    ap.add_argument("--check", action="store_true", help="sharded configs: every rank compares the gathered D x N matrix with the single-process result over the same chunks, bit for bit")
    ap.add_argument("--only", default="", help="comma-separated sections of the one-GPU run: c0,c1,c2,c3,extract,c4 (default: all)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle legs (about 8 s each) of the config lines")
    ap.add_argument("--sharded", action="store_true", help="run the sharded configs 4 / 5 even with --gpus 1 (one rank: the same code path, RCCL world size 1)")
    a = ap.parse_args()
    if "RANK" in os.environ and (a.gpus > 1 or a.dry_run or a.sharded):
        sharded_main(a)
    elif a.gpus > 1 or a.dry_run or a.sharded:
        sys.exit(self_launch(sys.argv[1:], max(1, a.gpus)))
    else:
        main(a)
