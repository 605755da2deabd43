"""bench.py launches its own ranks for --gpus N (one process per GPU): rehearsed here without a GPU in
--dry-run mode (gloo backend, CPU tensors, the same barrier / max-over-ranks timing and descriptor all-gather)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       timeout=600, check=False)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    return p.returncode, lines, p.stderr.decode()


def test_self_launch_two_ranks_dry_run():
    rc, lines, err = _run(["--gpus", "2", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines                       # exactly one JSON line on stdout
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 2 and doc["rccl_ranks"] == 2 and doc["dry_run"] is True and doc["steps"] == 2
    assert doc["scaling"] == "weak" and doc["higher_is_better"] is True


def test_single_rank_dry_run_line_shape():
    rc, lines, err = _run(["--dry-run", "--steps", "2", "--warmup", "1"])
    assert rc == 0, err[-2000:]
    doc = json.loads(lines[-1])
    assert doc["n_gpus"] == 1 and doc["rccl_ranks"] == 1
    for key in ("metric", "value", "unit", "steps", "warmup", "ms_per_step", "vs_baseline", "dtype", "data", "config"):
        assert key in doc


def test_launcher_mismatch_is_an_error():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu-baseline"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600, check=False)
    assert p.returncode == 2 and b"launcher mismatch" in p.stderr


def test_sharded_configs_4_and_5_two_ranks_dry_run():
    """`python bench_configs.py --gpus 2 --dry-run`: BASELINE configs 4 (multi-scale GeM-ResNet-101 + lw whitening) and 5 (augment then
    embed) sharded over two gloo ranks through the hub / wrapper / container API -- per-rank chunk, one all-gather of the descriptor block
    (mdir/components/data/wrapper.py:197-263,308-322; mdir/learning/network.py:664-677 for the chain).  Every rank checks that the gathered
    D x N matrix equals the single-process result over the same chunks bit for bit."""
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench_configs.py"), "--gpus", "2", "--dry-run"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=900, check=False)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    text = p.stdout.decode()
    doc = json.loads(text[text.index("{"):])
    assert doc["n_gpus"] == 2 and doc["backend"] == "gloo" and doc["dry_run"] is True
    for key in ("c3_gem_resnet101_ms_hub_default", "c3_gem_resnet101_ms_sms", "c4_augment_then_embed"):
        assert doc[key]["sharded_equals_single_process_bitwise"] is True, key
        assert doc[key]["gathered"] == [2048, 4], key


def test_self_launch_four_ranks_dry_run():
    """world size 4 through bench.py's own launcher (the driver's `--gpus 4` case): the emitted line reports the ranks torch.distributed
    actually formed"""
    rc, lines, err = _run(["--gpus", "4", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 4 and doc["rccl_ranks"] == 4 and doc["dry_run"] is True


def test_self_launch_eight_ranks_dry_run():
    """the driver's `--gpus 8` case (one rank per GPU of a node), rehearsed on gloo"""
    rc, lines, err = _run(["--gpus", "8", "--dry-run", "--steps", "2", "--warmup", "1"])
    assert rc == 0, err[-2000:]
    assert len(lines) == 1, lines
    doc = json.loads(lines[0])
    assert doc["n_gpus"] == 8 and doc["rccl_ranks"] == 8 and doc["dry_run"] is True


def _configs(args):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench_configs.py")] + args, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=1200, check=False)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    text = p.stdout.decode()
    return json.loads(text[text.index("{"):])


def test_sharded_configs_4_and_5_four_ranks_ragged_batch_dry_run():
    """world size 4 with a global batch the ranks do not divide (7 images: chunks 2, 2, 2, 1 -- the last chunk zero-padded for the
    collective and trimmed after it, SURVEY.md section 8e "Ragged N"): both sharded BASELINE configs, gathered D x 7 on every rank equal
    to the single-process result over the same chunks bit for bit; `rccl_ranks` is the world size torch.distributed formed."""
    doc = _configs(["--gpus", "4", "--dry-run", "--global-batch", "7"])
    assert doc["n_gpus"] == 4 and doc["rccl_ranks"] == 4 and doc["backend"] == "gloo"
    for key in ("c3_gem_resnet101_ms_hub_default", "c3_gem_resnet101_ms_sms", "c4_augment_then_embed"):
        assert doc[key]["sharded_equals_single_process_bitwise"] is True, key
        assert doc[key]["gathered"] == [2048, 7] and doc[key]["global_batch"] == 7, key


def test_sharded_configs_two_ranks_empty_last_chunk_dry_run():
    """3 ranks' worth of work on 2 ranks is ragged; 1 image on 2 ranks leaves the second rank's chunk EMPTY (it learns D from the
    all-reduce, contributes a zero-padded block and still takes part in the collective)"""
    doc = _configs(["--gpus", "2", "--dry-run", "--global-batch", "1"])
    assert doc["rccl_ranks"] == 2
    for key in ("c3_gem_resnet101_ms_hub_default", "c4_augment_then_embed"):
        assert doc[key]["sharded_equals_single_process_bitwise"] is True, key
        assert doc[key]["gathered"] == [2048, 1], key
