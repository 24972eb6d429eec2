"""bench.py --gpus N: the job's alignments are sharded over the ranks by size (strong scaling, BASELINE config 3)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lpt_parts_cover_the_job_once_and_balance():
    import bench
    from make_prg_amd.utils.synthetic import config_shape
    seeds = list(range(3000))
    for n in (1, 2, 4, 8):
        parts = bench.lpt_parts(seeds, n)
        assert sorted(s for p in parts for s in p) == seeds
        loads = [sum(config_shape("C", s)[0] * config_shape("C", s)[1] for s in p) for p in parts]
        assert max(loads) <= 1.02 * (sum(loads) / n)
    assert bench.lpt_parts(seeds, 4) == bench.lpt_parts(seeds, 4)          # deterministic


@pytest.mark.gpu
def test_two_ranks_share_one_job_strong_scaling():
    """Two ranks on ONE device (gloo + MPRG_DEVICE_MODULO: RCCL refuses two ranks on one GPU): every rank builds and verifies
    its own shard, the line reports the whole job once."""
    env = dict(os.environ, MPRG_DIST_BACKEND="gloo", MPRG_DEVICE_MODULO="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--batch", "600", "--steps", "2", "--warmup", "1",
           "--workers", "1", "--no-cpu-baseline", "--no-end-to-end", "--no-cli-leg"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["n_gpus"] == 2
    assert line["config"]["alignments_per_step"] == 600 and 250 <= line["config"]["alignments_rank0"] <= 350
    assert line["config"]["verified"]["mismatches"] == 0 and line["config"]["verified"]["mismatches_all_ranks"] == 0
    assert abs(line["value"] - 600 * 2 / (line["ms_per_step"] * 2 / 1000.0)) < 1.0


@pytest.mark.gpu
def test_one_rank_under_rccl():
    """`bench.py --gpus 1` under the launcher with the `nccl` (= RCCL) process group forced for ONE rank (MPRG_DIST_FORCE=1): the barrier and
    the all-reduces of the multi-GPU line run on device tensors through RCCL in the process that also loads the kernels' library."""
    env = dict(os.environ, MPRG_DIST_FORCE="1")
    env.pop("MPRG_DIST_BACKEND", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29573", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--batch", "600", "--steps", "2", "--warmup", "1",
           "--workers", "1", "--no-cpu-baseline", "--no-end-to-end", "--no-cli-leg", "--no-single-worker-leg", "--no-shard-projection", "--no-deep-leg"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["process_group"].startswith("nccl")
    assert line["config"]["verified"]["mismatches"] == 0 and line["config"]["verified"]["mismatches_all_ranks"] == 0
