"""BASELINE config 4 rehearsed on the one GPU of the box: two fresh child ranks share cuda:0 (gloo carries the
exchange; RCCL needs one GPU per rank) and run bench.py's step with the HIP path -- shard, scan, gather_records,
restore_hits -- against the unsharded scan; then bench.py itself, started WITHOUT a launcher, must start its two ranks
and print a line whose n_gpus is the number of ranks that ran, with the hit-buffer gather inside the timed steps."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _clean_env():
    env = {k: v for k, v in os.environ.items() if not k.startswith("KG_") and k not in
           ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return env


def test_two_ranks_share_the_gpu_hip_path_vs_unsharded(tmp_path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "two_rank_worker.py"), str(tmp_path), "30000000", "20000003"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=_clean_env(), timeout=900)
    assert r.returncode == 0 and (tmp_path / "ok").exists(), r.stdout.decode()[-2000:] + r.stderr.decode()[-4000:]


@pytest.mark.parametrize("gather,overlap", [(True, True), (True, False), (False, True)])
def test_bench_self_launch_two_ranks(gather, overlap):
    env = _clean_env()
    env["KG_BENCH_DEVICE"] = "0"                       # both ranks on the one GPU
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--total-bp", "40000000",
           "--num-sigs", "20000003", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + ([] if gather else ["--no-gather-hits"]) + ([] if overlap else ["--no-overlap-exchange"])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-4000:]
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    cfg = line["config"]
    assert cfg["total_bp_all_ranks"] == 40000000 and cfg["hits_all_ranks"] > 0
    assert ("CALL/OTU/hit" in cfg["exchange"]) == gather
    assert cfg["hits_gathered_rank0"] == (cfg["hits_all_ranks"] if gather else None)      # every rank's hits reached rank 0
    assert (cfg["hits_gather_probe"] is None) == gather
    assert 0 < line["roofline"]["frac_step"] <= line["roofline"]["frac"] * 1.05
