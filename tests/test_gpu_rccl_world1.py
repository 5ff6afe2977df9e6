"""The first RCCL execution of the exchange path, without a second GPU: ONE rank under torch.distributed.run with
--backend nccl drives bench.py's step + exchange_start / finish (bench.py --exchange-at-world-1).  This proves, on real RCCL,
init_process_group("nccl", device_id=...), the load order of torch's libamdhip64 and the library's (kmergutsjava_amd/_native.py),
the device-tensor size gather (a real RCCL kernel even at world size 1), rank 0's pass-through of its own zero-copy views and
kg_restore_hits_device on torch's stream followed by the release of the result.  What it cannot reach: batch_isend_irecv
between two GPUs (RCCL refuses two ranks on one device); that part is rehearsed with gloo in tests/test_gpu_two_ranks.py."""
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


@pytest.mark.parametrize("overlap", [True, False])
def test_one_rank_through_rccl(overlap):
    env = {k: v for k, v in os.environ.items() if not k.startswith("KG_") and k not in
           ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--backend", "nccl",
           "--exchange-at-world-1", "--total-bp", "60000000", "--num-sigs", "200000033", "--steps", "3", "--warmup", "1",
           "--no-cpu-baseline", "--no-extra-configs"] + ([] if overlap else ["--no-overlap-exchange"])
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, timeout=900)
    assert r.returncode == 0, r.stdout.decode()[-2000:] + r.stderr.decode()[-6000:]
    line = json.loads([l for l in r.stdout.decode().splitlines() if l.startswith("{")][-1])
    cfg = line["config"]
    assert line["n_gpus"] == 1 and line["scaling"] is None and line["value"] > 0
    assert "RCCL" in cfg["exchange"] and "CALL/OTU/hit" in cfg["exchange"]
    assert cfg["hits_all_ranks"] > 100_000 and cfg["hits_gathered_rank0"] == cfg["hits_all_ranks"]
    assert cfg["rank0_restore_ms"] is not None and cfg["rank0_restore_ms"] > 0
