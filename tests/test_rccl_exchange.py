"""Hardware-gated: the rank exchange of the path over RCCL on REAL peer GPUs.  The reference concatenates
its workers' tile blobs in tile order (libxpng.c:764-769); with one process per GPU that is the one exchange step of the path
(xpng_amd/shard.py exchange_blobs_round_robin, backend "nccl" = RCCL over xGMI).  Until now only gloo on the CPU has run it
(tests/test_shard_gloo.py).  With fewer than two visible devices the test is skipped - which is every box this suite has seen."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_round_robin_exchange_over_rccl_on_two_real_devices(tmp_path):
    import torch
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need a HIP device")
    if torch.cuda.device_count() < 2:
        pytest.skip("one visible device: the RCCL exchange has still never run on real peers")
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    worker = os.path.join(ROOT, "tests", "_rccl_exchange_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), "2", str(port), str(tmp_path)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("an RCCL rank hung")
        outs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r}:\n{outs[r][-3000:]}"
        assert int(open(tmp_path / f"ok_{r}").read()) >= 2
