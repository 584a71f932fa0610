"""bench.py --gpus N run PLAINLY must start N ranks itself: the driver's SCALE run is
`python -m torch.distributed.run ... bench.py --gpus N`, but a bare `python bench.py --gpus 8` used to see WORLD_SIZE unset,
run as one rank and print an n_gpus: 1 line with rc 0.  The reference picks its own worker count too (libxpng.c:146-149).

CPU: the rank start-up alone (--spawn-check: gloo rendezvous, no GPU, nothing measured).  GPU: the real 2-rank line on the one
GPU of the box (gloo moves the blobs through the host; both ranks share device 0 - a rehearsal, not a scaling number)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return env


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert lines, stdout
    return json.loads(lines[-1])


def test_plain_command_with_gpus_2_starts_two_ranks():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--spawn-check"], env=_clean_env(),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = _json_line(out.stdout)
    assert line["n_gpus"] == 2 and line["spawn_check"] is True


def test_one_rank_needs_no_launcher():
    out = subprocess.run([sys.executable, BENCH, "--spawn-check"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and _json_line(out.stdout)["n_gpus"] == 1


@pytest.mark.parametrize("world", ["1", "3"])
def test_a_world_size_other_than_gpus_is_refused(world):
    env = _clean_env()
    env.update(WORLD_SIZE=world, RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    for extra in (["--spawn-check"], []):  # (the real path refuses in Env.__init__, before any GPU call)
        out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo"] + extra, env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode != 0
        assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")], out.stdout


@pytest.mark.gpu
def test_plain_bench_gpus_2_prints_a_two_rank_line():
    out = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--backend", "gloo", "--share", "1024", "--batch", "2", "--pipeline", "2",
                          "--steps", "2", "--warmup", "1", "--roofline-reps", "2", "--no-cpu", "--no-legs", "--no-config4"],
                         env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = _json_line(out.stdout)
    assert line["n_gpus"] == 2 and line["verified"]["roundtrip"] is True
    assert line["config"]["parallelism"].startswith("tile-range x2")
