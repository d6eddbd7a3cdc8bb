"""CPU: `python bench.py --gpus N` starts N ranks itself and never prints a line whose n_gpus differs from --gpus
(BASELINE.json configs[2]; the reference has no launcher: nmt_multimodal_beam_DE.py:277-282 is a commented-out
nn.DataParallel).  VAG_BENCH_LAUNCH_ONLY=1 stops every rank after the rendezvous (gloo), so no GPU is needed; the GPU form of
the same path (two ranks sharing the test box's card) is tests/test_gpu_round6.py::test_bench_self_launch_two_ranks_smoke."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env_extra, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra)
    return subprocess.run([sys.executable, BENCH] + args, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.timeout(400)
def test_plain_python_gpus2_launches_two_ranks():
    r = _run(["--gpus", "2", "--steps", "2", "--warmup", "1"], {"VAG_BENCH_LAUNCH_ONLY": "1"})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                      # ONE JSON line on stdout, whatever the ranks print
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks"] == [0, 1] and d["pids"] == 2 and d["steps"] == 2 and d["warmup"] == 1


@pytest.mark.timeout(120)
def test_fewer_devices_than_ranks_is_an_error_not_a_smaller_run():
    # this container has no GPU: device_count() is 0 < 2 -- the launcher must refuse, not measure one rank
    r = _run(["--gpus", "2", "--steps", "2"], {})
    assert r.returncode != 0
    assert not any(ln.strip().startswith("{") for ln in r.stdout.splitlines()), r.stdout
    assert "GPU" in r.stderr


@pytest.mark.timeout(120)
@pytest.mark.parametrize("gpus,world", [(8, 1), (2, 4), (1, 2)])
def test_world_size_mismatch_is_an_error(gpus, world):
    r = _run(["--gpus", str(gpus), "--steps", "2"], {"WORLD_SIZE": str(world), "RANK": "0", "LOCAL_RANK": "0",
                                                     "VAG_BENCH_LAUNCH_ONLY": "1"})
    assert r.returncode != 0
    assert r.stdout.strip() == "" and "WORLD_SIZE" in r.stderr
