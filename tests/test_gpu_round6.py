"""GPU: round-6 additions -- the launcher of bench.py with real ranks on the test box's card, per-tensor relative-L2 gradient
parity, the wider persistent recurrence kernels against the launch chains they replace and against the oracle."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT, PKG

pytestmark = pytest.mark.gpu


@pytest.mark.timeout(900)
def test_bench_self_launch_two_ranks_smoke():
    """`python bench.py --gpus 2` (no launcher around it) on a one-GPU box: VAG_DP_SMOKE=1 puts both ranks on cuda:0 over gloo with the
    launch-chain recurrences (two processes cannot both keep a persistent grid resident).  The line must say n_gpus 2 and carry the
    data-parallel block; SURVEY 8e, BASELINE.json configs[2]."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env["VAG_DP_SMOKE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "2",
                        "--no-operators", "--no-cpu-baseline", "--no-extras", "--single-window"],
                       env=env, capture_output=True, text=True, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3
    assert d["dp"]["backend"] == "gloo" and len(d["dp"]["per_rank_ms_per_step"]) == 2
    assert np.isfinite(d["final_loss"]) and d["value"] > 0
