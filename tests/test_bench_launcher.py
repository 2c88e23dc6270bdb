"""bench.py --gpus N started the way the driver starts it (no external launcher): the parent spawns N fresh rank processes
(torch.distributed.run on 127.0.0.1) before touching any device, the ranks shard the batch through
ccsd_amd.distributed.load_sampling_fn_sharded, all-gather the samples, and rank 0's JSON line is the only thing on stdout.
Here the ranks run the product code over the host emulation of the kernels with gloo (`--emulate`, test only): the launcher,
sharding and reporting path is executed end to end on the GPU-less build box."""
import json
import os
import subprocess
import sys

from tests.emu_util import emu_library
from tests.helpers import ROOT


def test_bench_self_launches_two_ranks_on_the_emulation():
    emu_library()                                  # build once, before two ranks race for it
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--emulate", "--steps", "2", "--warmup", "1",
                        "--batch", "4"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 8 and line["config"]["finite"] is True
    assert line["value"] > 0 and abs(line["value"] - 8 / line["ms_per_step"]) < 1e-9
    for key in ("metric", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "data", "roofline", "cpu_baseline"):
        assert key in line
    # per-rank split of the timed region (diagnosis of a scaling curve): PC-step time without the all-gather, and the all-gather
    pr = line["per_rank"]
    assert [r_["rank"] for r_ in pr] == [0, 1]
    for r_ in pr:
        assert r_["ms_per_step"] > 0 and r_["all_gather_ms"] >= 0 and r_["region_ms"] <= line["ms_per_step"] * line["steps"] * 1.001


def test_bench_worker_failure_is_reported():
    """A rank that dies makes the parent exit non-zero (no JSON line)."""
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--emulate", "--steps", "1", "--warmup", "1",
                        "--batch", "3", "--workload", "qm9_CC", "--event-stride", "1"],
                       capture_output=True, text=True, timeout=900, env=dict(env, CCSD_BENCH_FAIL_RANK="1"), cwd=ROOT)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
