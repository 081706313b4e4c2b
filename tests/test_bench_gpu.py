"""bench.py through the path the driver's scaling run takes: `python bench.py --gpus N` from a bare interpreter spawns its own
ranks (`python -m torch.distributed.run --nproc-per-node N ... bench.py`), every rank runs the data-parallel step and rank 0
prints ONE JSON line.  On a one-GPU box the N = 2 case is rehearsed with both ranks on cuda:0 and gloo as the backend
(bench.py's SDA_FORCE_DEVICE / SDA_DIST_BACKEND overrides): the spawn, the rendezvous, the prefetch-ahead of the next batch's
speech rows, the collectives' call sequence and the JSON contract are the real ones, only the transport is not RCCL."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _json_lines(text):
    out = []
    for line in text.splitlines():
        line = line.strip()
        if line.startswith("{") and line.endswith("}"):
            try:
                out.append(json.loads(line))
            except ValueError:
                pass
    return out


def test_bench_spawns_two_ranks_and_prints_one_json_line():
    env = dict(os.environ, SDA_FORCE_DEVICE="0", SDA_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
           "--batch", "32", "--timer-steps", "1", "--coll-timer-steps", "2"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    log_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(log_dir):
        open(os.path.join(log_dir, "bench_two_ranks.log"), "w").write(out.stdout + "\n---- stderr ----\n" + out.stderr)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = _json_lines(out.stdout)
    assert len(lines) == 1, out.stdout[-2000:]
    line = lines[0]
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 64 and line["config"]["parallelism"] == "dp2"
    assert line["value"] > 0 and line["unit"] == "segments/s" and line["higher_is_better"] is True
    assert line["value"] == pytest.approx(64 * 3 / (line["ms_per_step"] * 3e-3), rel=1e-3)
    assert "roofline" in line and "cpu_baseline" not in line          # the CPU leg is rank 0's at N = 1 only
    assert 0.0 <= line["top10_acc"] <= 1.0 and line["final_loss"] == line["final_loss"]
    assert "held-out" in line["top10_note"]
    # the collective brackets of tools/scale_run.sh: 20 SyncBN statistics all-reduces per step (10 forward + 10 backward), the
    # speech-row all-gather, the loss's row-statistics all-gather, the gradient buckets
    coll = line["collectives_us_per_step"]
    assert sum(v["calls_per_step"] for k, v in coll.items() if k.startswith("all_reduce 2560 B")) >= 20
    assert any(k.startswith("all_gather_into_tensor") for k in coll) and all(v["us_per_step"] >= 0 for v in coll.values())


def test_bench_single_gpu_line_carries_the_contract_fields():
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--batch", "32",
           "--timer-steps", "1", "--no-host-sync-leg"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    (line,) = _json_lines(out.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["dtype"] == "bf16" and line["vs_baseline"] is None
    assert line["host_enqueue_ms_per_step"] > 0          # the host's share of the timed region (it runs ahead when < ms_per_step)
    r = line["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["frac"] == pytest.approx(r["achieved"] / r["peak"], rel=1e-3)


def test_bench_emulated_world_config_and_feed_legs():
    """Round-5 modes: `--emulate-world` (ONE JSON line although RCCL prints a banner on stdout; global batch, workload text and
    collective brackets of the emulated world), `--config 4` (per-GPU shape of configs[3]) and the `with_feed` leg."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "SDA_DP_SINGLE_RANK"):
        env.pop(k, None)
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--timer-steps", "1", "--no-host-sync-leg"]
    out = subprocess.run(base + ["--batch", "32", "--emulate-world", "4", "--coll-timer-steps", "1"], env=env, capture_output=True,
                         text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    assert out.stdout.count("\n") == 1, out.stdout[-1000:]            # exactly one line on stdout: the JSON
    (line,) = _json_lines(out.stdout)
    assert line["n_gpus"] == 1 and line["config"]["global_batch"] == 128 and "emulating dp4" in line["config"]["parallelism"]
    assert "PER-RANK step of a 4-rank job" in line["config"]["workload"] and "cpu_baseline" not in line and "with_feed" not in line
    coll = line["collectives_us_per_step"]
    assert sum(v["calls_per_step"] for k, v in coll.items() if k.startswith("all_reduce 2560 B")) >= 20
    assert line["roofline"]["traffic"] is None
    out = subprocess.run(base + ["--config", "4", "--batch", "32"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    (line,) = _json_lines(out.stdout)
    assert "60 ch x 360" in line["config"]["workload"] and "configs[3]" in line["config"]["workload"] and "cpu_baseline" not in line
    out = subprocess.run(base + ["--batch", "32", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    (line,) = _json_lines(out.stdout)
    assert line["with_feed"]["value"] > 0 and line["with_feed"]["host_enqueue_ms_per_step"] > 0
