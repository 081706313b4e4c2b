"""Race detector for the multi-stream schedule of a training step (engine.py: weight-gradient stream, packing stream,
bias-sum reductions; loss.py: speech-side prefetch stream; under data parallelism RCCL's streams on top).

A missing event between two streams does not fail loudly: it produces results that are wrong by a little, sometimes
(round 2: commit 2ef6d75, found by luck).  The schedule is therefore checked against ITSELF on one stream: same kernels,
same summation orders, so every output of two consecutive training steps must be bitwise equal — in a fresh process, where
every buffer is first-use, in fp32 and bf16, single-process and through the data-parallel path at world size 1.
The work is done by tests/stream_race_worker.py (one process per compute dtype)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_multi_stream_step_is_bitwise_equal_to_single_stream_step(dtype):
    with __import__("socket").socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, SDA_DP_SINGLE_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    for k in [k for k in env if k.startswith("SDA_ENGINE_")]:
        del env[k]
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "stream_race_worker.py"), dtype], env=env,
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    log_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(log_dir):
        open(os.path.join(log_dir, f"stream_race_{dtype}.log"), "w").write(out.stdout + "\n---- stderr ----\n" + out.stderr)
    assert out.returncode == 0 and "stream race check ok" in out.stdout, (out.stdout[-3000:] + "\n" + out.stderr[-3000:])
