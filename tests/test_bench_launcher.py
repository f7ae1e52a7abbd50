"""CPU: bench.py's multi-rank path.  `python bench.py --gpus 2` must start its two ranks itself (the driver invokes it
that way) and, started under torch.distributed.run, use the ranks it is given; both run the same launcher / barrier /
MAX-over-ranks / one-JSON-line code the GPU benchmark runs, with the step stubbed out (--stub-cpu, gloo backend)."""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def _line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--repeats", "2", "--stub-cpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 2 and j["steps"] == 3 and j["warmup"] == 1 and j["valid"] is False
    assert j["scaling"] == "weak" and j["config"]["parallelism"] == "replicas x2" and len(j["repeats"]["ms_per_step_each"]) == 2
    # 2 ranks x 3 steps of >= 2 ms each
    assert j["ms_per_step"] >= 2.0 and abs(j["value"] - 2 * 3 / (j["ms_per_step"] * 3 / 1e3)) < 1e-2 * j["value"]


def test_bench_under_torch_distributed_run():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "0",
           "--repeats", "1", "--stub-cpu"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert _line(r.stdout)["n_gpus"] == 2


def test_single_rank_stub():
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "0", "--repeats", "3",
                        "--stub-cpu"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    j = _line(r.stdout)
    assert j["n_gpus"] == 1 and "roofline" not in j and "cpu_baseline" not in j
