"""bench.py's multi-process path, for real, on ONE GPU: `python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2` with both
ranks on cuda:0 and torch.distributed's gloo backend (collectives staged through host memory) -- everything of an N-GPU run
except RCCL itself: rendezvous, row-block generation, ghost-row plan over all_gather_object, halo exchange and all-reduce hooks
called from the C library in separate processes, the barrier + max-over-ranks timing, ONE JSON line from rank 0."""
import json
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(nranks, rows, port):
    cmd = [sys.executable]
    if nranks > 1:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nranks), "--master-addr", "127.0.0.1", "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py"), "--gpus", str(nranks), "--rows-per-gpu", str(rows), "--no-cpu", "--steps", "6", "--warmup", "4", "--spmm-reps", "2"]
    if nranks > 1:
        cmd += ["--backend", "gloo", "--one-device"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]  # exactly one JSON line on stdout, from rank 0
    first = [float(m.group(1)) for m in re.finditer(r"Lanczos estimates ([0-9.e+-]+) ->", p.stderr)]
    return json.loads(lines[0]), first, p.stderr


def test_two_ranks_share_one_gpu():
    line2, first2, err2 = _bench(2, 60000, 29631)
    assert line2["n_gpus"] == 2 and line2["steps"] == 6 and line2["value"] > 0 and line2["scaling"] == "weak"
    assert "global 120000" in line2["config"]["workload"]
    assert len(first2) == 2 and first2[0] == first2[1]  # both ranks hold the same replicated small quantities
    assert err2.count("ghosts=") == 2 and "ghosts=0" not in err2  # rows really crossed the partition
    # ... and the products overlapped their exchange: interior rows on the second stream while the ghost rows travelled
    overl = [int(m.group(1)) for m in re.finditer(r'"spmm_halo_overlapped": (\d+)', err2)]
    assert len(overl) >= 2 and min(overl) > 0, overl
    # (numerical equivalence with the undivided problem is what tests/test_gpu_partition.py checks: bench.py generates every
    # rank's row block from its own seed, so there is no single-process twin of this run)
    last = [float(m.group(1)) for m in re.finditer(r"Lanczos estimates [0-9.e+-]+ -> ([0-9.e+-]+);", err2)]
    assert len(last) == 2 and last[0] == last[1] and last[0] < 1e-3 * first2[0]  # and the iteration converges


def test_bench_line_contract():
    """the one JSON line of `python bench.py` (N = 1, small sizes): every field the bench contract names, with sane values"""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--rows-per-gpu", "60000", "--steps", "5", "--warmup", "3", "--spmm-reps", "2", "--cpu-trips", "2", "--cpu-warmup", "2", "--direct-steps", "3"]
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 3 and d["higher_is_better"] is True and d["scaling"] == "weak"
    assert d["vs_baseline"] is None and d["dtype"] == "f64" and d["data"] == "synthetic" and d["unit"] == "iterations/s"
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    assert "workload" in d["config"] and "model" not in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert "traffic" in r and r["achieved"] > 0
    # the other kernels of a trip, each with its own live timing: bytes (or flops), average duration, fraction of its roofline
    names = [q["kernel"] for q in d["roofline_kernels"]]
    for want in ("k_update_gram", "k_panel_gemm_wide", "k_lanczos_pass", "k_gram_cols", "k_panel_gemm"):
        assert want in names, names
    for q in d["roofline_kernels"]:
        assert q["avg_ms"] > 0 and 0 < q["frac"] < 1.5 and q["bound"] in ("hbm", "mfma") and ("algorithmic_bytes" in q or "flops" in q)
    cfg = d["config"]
    for key in ("steady_it_s", "restart_trips", "host_ms", "device_critical_ms", "sections_ms_per_trip", "median_trip_ms", "gpu_busy_frac", "direct_backend_it_s"):
        assert key in cfg, key
    assert cfg["host_ms"] > 0 and cfg["device_critical_ms"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == "iterations/s" and "sample" in c
