"""bench.py contract on the GPU box: the single-GPU line, and the multi-rank control flow (one process per rank under
torch.distributed.run, barrier-bracketed timing, max over ranks, whole-job aggregate) rehearsed with two ranks on ONE
device over gloo - RCCL cannot place two ranks on one GPU, so only its transport is left to the 8-GPU node."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

REQUIRED = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")


def _last_json(text):
    return json.loads([l for l in text.splitlines() if l.startswith("{")][-1])


def test_single_gpu_line_has_the_contract_fields():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "cfg1", "--steps", "5", "--warmup", "2",
                          "--no-alt-engine"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    d = _last_json(out.stdout)
    for k in REQUIRED:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 5 and d["warmup"] == 2 and d["value"] > 0
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["one_thread"]["value"] > 0
    assert d["value"] > 10 * c["value"]                          # north_star: >= 10 x the CPU path
    # config 1's oracle comparison: all 500 samples of the last timed step, scored by the C restatement
    p = d["parity"]
    assert p["pass"] and p["reproduces_timed_step"] and p["samples"] == 500 and p["of"] == 500
    assert p["max_abs_dE_per_site"] < 1e-4 and d["exit_code"] == 0


def test_two_rank_control_flow_over_gloo_on_one_device():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
                          "--gpus", "2", "--workload", "cfg1", "--steps", "5", "--warmup", "2", "--transport", "gloo",
                          "--same-device"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                       # rank 0 alone prints
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert d["config"]["global_numsamples"] == 2 * d["config"]["numsamples_per_gpu"]
    assert abs(d["value"] - 2 * d["config"]["numsamples_per_gpu"] * d["config"]["sites"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
