"""The C++ client of the drop-in boundary (bench/bench_abi.cpp: include/piccolo_hip.h + the HIP runtime, no Python, no torch)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


@pytest.mark.gpu
def test_cpp_client_runs_the_path_through_the_c_abi():
    """pcl_create -> structure (1-based, as a Julia host asks) -> pcl_eval_jac_dev in a timed loop -> pcl_eval_jac on host buffers;
    the two delivery paths must agree bit for bit and the client exits 0."""
    import __graft_entry__ as g

    exe = g.build_abi_client()
    r = subprocess.run([exe, os.path.join(ROOT, "bench", "config3_inputs.bin"), "20", "3"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["device_and_host_paths_bitwise_equal"] is True
    assert line["value"] > 0 and line["last_kernel"] > 0 and 0 < line["max_abs_delta"] < 1.0


@pytest.mark.gpu
def test_cpp_client_drives_the_collective_through_the_c_abi():
    """pcl_comm_get_unique_id -> pcl_comm_init(nranks = 1) -> pcl_reduce_sum_dev behind every evaluation -> pcl_comm_destroy from the C++ client (no
    Python, no torch): what one GPU can execute of the non-Python multi-rank path.  `--ranks N` with N > 1 is the same binary, one process per device,
    the id through a file -- for the first multi-GPU node (two RCCL ranks on ONE device are refused by the library).  More ranks than devices: refused."""
    import __graft_entry__ as g

    exe = g.build_abi_client()
    inputs = os.path.join(ROOT, "bench", "config3_inputs.bin")
    r = subprocess.run([exe, inputs, "20", "3", "--ranks", "1"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [json.loads(ln) for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 2 and lines[0]["device_and_host_paths_bitwise_equal"] is True
    c = lines[1]
    assert c["rccl_ranks"] == 1 and c["sum_exact_on_rank0"] is True and c["other_ranks_ok"] is True and c["payload_doubles"] == 1 + 99 * 7
    assert 0 < c["us_per_reduce_alone"] < c["us_per_eval_plus_reduce"] < 1000
    import torch

    r = subprocess.run([exe, inputs, "5", "1", "--ranks", str(torch.cuda.device_count() + 1)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "device(s) visible" in r.stderr
