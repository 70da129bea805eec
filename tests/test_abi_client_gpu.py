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
