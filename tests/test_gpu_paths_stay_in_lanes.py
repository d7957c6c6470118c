"""The bounce kernel of the fast arithmetic keeps a path in its lane while the path needs nothing from the triangle
stage, and gives the lanes that fall free the next paths of the wavefront's stretch of the input (k_shade_fast).
Its images are judged by tolerance; its CONTROL FLOW is held to the bit here: RT_AMD_EXACT_STAYS=1 runs the same
kernel body with the exact arithmetic (k_shade_stays) and the parity tests then compare it with the oracle.

The library reads the knobs once per process, so the tests run in a child process.  RT_AMD_STRETCH_WAVES=1 gives
small frames the long stretches (lanes taking new paths beside paths that stay, entries of two list blocks in one
fetch) that only frames of millions of paths get otherwise.
"""
import os
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SMALL = ["test_gpu_random_scenes.py", "test_gpu_parity.py", "test_golden_fixtures.py", "test_gpu_large_mesh.py"]


def run_child(env_extra, args):
    env = dict(os.environ, RT_AMD_EXACT_STAYS="1", **env_extra)
    cmd = [sys.executable, "-u", "-m", "pytest", "-x", "-q", "-m", "gpu", "-p", "no:cacheprovider", "--timeout", "300",
           "--timeout-method", "thread"] + args
    r = subprocess.run(cmd, env=env, cwd=os.path.dirname(HERE), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "failed" not in r.stdout, r.stdout[-2000:]
    return r.stdout


@pytest.mark.gpu
def test_long_stretches_on_small_frames_match_the_oracle_bitwise():
    out = run_child({"RT_AMD_STRETCH_WAVES": "1"}, [os.path.join(HERE, f) for f in SMALL])
    print(out.strip().splitlines()[-1])


@pytest.mark.gpu
def test_default_stretches_match_the_oracle_bitwise():
    # small frames (stretches of one fetch) and the full-size ones, whose stretches are long
    out = run_child({}, [os.path.join(HERE, f) for f in SMALL] + [os.path.join(HERE, "test_gpu_configs.py"), "-k",
                    "not two_ranks and not rccl and not in_flight"])
    print(out.strip().splitlines()[-1])
