"""The kernel variants that are selectable at run time for A/B measurements must
stay correct: the emulated forward/backward parity tests are re-run in a child process with the
selector set (the library reads it once per process).
  LWM_FWD_PP=1        ping-pong schedule of the forward
  LWM_DKDV_WAVES=4    4-wave / 64-keys-per-wave dK/dV
  LWM_DKDV_WAVES=84   8-wave dK/dV with the late half delayed"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("env,select", [
    ({"LWM_FWD_PP": "1"}, "fwd or forward or packed or future"),
    ({"LWM_DKDV_WAVES": "4"}, "bwd or backward or packed"),
    ({"LWM_DKDV_WAVES": "84"}, "bwd or backward"),
])
def test_variant_passes_the_emulated_parity_tests(env, select):
    e = dict(os.environ, **env)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_emu_kernels.py", "-q", "-x", "-k", select,
                        "-p", "no:cacheprovider"], cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " passed" in r.stdout
