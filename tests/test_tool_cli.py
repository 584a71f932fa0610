"""CPU: the `tool` command (flips / rotations of a .7, reference Mirroring_and_Rotating/tool.c) against numpy, and the
reference's own reversibility checks (reversibility.rb)."""
import os
import subprocess

import numpy as np
import pytest

from xpng_amd import api
from xpng_amd.synth import load_seven, synth_raster, to_seven_bytes

TOOL = os.path.join(os.path.dirname(api.CLI), "tool")
NP = {"mv": lambda r: r[::-1], "mh": lambda r: r[:, ::-1], "mvh": lambda r: r[::-1, ::-1],
      "r90": lambda r: np.rot90(r, k=-1), "r270": lambda r: np.rot90(r, k=1), "tl": lambda r: r, "tr": lambda r: r}


@pytest.fixture(scope="module", autouse=True)
def built():
    api.build_native(("hip", "host"))


def run(op, src, dst):
    return subprocess.run([TOOL, "--" + op, str(src), str(dst)]).returncode


@pytest.mark.parametrize("alpha", [True, False])
def test_every_operation_matches_numpy(tmp_path, alpha):
    r = synth_raster("photo", 53, 31, alpha)
    a, b = tmp_path / "a.7", tmp_path / "b.7"
    a.write_bytes(to_seven_bytes(r))
    for op, f in NP.items():
        assert run(op, a, b) == 0
        assert np.array_equal(load_seven(str(b)), np.ascontiguousarray(f(r))), op


def test_reversibility_like_the_reference_script(tmp_path):
    r = synth_raster("noise", 40, 27, True)
    s, x, back = tmp_path / "s.7", tmp_path / "x.7", tmp_path / "r.7"
    s.write_bytes(to_seven_bytes(r))
    assert run("r90", s, x) == 0 and run("r270", x, back) == 0 and back.read_bytes() == s.read_bytes()
    for op in ("mv", "mh", "mvh", "tl", "tr"):
        assert run(op, s, x) == 0 and run(op, x, back) == 0 and back.read_bytes() == s.read_bytes(), op


def test_usage():
    p = subprocess.run([TOOL, "--nope", "a", "b"], capture_output=True, text=True)
    assert p.returncode == 1 and "r90|r270|mv|mh|mvh|tl|tr" in p.stdout
