"""Host-only check of the lattice decomposition behind the one-launch MGM kernel (s2p_amd/csrc/mgm_geom.hpp, shared
by host and device code): tools/probes/mgm_geom_check.cpp is compiled with g++ and verifies, for a set of image
sizes and for 4, 8 and 16 directions (4 / 12 / 52 lattices: a knight's move splits into five residue classes), that the
lattices cover every (direction, pixel) pair exactly once, that the lattice predecessors (u - 1, v), (u, v - 1) and
(u - 1, v - 1) are exactly the MGM predecessors p - r, p - r_perp and p - r - r_perp, and that mgm_row_interval returns
exactly the in-image points of every lattice row."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="needs g++")
def test_lattices_cover_and_predecessors(tmp_path):
    exe = str(tmp_path / "mgm_geom_check")
    subprocess.run(["g++", "-O2", "-o", exe, os.path.join(ROOT, "tools", "probes", "mgm_geom_check.cpp")], check=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout
    assert "FAIL" not in r.stdout and r.stdout.count(": ok") >= 51
