"""The dev aids and evidence scripts are committed code that only ever runs on a GPU box: at least they parse (no GPU needed)."""
import glob
import os
import py_compile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = sorted(glob.glob(os.path.join(ROOT, "tests", "tools", "*.py")) + glob.glob(os.path.join(ROOT, "scripts", "*.py")) +
               glob.glob(os.path.join(ROOT, "scripts", "experiments", "*.py")) + [os.path.join(ROOT, "bench.py"), os.path.join(ROOT, "__graft_entry__.py")])


@pytest.mark.parametrize("path", FILES, ids=[os.path.relpath(f, ROOT) for f in FILES])
def test_script_parses(path, tmp_path):
    py_compile.compile(path, cfile=str(tmp_path / "x.pyc"), doraise=True)


def test_round_scripts_parse():
    import subprocess
    for sh in sorted(glob.glob(os.path.join(ROOT, "scripts", "rounds", "*.sh")) + glob.glob(os.path.join(ROOT, "scripts", "*.sh"))):
        assert subprocess.run(["bash", "-n", sh]).returncode == 0, sh
