"""The committed golden fixtures are exactly what the committed generator produces from the reference at HEAD.

tests/golden/generate_golden.py imports the reference's own nn code (build container only: /root/reference does not exist on
the GPU box, so the test is skipped there) and is re-run here into a temporary directory; every array of every case and model
file must be bit-identical to the committed .npz (array contents are compared, not zip bytes: the archive stores timestamps)."""
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLDEN = ROOT / "tests" / "golden"


@pytest.mark.skipif(not Path("/root/reference/src/torch_m3gnet").is_dir(), reason="the reference lives in the build container only")
def test_generator_reproduces_the_committed_fixtures(tmp_path):
    proc = subprocess.run([sys.executable, str(GOLDEN / "generate_golden.py"), "--out", str(tmp_path)], capture_output=True, text=True,
                          timeout=600)
    assert proc.returncode == 0, proc.stderr[-2000:]
    committed = sorted(p.name for p in GOLDEN.glob("*.npz"))
    assert sorted(p.name for p in tmp_path.glob("*.npz")) == committed
    for name in committed:
        new, old = np.load(tmp_path / name), np.load(GOLDEN / name)
        assert sorted(new.files) == sorted(old.files), name
        for key in old.files:
            assert new[key].dtype == old[key].dtype and new[key].shape == old[key].shape, (name, key)
            assert new[key].tobytes() == old[key].tobytes(), (name, key)
