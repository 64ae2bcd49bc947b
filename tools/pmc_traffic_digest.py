"""Digest of the kernel sources of this tree (csrc/*.hip, csrc/*.h, include/*.h): ties a PMC profile set to the build it was
collected on -- the same digest bench.py computes at run time."""
import hashlib
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def csrc_digest():
    h = hashlib.sha256()
    files = sorted((ROOT / "torch-m3gnet_amd" / "csrc").glob("*.h*")) + sorted((ROOT / "include").glob("*.h"))
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]
