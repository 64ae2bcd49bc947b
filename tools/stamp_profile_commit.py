"""Build container only: record the git commit whose kernel sources match a PMC profile set.

    python tools/stamp_profile_commit.py profiles/r03_pmc_hbm_traffic.json

Walks back from HEAD to the most recent commit whose csrc/ + include/ digest equals the set's `_source.csrc_sha256` (the GPU box
has no .git, so the set is stamped with the digest there and with the commit here)."""
import hashlib
import json
import subprocess
import sys
from pathlib import Path

path = Path(sys.argv[1])
doc = json.loads(path.read_text())
want = doc["_source"]["csrc_sha256"]


def digest_at(commit):
    names = subprocess.run(["git", "ls-tree", "-r", "--name-only", commit, "torch-m3gnet_amd/csrc", "include"], capture_output=True, text=True,
                           check=True).stdout.split()
    csrc = sorted(n for n in names if n.startswith("torch-m3gnet_amd/csrc/") and (n.endswith(".h") or n.endswith(".hip")))
    inc = sorted(n for n in names if n.startswith("include/") and n.endswith(".h"))
    h = hashlib.sha256()
    for n in sorted(csrc, key=lambda n: Path(n).name) + sorted(inc, key=lambda n: Path(n).name):
        h.update(Path(n).name.encode())
        h.update(subprocess.run(["git", "show", f"{commit}:{n}"], capture_output=True, check=True).stdout)
    return h.hexdigest()[:16]


for commit in subprocess.run(["git", "rev-list", "-n", "200", "HEAD"], capture_output=True, text=True, check=True).stdout.split():
    if digest_at(commit) == want:
        doc["_source"]["git_commit"] = commit
        path.write_text(json.dumps(doc, indent=1))
        print("stamped", commit)
        break
else:
    print("no commit among the last 200 matches digest", want)
    sys.exit(1)
