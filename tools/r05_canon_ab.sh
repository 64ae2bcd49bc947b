# same-box A/B of the six-launch canonical topology build (M3G_CANON_FAST=0: the general build under the canonical flag)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for f in 1 0; do
    for m in refill rebuild; do echo "fast=$f $(M3G_CANON_FAST=$f python tools/profile_md_iteration.py $m 60 2>/dev/null | tail -1)"; done
  done
done
