cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
HBS_KERNEL=5 timeout 900 python scripts/nal_sweep.py --gib 2 --sizes 64,128,256,384,448 2>&1 | grep mean_nal | cut -c1-330
