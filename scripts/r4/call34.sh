cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for m in 10240 1024 512; do timeout 300 python3 scripts/r4/emit_phase_mean.py $m 2>&1 | grep -v amdgpu.ids; done
