cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call27; mkdir -p $O
python3 tools/debug/overlap_prep.py semantic 2>&1 | grep -v amdgpu | tee $O/overlap.txt
python3 tools/debug/overlap_prep.py fov 2>&1 | grep -v amdgpu | tee -a $O/overlap.txt
