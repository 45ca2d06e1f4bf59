cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for t in 512 1024 256; do WITW_SELFSYNC_THREADS=$t WITW_LIB=$GRAFT_REPO_ROOT/tools/debug/libwitw_hip_ssdiag.so python3 tools/debug/selfsync_rounds.py 2>&1 | grep -v amdgpu; done
