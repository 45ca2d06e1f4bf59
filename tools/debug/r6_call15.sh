cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for m in fp32train bf16train semtrain seminfer baseline; do python3 tools/debug/pair_after_train.py $m 2>&1 | grep -v amdgpu; done
