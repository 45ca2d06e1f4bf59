cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call35; mkdir -p $O
for d in 0 1 2; do
if [ $d = 0 ]; then unset WITW_LIB; else export WITW_LIB=$GRAFT_REPO_ROOT/tools/debug/libwitw_ssdiag$d.so; fi
rocprofv3 --kernel-trace --stats -d $O/prof$d -o p -- python3 tools/debug/selfsync_bench.py > $O/log$d.txt 2>&1
echo "diag $d:"; python3 tools/debug/rocprof_db.py $O/prof$d selfsync | grep "us grid" | awk '{print $1}' | tr '\n' ' '; echo
done
