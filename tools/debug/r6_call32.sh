cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call32; mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/prof -o p -- python3 tools/debug/jpeg_huff_lockstep.py 2 > $O/log.txt 2>&1
grep -v amdgpu $O/log.txt | grep "decode_packed"
python3 tools/debug/rocprof_db.py $O/prof jpeg_huffman
