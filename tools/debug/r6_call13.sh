cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call13; mkdir -p $O
WITW_BF_S16=1 bash tools/ab_lib.sh "--precision bf16 --model semantic" dma=tools/bin/lib_s16_dma.so dma_sp5=tools/bin/lib_s16_dma_sp5.so sp5=tools/bin/lib_s16_sp5.so > $O/ab_infer.txt 2>&1
cat $O/ab_infer.txt
