cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_call8; mkdir -p $O
D=$(mktemp -d /tmp/witw_rst_XXXX)
timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers 4 --e2e-pairs 8192 --jpeg-restart-rows 1 --e2e-dir $D --detail-out $O/e2e_rst_w4.json > $O/e2e_rst_w4.line 2> $O/e2e_rst_w4.err
echo "rst w4 rc=$?"; cut -c1-900 $O/e2e_rst_w4.line
timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers 8 --e2e-pairs 8192 --jpeg-restart-rows 1 --e2e-dir $D --detail-out $O/e2e_rst_w8.json > $O/e2e_rst_w8.line 2> $O/e2e_rst_w8.err
echo "rst w8 rc=$?"; cut -c1-600 $O/e2e_rst_w8.line
WITW_JPEG_DEVICE_ENTROPY=0 timeout -k 10 400 python3 bench.py --mode e2e --precision bf16 --workers 4 --e2e-pairs 4096 --jpeg-restart-rows 1 --e2e-dir $D --no-decode-scaling --detail-out $O/e2e_host_w4.json > $O/e2e_host_w4.line 2> $O/e2e_host_w4.err
echo "host-entropy w4 rc=$?"; cut -c1-600 $O/e2e_host_w4.line
rm -rf $D
tail -3 $O/e2e_rst_w4.err
