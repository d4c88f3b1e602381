set -e
timeout -k 10 120 tools/micro/spmm_lab.bin 128 4 48 0,10,8 8 | grep -E "nt=1"
timeout -k 10 120 tools/micro/spmm_lab.bin 128 1 48 0,10,8 8 | grep -E "nt=1" | sed "s/^/warm /"
timeout -k 10 120 tools/micro/spmm_lab.bin 256 4 48 0,10,8 8 | grep -E "nt=1" | sed "s/^/H256 /"
timeout -k 10 120 tools/micro/spmm_lab.bin 256 1 48 0,10,8 8 | grep -E "nt=1" | sed "s/^/H256 warm /"
