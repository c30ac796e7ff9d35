#!/bin/bash
# Runs ON the GPU box: file-inclusive gzip ingest against the device group size (GS_GZIP_GROUP members per inflate launch) and the k_inflate form.
# usage: tools/ingest_groups.sh [n_files] [groups...]
N=${1:-10000}; shift
echo "# nproc $(nproc), cgroup cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
run() { echo "## $*"; env "$@" 2>&1 | grep -E "^rep 1"; }
for g in ${@:-1536 3072 5376}; do
  run GS_GZIP_GROUP=$g python tools/ingest_rate.py $N 5000000 gz 0 0 6 16
  run GS_GZIP_GROUP=$g GS_GZIP_DEVICE_ONLY=1 python tools/ingest_rate.py $N 5000000 gz 0 0 6 16
done
