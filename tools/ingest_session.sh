#!/bin/bash
# Runs ON the GPU box: the file-inclusive ingest rates DESIGN.md 3.7 quotes (page-cache resident synthetic 5 Mbp FASTA files; 16 distinct
# genomes, the others hard links). usage: tools/ingest_session.sh > profiles/rNN_ingest_rate.log
echo "# nproc $(nproc), cgroup cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
run() { echo "## $*"; env "$@" 2>&1 | grep -E "^wrote|^rep 1"; }
run python tools/ingest_rate.py 2048 5000000 plain
run python tools/ingest_rate.py 10000 5000000 gz 0 0 6 16
run python tools/ingest_rate.py 4096 5000000 gz 0 0 6 16
run python tools/ingest_rate.py 2048 5000000 gz 0 0 6 16
run python tools/ingest_rate.py 6144 5000000 bgzf 0 0 6 16
run GS_GZIP_DEVICE_ONLY=1 python tools/ingest_rate.py 6144 5000000 gz 0 0 6 16
run GS_INFLATE_WINDOW=global python tools/ingest_rate.py 6144 5000000 gz 0 0 6 16
run GS_GZIP_DEVICE=0 python tools/ingest_rate.py 4096 5000000 gz 0 0 6 16
run python tools/ingest_rate.py 4096 5000000 gz 0 0 1 16
run GS_GZIP_DEVICE=0 python tools/ingest_rate.py 4096 5000000 gz 0 0 1 16
echo "## device inflate alone (gs_gunzip_batch; kernel times from rocprofv3 --kernel-trace --stats)"
for cfg in "lds 1024 5000000" "global 1536 5000000" "pipe 1536 5000000" "global 4096 5000000" "pipe 4096 5000000" "pipe 4096 1000000"; do
  set -- $cfg
  GS_INFLATE_WINDOW=$1 bash tools/kstats.sh gpurun_out/_is.txt python tools/inflate_rate.py $2 $3 6 > /dev/null 2>&1
  echo "GS_INFLATE_WINDOW=$1  $2 members x $3 bp:"; head -3 gpurun_out/_is.txt | tail -2
done
