# usage (on the GPU box, from the repo root): tools/build_trace_scale.sh <n> [tag]   - rocprofv3 kernel trace of tools/build_scale.py --n <n>, per-kernel sums by decile of the build
N=${1:-600000}; T=${2:-r05}
R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt -o kt -- python $R/tools/build_scale.py --n $N --nq 64 --reps 1 > $R/gpurun_out/${T}_build_trace_${N}.log 2>&1
cd $R
python - <<P >> gpurun_out/${T}_build_trace_${N}.log
import csv, glob, collections
fn = glob.glob('gpurun_out/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
t0 = min(int(r['Start_Timestamp']) for r in rows); t1 = max(int(r['End_Timestamp']) for r in rows)
def key(n):
    for k, s in (('join', 'k_match_join'), ('prepass', 'k_hnsw_search_dense'), ('plan', 'k_hnsw_plan'), ('merge', 'k_link_merge'), ('link', 'k_link_'), ('sparse_fill', 'k_sparse_fill'), ('cache_rows', 'k_cache_rows'),
                 ('r2c', 'k_rows_to_cols'), ('tile', 'k_hamming_qxc'), ('sketch', 'k_sketch_min'), ('synth', 'k_synth'), ('fill', 'fillBuffer'), ('copy', 'copyBuffer'), ('heavy', 'k_heavy'), ('label', 'k_label'), ('qcols', 'k_query_cols')):
        if s in n: return k
    return 'other'
D = 10
acc = collections.defaultdict(lambda: [0.0] * D); cnt = collections.Counter()
for r in rows:
    k = key(r['Kernel_Name']); s = int(r['Start_Timestamp']); d = min(D - 1, (s - t0) * D // max(1, t1 - t0))
    acc[k][d] += (int(r['End_Timestamp']) - s) / 1e6; cnt[k] += 1
print('kernel time (ms) by tenth of the traced run (%.1f s wall):' % ((t1 - t0) / 1e9))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
    print('%-12s n=%7d total %9.0f | ' % (k, cnt[k], sum(v)) + ' '.join('%7.0f' % x for x in v))
P
rm -rf gpurun_out/kt
tail -22 gpurun_out/${T}_build_trace_${N}.log
