R=$(pwd); export TMPDIR=/tmp; cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt -o kt -- python $R/tools/trav_ab.py --genomes --reps 1 --nq 256 "" > $R/gpurun_out/kt.log 2>&1
cd $R
python - <<'P'
import csv, glob
fn = glob.glob('gpurun_out/kt/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(fn)))
names = {}
for r in rows:
    n = r['Kernel_Name']
    key = 'join' if 'k_match_join' in n else 'sample' if 'k_match_sample' in n else 'dense' if 'k_hnsw_search_dense' in n else 'plan' if 'k_hnsw_plan' in n else 'merge' if 'k_link_merge' in n else 'link' if 'k_link_' in n else 'cache' if 'k_cache_rows' in n else 'r2c' if 'k_rows_to_cols' in n else 'fill' if 'fillBuffer' in n else 'copy' if 'copyBuffer' in n else 'tile' if 'k_hamming_qxc' in n else 'sketch' if 'k_sketch_min' in n else None
    if key: names.setdefault(key, []).append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp'])))
if 'plan' in names:
    v = sorted(names['plan'])
    a = [x[1] / 1e6 for x in v[0::2]]; b = [x[1] / 1e6 for x in v[1::2]]
    print('plan alternate launches (phase 1 / phase 2 once the pre-pass is on): %.0f ms / %.0f ms' % (sum(a), sum(b)))
for k, v in names.items():
    v.sort()
    d = [x[1] / 1e6 for x in v]
    q = len(d) // 8
    print(k, len(d), 'total %.0f ms' % sum(d), 'by octile of the build (ms avg):', ' '.join('%.2f' % (sum(d[i*q:(i+1)*q]) / max(q,1)) for i in range(8)))
P
rm -rf gpurun_out/kt
