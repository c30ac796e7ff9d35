# usage (GPU box, repo root): tools/pipe3_trace.sh [tag]  - kernel trace of the request as one call in its pipelined forms; how much of the three hot kernels' time overlaps
T=${1:-r05}; R=$(pwd); export TMPDIR=/tmp; cd /tmp
for V in "GS_REQUEST_PIPELINE=1" "GS_REQUEST_PIPELINE=3,GS_PIPE_SKETCH_LDS=0" "GS_REQUEST_PIPELINE=3,GS_PIPE_SKETCH_LDS=102400"; do
  rm -rf $R/gpurun_out/p3t
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/p3t -o p3 -- python $R/tools/request_fused_probe.py 300000 10000 2 "$V" > $R/gpurun_out/p3t.log 2>&1
  cd $R
  V="$V" python - <<'P' >> gpurun_out/${T}_pipe3_overlap.txt
import csv, glob, os
fn = glob.glob('gpurun_out/p3t/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(fn))]
def cls(n):
    return 'sketch' if 'k_sketch_min' in n else 'join' if 'k_match_join' in n else 'traversal' if 'k_hnsw_search_dense' in n else None
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), cls(r['Kernel_Name'])) for r in rows if cls(r['Kernel_Name'])]
# the LAST one-call request: from the last sketch launch group of the one-call variant ... take the last 4 traversal launches (pipeline 3) or the last one
tr = sorted([e for e in ev if e[2] == 'traversal'])
# the one-call runs are the middle of the run (two calls first and last): use every event between the 3rd and the 2nd-to-last search
def overlap(a, b):
    tot = 0
    for s1, e1, _ in a:
        for s2, e2, _ in b:
            lo, hi = max(s1, s2), min(e1, e2)
            if hi > lo: tot += hi - lo
    return tot / 1e6
S = [e for e in ev if e[2] == 'sketch' and e[1] - e[0] > 5e6]; J = [e for e in ev if e[2] == 'join' and e[1] - e[0] > 1e6]; T = [e for e in tr if e[1] - e[0] > 1e6]
print('variant [%s]: over the whole probe run (two calls x 2, one call x 2, two calls x 2; build excluded by kernel size):' % os.environ['V'])
print('  kernel ms: sketch %.0f in %d launches, join %.0f in %d, traversal %.0f in %d' % (sum(e[1] - e[0] for e in S) / 1e6, len(S), sum(e[1] - e[0] for e in J) / 1e6, len(J), sum(e[1] - e[0] for e in T) / 1e6, len(T)))
print('  concurrent ms: sketch||join %.1f, sketch||traversal %.1f, join||traversal %.1f' % (overlap(S, J), overlap(S, T), overlap(J, T)))
P
  grep -a "one call\|two calls" gpurun_out/p3t.log >> gpurun_out/${T}_pipe3_overlap.txt
  cd /tmp
done
rm -rf $R/gpurun_out/p3t; cat $R/gpurun_out/${T}_pipe3_overlap.txt
