#!/bin/bash
out=/tmp/r03r
mkdir -p $out gpurun_out/r03r
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
for sz in 100 100000; do
  timeout 200 rocprofv3 --hip-trace --kernel-trace --memory-copy-trace -d $out/prof_$sz -o t -- python scripts/prof_prefilter_fused.py --sizes $sz --iters 40 > $out/run_$sz.log 2>&1; echo "size $sz rc=$? $(grep '^{' $out/run_$sz.log)"
done
python - <<'P' | tee gpurun_out/r03r/anatomy.txt
import sqlite3,glob
for sz in (100,100000):
    f=glob.glob(f'/tmp/r03r/prof_{sz}/*.db')
    if not f: continue
    db=sqlite3.connect(f[0]); cur=db.cursor()
    cols=[r[1] for r in cur.execute("pragma table_info(regions)")]
    t0=cur.execute("select min(start) from kernels where name like '%bfs_level_kernel%'").fetchone()[0]
    n=cur.execute("select count(*) from kernels where name like '%bfs_level_kernel%'").fetchone()[0]
    print(f"== candidates {sz}: {n} fused calls in the window; per call (us):")
    print(" kernels:")
    for r in cur.execute("select name, count(*), sum(end-start) from kernels where start>=? group by name order by 3 desc", (t0,)):
        print(f"   {r[0][:70]:70s} x{r[1]/n:5.2f}  {r[2]/n/1e3:8.2f}")
    print(" HIP API (host):")
    for r in cur.execute("select name, count(*), sum(end-start) from regions where start>=? group by name order by 3 desc limit 14", (t0,)):
        print(f"   {r[0][:50]:50s} x{r[1]/n:5.2f}  {r[2]/n/1e3:8.2f}")
    print(" copies:")
    try:
        for r in cur.execute("select name, count(*), sum(end-start), sum(size) from memory_copies where start>=? group by name", (t0,)):
            print(f"   {r[0][:40]:40s} x{r[1]/n:5.2f}  {r[2]/n/1e3:8.2f} us  {r[3]/n:10.0f} B")
    except Exception as e: print('  copies query failed', e, [c[1] for c in cur.execute("pragma table_info(memory_copies)")])
P
