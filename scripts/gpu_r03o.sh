#!/bin/bash
# round-3 GPU call O: where the batched link step spends its time (kernel totals, contention counters)
out=gpurun_out/r03o
mkdir -p $out
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
export HVX_LIB_PATH=$GRAFT_REPO_ROOT/helix-db_amd/libhelix_vec_gfx950_tuning.so
HVX_BUILD_DEBUG=1 timeout 300 rocprofv3 --kernel-trace --stats -d $out/prof -o t -- python scripts/bench_build.py 1000000 2048 32 embedding 0 > $out/build.log 2>&1; echo "rc=$?"; grep '^{\|hvx build' $out/build.log
python - <<'P'
import sqlite3,glob
db=sqlite3.connect(glob.glob('gpurun_out/r03o/prof/*.db')[0]); cur=db.cursor()
for r in cur.execute("select name, count(*), avg(end-start), min(end-start), max(end-start), sum(end-start) from kernels group by name order by 6 desc limit 8"):
    print(r[0][:80], r[1], 'avg_us', round(r[2]/1e3,1), 'max_us', round(r[4]/1e3,1), 'total_ms', round(r[5]/1e6,1))
# link kernel duration by batch index (every 50th)
rows=[r for r in cur.execute("select start, end-start from kernels where name like '%build_link_wg%' order by start")]
print('link wg by batch:', [round(d/1e3) for i,(s,d) in enumerate(rows) if i%60==0])
rows=[r for r in cur.execute("select start, end-start from kernels where name like '%hnsw_wave_kernel%' order by start")]
print('search by batch:', [round(d/1e3) for i,(s,d) in enumerate(rows) if i%60==0])
P
