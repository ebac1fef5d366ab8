cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02H
export TMPDIR=/tmp
for o in 1 2; do
timeout 300 rocprofv3 --kernel-trace -d gpurun_out/r02H/p$o -o t -- python tools/bench_cnn.py --steps 3 --warmup 1 --dtype bf16 --batch 16 --planes 64 --opt 2=$o > gpurun_out/r02H/p$o.log 2>&1
tail -1 gpurun_out/r02H/p$o.log
python - <<PY
import sqlite3
c=sqlite3.connect('gpurun_out/r02H/p$o/t_results.db')
rows=c.execute("select name,start,end from kernels order by start").fetchall()
ck=[r for r in rows if 'conv_igemm' in r[0] or 'conv_halo' in r[0]][-18:]
names=["conv1_1","conv1_2","conv2_1","conv2_2","conv3_1","conv3_2","conv3_3","conv4_1","conv4_2","conv4_3","conv6_1","conv6_2","conv6_3","conv7_1","conv7_2","conv8_1","conv8_2","head"]
print("BIGTILE=$o", ' '.join("%s:%s %.0f"%(n,('H' if 'halo' in r[0] else 'T')+r[0].split('kernel<')[1].split('>')[0].replace(', ','_'),(r[2]-r[1])/1e3) for n,r in zip(names,ck)))
PY
done
