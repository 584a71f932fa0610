#!/bin/bash
# the decode of one context beside ONE kind of other work on a second context: durations of its chain / walk kernels (rocprofv3 kernel trace)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for o in none transform encode decode; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/corun_$o -o p -- python3 $R/tools/corun_walk.py $o 6 > $R/gpurun_out/corun_$o.log 2>&1 || { echo "run $o failed"; tail -3 $R/gpurun_out/corun_$o.log; exit 1; }
  python3 - "$R/gpurun_out/corun_$o/p_kernel_trace.csv" "$o" <<'P'
import csv,re,sys,collections
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void xpng::","").replace("xpng::","")
    if any(k in n for k in ("walk","dec_chain","dec_resid","dec_recon_band","dec_alpha")):
        wg=int(r["Grid_Size_X"])//int(r["Workgroup_Size_X"])
        if wg>=60: d[(n,wg)].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6)
print("== decode beside:", sys.argv[2])
for k,v in sorted(d.items()):
    v=sorted(v); print("   %-40s wg %6d  n %3d  min %6.2f  med %6.2f  max %6.2f"%(k[0][:40],k[1],len(v),v[0],v[len(v)//2],v[-1]))
P
done
