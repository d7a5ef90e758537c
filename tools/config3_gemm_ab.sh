#!/bin/bash
# Round 5: the 200-mode configuration with the vendor's dgemm for its wide Gram blocks and basis updates (default)
# against our own fp64 MFMA kernels for them (MH_TEST=own_gemm); time per solve, then the kernel statistics of each.
out=gpurun_out/r05_config3_gemm_ab.txt
export TMPDIR=/tmp
: > $out
for v in "X=0" "MH_TEST=own_gemm" "X=0" "MH_TEST=own_gemm"; do
  echo "== $v" >> $out
  env $v python tools/scan_probe.py config3_s100k_repaired skillet_s100k config3_s30k --reps 2 2>&1 | grep workload | python -c "import sys,json
for l in sys.stdin:
    r=json.loads(l); print(r['workload'], r['iterations'], r['all_ms'], r.get('max_rel_err_vs_oracle'))" >> $out
done
for v in "X=0" "MH_TEST=own_gemm"; do
  rm -rf /tmp/c3prof
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c3prof -- python3 tools/scan_probe.py config3_s100k_repaired --reps 1 > /dev/null 2>&1
  f=$(ls -t /tmp/c3prof/*/*kernel_stats.csv | head -1)
  tag=$( [ "$v" = "X=0" ] && echo vendor || echo own )
  cp $f gpurun_out/r05_config3_s100k_repaired_${tag}_kernel_stats.csv
  echo "== kernel statistics, $v (2 solves)" >> $out
  python3 - $f >> $out <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
lib = sum(float(r["TotalDurationNs"]) for r in rows if r["Name"].startswith("Cijk"))
print("total %.1f ms, vendor Cijk_* %.1f ms = %.1f %%" % (tot / 1e6, lib / 1e6, 100 * lib / tot))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:12]:
    print("%8.1f ms %5.1f %% x%6s avg %8.1f us  %s" % (float(r["TotalDurationNs"]) / 1e6, 100 * float(r["TotalDurationNs"]) / tot, r["Calls"], float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
done
cat $out
