#!/bin/bash
# fabric read bytes of k_gemm_tn16 at M = 1285158 (138 x (138+138), three quarters of G zero), unpaced vs paced
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for v in never paced; do
  mb=100000000; [ $v = paced ] && mb=0
  rm -rf /tmp/tnp_$v
  SPARSE=0.75 TN=1 HGVAE_TN_PACE_MIN_MB=$mb timeout -s KILL 90 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum -d /tmp/tnp_$v -o tn --output-format csv -- tools/bin/nt16_base 1285158 138 138 138 $v > /tmp/tnp_$v.log 2>&1 || { tail -5 /tmp/tnp_$v.log; exit 1; }
  grep TN /tmp/tnp_$v.log
  python3 - "$v" <<'P'
import csv, glob, sys
v = sys.argv[1]
rows = [r for f in glob.glob(f"/tmp/tnp_{v}/**/*counter_collection.csv", recursive=True) for r in csv.DictReader(open(f))]
acc = {}
for r in rows:
    if "k_gemm_tn16" in r["Kernel_Name"]:
        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
req, r32 = (sum(acc[k]) / len(acc[k]) for k in ("TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"))
print(f"{v}: launches {len(acc['TCC_EA0_RDREQ_sum'])}  read requests {req:.3e} (32 B: {r32:.3e})  bytes {(r32 * 32 + (req - r32) * 64) / 1e9:.3f} GB per launch")
P
done
