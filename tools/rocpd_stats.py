"""Per-kernel totals from a rocprofv3 run that wrote a rocpd database (*.db) instead of CSV:
    python tools/rocpd_stats.py gpurun_out/prof_dir [top]"""
import glob
import sqlite3
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
f = glob.glob(d + "/**/*.db", recursive=True)[0]
con = sqlite3.connect(f)
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e6 from {kd} d "
     f"join {ks} s on d.kernel_id=s.id group by s.kernel_name order by 4 desc limit {top}")
print("kernel,calls,avg_us,total_ms")
for r in con.execute(q):
    print(f"{r[0]},{r[1]},{r[2]:.1f},{r[3]:.2f}")
