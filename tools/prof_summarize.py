"""Reduce a rocprofv3 results database (rocpd sqlite) to the small CSVs kept under profiles/:
  kernel trace -> name, calls, total_ms, avg_us, pct          (view `kernels`)
  --pmc pass   -> name, counter, dispatches, total, per_dispatch (view `pmc_events`; a counter's rows of one dispatch are summed)
usage: prof_summarize.py results.db out_prefix"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1]); out = sys.argv[2]
short = lambda n: re.sub(r'\(.*', '', n).replace('void ', '').replace('h_bf16::', '').replace('h_f16::', 'f16:').strip()
rows = db.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
if rows:
  tot = sum(r[2] for r in rows)
  with open(out + '_kernel_stats.csv', 'w') as f:
    f.write('kernel,calls,total_ms,avg_us,pct\n')
    for n, c, ms, avg in rows:
      f.write(f'"{short(n)}",{c},{ms:.3f},{avg:.1f},{100*ms/tot:.2f}\n')
    f.write(f'"TOTAL",{sum(r[1] for r in rows)},{tot:.3f},,100\n')
try:
  pm = db.execute("select name, counter_name, count(distinct dispatch_id), sum(counter_value) from pmc_events group by name, counter_name order by 4 desc").fetchall()
except sqlite3.OperationalError:
  pm = []
if pm:
  with open(out + '_pmc.csv', 'w') as f:
    f.write('kernel,counter,dispatches,total,per_dispatch\n')
    for n, cn, nd, tv in pm:
      if nd and tv: f.write(f'"{short(n)}",{cn},{nd},{tv:.6g},{tv/nd:.6g}\n')
