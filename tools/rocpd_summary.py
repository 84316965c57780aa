#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd SQLite output (ROCm 7.2 default format): per-kernel time stats
and per-kernel PMC counter totals.  usage: rocpd_summary.py <results.db> [kernel-substring]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    like = f"%{sys.argv[2]}%" if len(sys.argv) > 2 else "%"
    cur = sqlite3.connect(db).cursor()
    print("kernel,calls,avg_ns,min_ns,max_ns,total_ns")
    for r in cur.execute(
            "select s.kernel_name,count(*),avg(d.end-d.start),min(d.end-d.start),max(d.end-d.start),sum(d.end-d.start) "
            "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id "
            "where s.kernel_name like ? group by s.kernel_name order by 6 desc limit 24", (like,)):
        print(f"{r[0][:70]},{r[1]},{r[2]:.0f},{r[3]},{r[4]},{r[5]}")
    rows = list(cur.execute(
        "select s.kernel_name,p.name,count(*),avg(e.value),count(distinct d.id) from rocpd_pmc_event e "
        "join rocpd_info_pmc p on e.pmc_id=p.id join rocpd_kernel_dispatch d on d.event_id=e.event_id "
        "join rocpd_info_kernel_symbol s on d.kernel_id=s.id where s.kernel_name like ? group by s.kernel_name,p.name", (like,)))
    if rows:
        print("kernel,counter,avg_per_instance,instances_per_dispatch,per_dispatch_total")
        for k, name, n, avg, nd in rows:
            inst = n // max(nd, 1)
            print(f"{k[:50]},{name},{avg:.6g},{inst},{avg * inst:.6g}")


if __name__ == "__main__":
    main()
