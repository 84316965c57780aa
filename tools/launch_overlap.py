#!/usr/bin/env python3
"""Do successive launches of the step kernel overlap head to tail?  (VERDICT r4 #1a)

Reads the rocpd SQLite database of a `rocprofv3 --kernel-trace` run and prints, for the longest run of CONSECUTIVE dispatches of one kernel
(no other kernel of the process dispatched in between), every dispatch's start and end timestamp (ns, relative to the first start), its
duration, and the gap to its successor `start[i+1] - end[i]` — negative = the successor started before this dispatch ended.  The header is
computed from the table: mean duration (what `--stats` reports), mean period `start[i+1] - start[i]` (what a step costs at steady state),
the share of overlapped pairs and the mean overlap.

usage: launch_overlap.py <results.db> <kernel-substring> [max rows, default 256]      > profiles/rNN_launch_overlap.txt
"""
import sqlite3
import sys


def main():
    db, sub = sys.argv[1], sys.argv[2]
    limit = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start"))
    # longest run of consecutive dispatches whose name holds `sub`
    best, cur_run = [], []
    for st, en, name in rows:
        if sub in name:
            cur_run.append((st, en))
        else:
            if len(cur_run) > len(best):
                best = cur_run
            cur_run = []
    if len(cur_run) > len(best):
        best = cur_run
    if len(best) < 2:
        raise SystemExit(f"no run of consecutive `{sub}` dispatches in {db}")
    name = next(n for _, _, n in rows if sub in n)
    # the MIDDLE of the run: steady state, away from the process's first launches and from whatever the run ends with (bench.py ends its timed
    # region with event-bracketed blocks and 16 isolated event-bracketed dispatches, each between two idle gaps)
    mid = len(best) // 2
    run = best[max(0, mid - limit // 2):max(0, mid - limit // 2) + limit] if len(best) > limit else best
    t0 = run[0][0]
    dur = [en - st for st, en in run]
    gap = [run[i + 1][0] - run[i][1] for i in range(len(run) - 1)]
    period = [run[i + 1][0] - run[i][0] for i in range(len(run) - 1)]
    over = [-g for g in gap if g < 0]
    n = len(run)
    print(f"# launch overlap of `{name[:90]}`")
    print(f"# tools/launch_overlap.py on a rocprofv3 --kernel-trace database: the middle {n} of {len(best)} CONSECUTIVE dispatches of the kernel (nothing else dispatched in between)")
    print(f"# duration end - start: mean {sum(dur) / n / 1e3:.3f} us (min {min(dur) / 1e3:.2f}, max {max(dur) / 1e3:.2f})")
    print(f"# period start[i+1] - start[i]: mean {sum(period) / len(period) / 1e3:.3f} us  -> a step costs the period, a profiler's per-kernel average is the duration")
    print(f"# gap start[i+1] - end[i]: mean {sum(gap) / len(gap) / 1e3:+.3f} us; {len(over)} of {len(gap)} pairs overlap (successor started before this dispatch ended)"
          + (f", mean overlap of those {sum(over) / len(over) / 1e3:.3f} us (max {max(over) / 1e3:.2f})" if over else ""))
    print(f"# mean duration - mean period = {(sum(dur) / n - sum(period) / len(period)) / 1e3:+.3f} us per step")
    # the whole run of consecutive dispatches in blocks of 64: warm-up (clocks, first launches) against steady state
    print(f"# whole run of {len(best)} consecutive dispatches, blocks of 64: mean duration / mean period (us)")
    for b0 in range(0, len(best) - 1, 64):
        blk = best[b0:b0 + 65]
        if len(blk) < 2:
            break
        d_ = [en - st for st, en in blk[:-1]]
        p_ = [blk[i + 1][0] - blk[i][0] for i in range(len(blk) - 1)]
        print(f"#   launches {b0:5d}-{b0 + len(blk) - 2:5d}: duration {sum(d_) / len(d_) / 1e3:7.3f}   period {sum(p_) / len(p_) / 1e3:7.3f}")
    import statistics
    print(f"# median gap {statistics.median(gap) / 1e3:+.3f} us, median duration {statistics.median(dur) / 1e3:.3f} us, median period {statistics.median(period) / 1e3:.3f} us (the table's rows)")
    print("i,start_ns,end_ns,duration_ns,gap_to_next_ns")
    for i, (st, en) in enumerate(run):
        print(f"{i},{st - t0},{en - t0},{en - st},{gap[i] if i < len(gap) else ''}")


if __name__ == "__main__":
    main()
