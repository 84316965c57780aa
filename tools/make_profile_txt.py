#!/usr/bin/env python3
"""Compose a `profiles/*.txt` file from the CSV summaries a tools/profile_*.sh run left in gpurun_out/<dir>: the tables verbatim,
and above them a header whose every number is COMPUTED HERE from those tables (a hand-written header once quoted another
run than the table under it).

usage: make_profile_txt.py <dir> <kernel-substring> <algorithmic bytes per launch | flop:<useful FLOP per launch>> [title words ...]   > profiles/rNN_<name>.txt
"""
import glob
import json
import os
import sys

HBM_PEAK = 8.0e12


def read_tables(path):
    """[(header list, [row lists])] of one rocpd_summary.py csv (a stats table, optionally followed by a counter table)."""
    tables, cur = [], None
    for ln in open(path):
        ln = ln.rstrip("\n")
        if not ln:
            continue
        if ln.startswith("kernel,"):
            cur = (ln.split(","), [])
            tables.append(cur)
        elif cur is not None:
            # kernel names hold commas (template arguments): split from the right
            n = len(cur[0])
            parts = ln.rsplit(",", n - 1)
            cur[1].append(parts)
    return tables


def main():
    d, sub = sys.argv[1], sys.argv[2]
    flop = float(sys.argv[3][5:]) if sys.argv[3].startswith("flop:") else None
    nbytes = 0.0 if flop is not None else float(sys.argv[3])
    title = " ".join(sys.argv[4:]) or os.path.basename(d.rstrip("/"))
    files = sorted(glob.glob(os.path.join(d, "*.csv")))
    # the profile must belong to the library it claims to describe: refuse summaries older than the built library (VERDICT r4 #1c: a profile three
    # commits older than the kernel it was quoted for), and name the library by its digest (bench.py prints the same digest in its line)
    import hashlib
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multi-uav-pursuit-evasion_amd", "libhns.so")
    lib_note = ""
    if os.path.exists(lib):
        stale = [f for f in files if os.path.getmtime(f) < os.path.getmtime(lib)]
        if stale and not os.environ.get("HNS_PROFILE_ALLOW_STALE"):
            sys.stderr.write(f"make_profile_txt: {stale} older than {lib}: profile again with the current build\n")
            raise SystemExit(3)
        lib_note = "libhns.so sha256 " + hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16]
    stats, counters, pmc_ns = None, {}, []
    for f in files:
        for hdr, rows in read_tables(f):
            for r in rows:
                if sub not in r[0]:
                    continue
                if hdr[1] == "calls":
                    rec = {"calls": int(r[1]), "avg_ns": float(r[2]), "min_ns": float(r[3]), "max_ns": float(r[4])}
                    if os.path.basename(f).startswith("stats"):
                        stats = rec
                    else:
                        pmc_ns.append((os.path.basename(f), rec))
                elif hdr[1] == "counter":
                    counters[r[1]] = float(r[4])
    out = [f"# {title}", f"# composed by tools/make_profile_txt.py from {d}/*.csv (rocprofv3 --kernel-trace --stats, and separate --kernel-trace --pmc passes; "
           "tools/rocpd_summary.py); every number in this header is computed from the tables below" + (f"; {lib_note}" if lib_note else "")]
    if stats:
        t = stats["avg_ns"] * 1e-9
        head = f"# kernel `{sub}`: {stats['calls']} launches, avg {stats['avg_ns'] / 1e3:.3f} us (min {stats['min_ns'] / 1e3:.2f}, max {stats['max_ns'] / 1e3:.2f})"
        if flop is not None:
            out.append(head + f" -> {flop:.4g} useful FLOP per launch / avg = {flop / t / 1e12:.1f} TFLOP/s = {flop / t / 2.5e15:.4f} of the 2.5 PF dense f16 matrix peak")
        elif nbytes > 0:
            out.append(head + f" -> {nbytes:,.0f} B algorithmic per launch / avg = {nbytes / t / 1e12:.3f} TB/s = {nbytes / t / HBM_PEAK:.4f} of 8 TB/s")
        else:
            out.append(head)
    for name, rec in pmc_ns:
        out.append(f"# (under counter collection, {name}: {rec['calls']} launches, avg {rec['avg_ns'] / 1e3:.3f} us)")
    if "FETCH_SIZE" in counters and "WRITE_SIZE" in counters:
        tr = counters["FETCH_SIZE"] * 1024 * 2 + counters["WRITE_SIZE"] * 1024
        out.append(f"# traffic per launch: FETCH_SIZE {counters['FETCH_SIZE']:.1f} KiB x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE {counters['WRITE_SIZE']:.1f} KiB"
                   f" = {tr / 1e6:.2f} MB" + (f" = {tr / nbytes:.3f} x algorithmic" if nbytes > 0 else ""))
    if "SQ_INSTS_VALU" in counters:
        extra = "".join(f", {k} {counters[k]:.4g}" for k in ("SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_WAVES") if k in counters)
        out.append(f"# instructions per launch: SQ_INSTS_VALU {counters['SQ_INSTS_VALU']:.4g}{extra}")
    if "SQ_WAVE_CYCLES" in counters:
        wc = counters["SQ_WAVE_CYCLES"]
        parts = [f"{k} {counters[k] / wc:.3f}" for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS") if k in counters]
        out.append(f"# fractions of SQ_WAVE_CYCLES ({wc:.4g} quad-cycles per launch): " + ", ".join(parts))
    for lg in sorted(glob.glob(os.path.join(d, "*.log"))):
        for ln in open(lg, errors="replace"):
            if ln.startswith("{") and '"metric"' in ln:
                try:
                    j = json.loads(ln)
                except ValueError:
                    continue
                r = j.get("roofline") or {}
                out.append(f"# bench line of the profiled run ({os.path.basename(lg)}; under the profiler): ms_per_step {j.get('ms_per_step')}, roofline.kernel_us "
                           f"{r.get('kernel_us')} (blocks of consecutive launches), frac {r.get('frac')}, step_us {r.get('step_us')}, frac_step_rate {r.get('frac_step_rate')}")
                t = j.get("tp_mode")
                if t:
                    out.append(f"#   its predictor leg: tp_mode.observe_us {t.get('observe_us')}, roofline.frac {t['roofline'].get('frac')}, ms_per_step {t.get('ms_per_step')}, step_kernel_us {t.get('step_kernel_us')}")
    print("\n".join(out))
    for f in files:
        print(f"\n## {os.path.splitext(os.path.basename(f))[0]}")
        sys.stdout.write(open(f).read())


if __name__ == "__main__":
    main()
