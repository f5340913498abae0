#!/usr/bin/env python3
"""Turn rocprofv3 output (the rocpd sqlite database of `rocprofv3 --kernel-trace [--stats] [--pmc X]`) into the
small text summaries committed under profiles/.

    python tools/rocprof_summary.py stats  gpurun_out/prof_stats/bench_results.db  > profiles/r01_kernel_stats.txt
    python tools/rocprof_summary.py pmc    gpurun_out/prof_fetch/bench_results.db  > profiles/r01_pmc_FETCH_SIZE.txt
    python tools/rocprof_summary.py traffic FETCH.db WRITE.db > profiles/traffic.json

HBM traffic per launch follows MI355X_MICROARCH.md "HBM": rocprofv3 reports FETCH_SIZE / WRITE_SIZE in kilobytes
(value * 1024 = bytes); on gfx950 FETCH_SIZE counts 128-B read requests as 64 B, so the read side is doubled.
WRITE_SIZE was calibrated on this repo's fill2_kernel (2 x 512 MiB written -> 1048576.0 reported): taken as is.
"""
import json
import sqlite3
import sys


def short(name):
    name = name.split("(")[0]
    for pre in ("void ", "tsdf::"):
        name = name.replace(pre, "")
    return name


def kernel_stats(db):
    """Per kernel: calls, total, mean, min, max, share -- and the MEDIAN launch: the mean of a bench run also holds the first casts on
    an empty volume and the launches of the warm-up, several times the steady state's."""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) "
                       "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    out = ["%-70s %6s %14s %12s %12s %12s %7s %12s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct", "median_ns")]
    for name, calls, tot, avg, mn, mx in rows:
        durs = sorted(r[0] for r in cur.execute("select duration from kernels where name = ?", (name,)).fetchall())
        med = durs[len(durs) // 2] if durs else 0
        out.append("%-70s %6d %14d %12.0f %12d %12d %6.2f%% %12d" % (short(name)[:70], calls, tot, avg, mn, mx, 100.0 * tot / total, med))
    return "\n".join(out)


def pmc_by_kernel(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
                       "from counters_collection group by kernel_name, counter_name order by avg(value) desc").fetchall()
    return rows


def pmc_text(db):
    out = ["%-60s %-12s %6s %16s %16s %16s %12s" % ("kernel", "counter", "calls", "avg", "min", "max", "avg_ns")]
    for k, c, n, a, mn, mx, d in pmc_by_kernel(db):
        out.append("%-60s %-12s %6d %16.1f %16.1f %16.1f %12.0f" % (short(k)[:60], c, n, a, mn, mx, d or 0))
    return "\n".join(out)


def timeline(db, first=0, count=60):
    """Kernels in start order with the gap since the previous kernel's end and the overlap with it (ns): where a step's time
    goes between its launches.  python tools/rocprof_summary.py timeline run.db [first] [count]"""
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = cur.execute("select name, start, end, %s from kernels order by start" % q).fetchall()
    out = ["%-58s %6s %12s %10s %10s" % ("kernel", "queue", "start_us", "dur_us", "gap_us")]
    t0 = rows[0][1] if rows else 0
    prev_end = None
    for name, st, en, qid in rows[first:first + count]:
        gap = (st - prev_end) / 1e3 if prev_end is not None else 0.0
        out.append("%-58s %6s %12.1f %10.1f %10.1f" % (short(name)[:58], qid, (st - t0) / 1e3, (en - st) / 1e3, gap))
        prev_end = max(prev_end or en, en)
    return "\n".join(out)


def overlaps(db, skip_first=0):
    """What runs BESIDE each kernel (other queues): per kernel name, launches, mean duration, and per overlapping kernel name the mean
    overlap per launch -- for two-stream schedules (tsdf_pipeline_step).  Also the mean gap between a kernel's end and the start of
    the next kernel on the same queue.  python tools/rocprof_summary.py overlap run.db [skip the first n kernels]"""
    cur = sqlite3.connect(db).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = cur.execute("select name, start, end, %s from kernels order by start" % q).fetchall()[skip_first:]
    rows = [(short(n).split("<")[0], s, e, qq) for n, s, e, qq in rows]
    stat, gaps = {}, {}
    by_queue = {}
    for r in rows:
        by_queue.setdefault(r[3], []).append(r)
    for qq, lst in by_queue.items():
        for a, b in zip(lst, lst[1:]):
            g = gaps.setdefault((a[0], b[0]), [0, 0.0])
            g[0] += 1; g[1] += (b[1] - a[2]) / 1e3
    j0 = 0
    for i, (n, s, e, qq) in enumerate(rows):
        st = stat.setdefault(n, {"n": 0, "dur": 0.0, "beside": {}, "queues": set()})
        st["n"] += 1; st["dur"] += (e - s) / 1e3; st["queues"].add(qq)
        while j0 < len(rows) and rows[j0][2] < s - 2_000_000:
            j0 += 1
        for n2, s2, e2, q2 in rows[j0:]:
            if s2 > e:
                break
            if q2 == qq:
                continue
            ov = min(e, e2) - max(s, s2)
            if ov > 0:
                st["beside"][n2] = st["beside"].get(n2, 0.0) + ov / 1e3
    out = ["%-30s %6s %8s %10s   %s" % ("kernel", "calls", "queues", "avg_us", "mean overlap per launch with kernels of other queues (us)")]
    for n, st in sorted(stat.items(), key=lambda kv: -kv[1]["dur"]):
        bes = ", ".join("%s %.1f" % (k, v / st["n"]) for k, v in sorted(st["beside"].items(), key=lambda kv: -kv[1]) if v / st["n"] >= 0.05)
        out.append("%-30s %6d %8s %10.1f   %s" % (n[:30], st["n"], ",".join(str(x) for x in sorted(st["queues"])), st["dur"] / st["n"], bes or "-"))
    out.append("")
    out.append("%-62s %6s %10s" % ("same queue: end of A -> start of B", "pairs", "mean_gap_us"))
    for (a, b), (c, g) in sorted(gaps.items(), key=lambda kv: -kv[1][0]):
        if c >= 3:
            out.append("%-62s %6d %10.1f" % ((a[:30] + " -> " + b[:28]), c, g / c))
    return "\n".join(out)


def launches_of(db, counter):
    """{kernel display name: [(start, value, duration)] in dispatch order} for one counter."""
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, start, value, duration from counters_collection where counter_name = ? order by start",
                       (counter,)).fetchall()
    out = {}
    for k, st, v, d in rows:
        out.setdefault(k, []).append((st, v, d))
    return out


def traffic(fetch_db, write_db, skip=0, take=0, meta=None):
    """Bytes per launch, averaged over launches [skip, skip + take) of each kernel's most-launched template variant (the
    production one): with skip = warm-up steps and take = timed steps these are exactly the launches bench.py's timed region
    issued (the replays behind it are left out)."""
    res = {}
    for db, cname, scale in ((fetch_db, "FETCH_SIZE", 2.0), (write_db, "WRITE_SIZE", 1.0)):
        best = {}
        for k, rows in launches_of(db, cname).items():
            key = short(k).split("<")[0]
            if key not in best or len(rows) > len(best[key][1]):
                best[key] = (short(k), rows)
        for key, (variant, rows) in best.items():
            sel = rows[skip:skip + take] if take else rows
            if not sel:
                sel = rows
            a = sum(v for _, v, _ in sel) / len(sel)
            e = res.setdefault(key, {})
            e[cname + "_avg_raw_kb"] = a
            e[cname + "_variant"] = variant
            e[cname + "_launches_averaged"] = len(sel)
            e[cname + "_launches_total"] = len(rows)
            e[cname.lower().replace("_size", "") + "_bytes"] = a * 1024.0 * scale
    out = {"note": "bytes per launch = FETCH_SIZE*1024*2 (gfx950 half-count correction, MI355X_MICROARCH.md HBM) + "
                   "WRITE_SIZE*1024 (calibrated on fill2_kernel); separate --pmc passes of the bench command; mean over the "
                   "launches of the timed region (after `skip` warm-up launches)",
           "bytes_per_launch": {}, "detail": res}
    if meta:
        out.update(meta)
    for k, e in res.items():
        out["bytes_per_launch"][k] = int(e.get("fetch_bytes", 0) + e.get("write_bytes", 0))
    return json.dumps(out, indent=1)


def activity(db, skip=0, take=0, clock_ghz=2.4, simds=1024):
    """Per kernel (most-launched variant, launches [skip, skip + take) as in traffic()): vector instructions per launch, the share of
    the launch in which the SIMDs issue them, and the lanes active per vector instruction -- from ONE pass with
    SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES.
      valu_busy    = 4 * SQ_ACTIVE_INST_VALU / (launch duration * clock * SIMDs): SQ_ACTIVE_INST_VALU counts quad-cycles summed over
                     the chip's SIMDs (a wave64 vector instruction occupies its SIMD for 4 cycles); 2.4 GHz x 1024 SIMDs (MI355X)
      lanes_active = SQ_THREAD_CYCLES_VALU / SQ_INSTS_VALU / 64 (thread-cycles per vector instruction over the wave's 64 lanes)"""
    names = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_THREAD_CYCLES_VALU", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES")
    per = {}
    for c in names:
        best = {}
        for k, rows in launches_of(db, c).items():
            key = short(k).split("<")[0]
            if key not in best or len(rows) > len(best[key]):
                best[key] = rows
        for key, rows in best.items():
            sel = rows[skip:skip + take] if take else rows
            sel = sel or rows
            e = per.setdefault(key, {})
            e[c] = sum(v for _, v, _ in sel) / len(sel)
            e["avg_ns"] = sum(d or 0 for _, _, d in sel) / len(sel)
            e["launches_averaged"] = len(sel)
    out = {}
    for key, e in per.items():
        if not e.get("SQ_INSTS_VALU") or not e.get("avg_ns"):
            continue
        out[key] = {"valu_insts": int(e["SQ_INSTS_VALU"]),
                    "valu_busy": round(4.0 * e.get("SQ_ACTIVE_INST_VALU", 0.0) / (e["avg_ns"] * clock_ghz * simds), 4),
                    "lanes_active": round(e.get("SQ_THREAD_CYCLES_VALU", 0.0) / e["SQ_INSTS_VALU"] / 64.0, 4),
                    "waves_resident_mean": round(e.get("SQ_WAVE_CYCLES", 0.0) / (e["avg_ns"] * clock_ghz), 1) if e.get("SQ_WAVE_CYCLES") else None,
                    "avg_ns_under_counters": int(e["avg_ns"]), "launches_averaged": e["launches_averaged"]}
    return out


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "stats":
        print(kernel_stats(sys.argv[2]))
    elif mode == "pmc":
        print(pmc_text(sys.argv[2]))
    elif mode == "timeline":
        print(timeline(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0, int(sys.argv[4]) if len(sys.argv) > 4 else 60))
    elif mode == "overlap":
        print(overlaps(sys.argv[2], int(sys.argv[3]) if len(sys.argv) > 3 else 0))
    elif mode == "traffic":
        # traffic FETCH.db WRITE.db [skip take [key=value ...]]   (key=value pairs go into the JSON: tag, steps, warmup, ...)
        skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
        take = int(sys.argv[5]) if len(sys.argv) > 5 else 0
        meta = {}
        for kv in sys.argv[6:]:
            k, v = kv.split("=", 1)
            try:
                meta[k] = json.loads(v) if v[:1] in "{[0123456789" else v
            except ValueError:      # (a hex digest that starts with a digit)
                meta[k] = v
        print(traffic(sys.argv[2], sys.argv[3], skip, take, meta))
    elif mode == "traffic+activity":
        # traffic+activity FETCH.db WRITE.db ACTIVITY.db skip take [key=value ...]: traffic()'s JSON with an "activity" object beside it
        skip, take = int(sys.argv[5]), int(sys.argv[6])
        meta = {}
        for kv in sys.argv[7:]:
            k, v = kv.split("=", 1)
            try:
                meta[k] = json.loads(v) if v[:1] in "{[0123456789" else v
            except ValueError:
                meta[k] = v
        d = json.loads(traffic(sys.argv[2], sys.argv[3], skip, take, meta))
        d["activity"] = activity(sys.argv[4], skip, take)
        d["activity_note"] = activity.__doc__.strip()
        print(json.dumps(d, indent=1))
