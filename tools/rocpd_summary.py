"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share,
optionally split by grid size.  Usage: python tools/rocpd_summary.py <results.db> [--by-grid] [--top N]
       python tools/rocpd_summary.py <results.db> --timeline N   (the last N dispatches: start offset, stream/queue, duration)"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    by_grid = "--by-grid" in sys.argv
    top = 40
    if "--top" in sys.argv:
        top = int(sys.argv[sys.argv.index("--top") + 1])
    cur = db.cursor()
    if "--timeline" in sys.argv:
        n = int(sys.argv[sys.argv.index("--timeline") + 1])
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        qcol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0")
        rows = list(cur.execute("select name, start, end, %s, grid_x from kernels order by start desc limit %d" % (qcol, n)))
        rows.reverse()
        t0 = rows[0][1]
        print("columns:", cols)
        for name, st, en, q, gx in rows:
            print("%10.1f us  +%8.1f us  q=%-4s grid=%-7s %s" % ((st - t0) / 1e3, (en - st) / 1e3, q, gx, name[:60]))
        return
    if "--flowgaps" in sys.argv:
        # last flow-net pass: per-queue busy time vs span, and the idle gaps between consecutive kernels of the queue
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        qcol = "stream_id" if "stream_id" in cols else "queue_id"
        rows = list(cur.execute("select name, start, end, %s from kernels order by start" % qcol))
        ends = [i for i, r in enumerate(rows) if "k_flow_consistency" in r[0]]
        starts = [i for i, r in enumerate(rows) if "k_img_u8_to_flow_input" in r[0]]
        e = ends[-1]
        st = max(i for i in starts[::2] if i < e)
        q = rows[e][3]
        ks = [r for r in rows[st:e + 1] if r[3] == q]
        span = (ks[-1][2] - ks[0][1]) / 1e3
        busy = sum(r[2] - r[1] for r in ks) / 1e3
        gaps = [(ks[i + 1][1] - ks[i][2]) / 1e3 for i in range(len(ks) - 1)]
        print("flow pass: %d kernels, span %.1f us, busy %.1f us, idle %.1f us (mean gap %.2f us, max %.1f us)" % (
            len(ks), span, busy, span - busy, sum(gaps) / len(gaps), max(gaps)))
        import collections
        byname = collections.defaultdict(lambda: [0, 0.0])
        for r in ks:
            k = r[0].split("(")[0][-48:]
            byname[k][0] += 1
            byname[k][1] += (r[2] - r[1]) / 1e3
        for k, v in sorted(byname.items(), key=lambda kv: -kv[1][1])[:14]:
            print("  %-50s n=%3d %8.1f us" % (k, v[0], v[1]))
        return
    if "--netspans" in sys.argv:
        # flow-net passes: first k_img_u8_to_flow_input -> k_flow_consistency; busy = sum of net kernel durations inside
        rows = list(cur.execute("select name, start, end from kernels order by start"))
        starts = [r[1] for r in rows if "k_img_u8_to_flow_input" in r[0]][::2]
        ends = [r[2] for r in rows if "k_flow_consistency" in r[0]]
        print("flow-net passes: %d" % len(ends))
        prev_end = None
        for i, e in enumerate(ends):
            st = max(x for x in starts if x < e)
            gap = (st - prev_end) / 1e3 if prev_end else 0.0
            print("  pass %2d: span %8.1f us   gap since previous pass end %8.1f us" % (i, (e - st) / 1e3, gap))
            prev_end = e
        return
    if "--spans" in sys.argv:
        # per-pair span of the solver stage (k_kp_count start -> last k_scale_ransac / k_pnp_select end) and of the nets
        rows = list(cur.execute("select name, start, end from kernels order by start"))
        starts = [r for r in rows if "k_kp_count" in r[0]]
        ends = [r for r in rows if "k_scale_ransac" in r[0] or "k_pnp_select" in r[0]]
        spans = []
        for i, st in enumerate(starts):
            nxt = starts[i + 1][1] if i + 1 < len(starts) else 1 << 62
            e = [r[2] for r in ends if st[1] < r[2] < nxt]
            if e:
                busy = sum(r[2] - r[1] for r in rows if st[1] <= r[1] < max(e) and ("conv_" not in r[0]) and r[0].startswith("dfvo::k_") and not any(
                    t in r[0] for t in ("k_correlation", "k_warp", "k_reg_", "k_deconv", "k_flow", "k_resize", "k_img", "k_maxpool", "k_depth", "k_disp")))
                spans.append(((max(e) - st[1]) / 1e3, busy / 1e3))
        print("solver-stage spans per pair (us): n=%d" % len(spans))
        for sp, busy in spans[-12:]:
            print("  span %8.1f   sum of its kernel durations %8.1f" % (sp, busy))
        return
    key = "name, grid_x, grid_y, grid_z" if by_grid else "name"
    rows = list(cur.execute(
        "select %s, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by %s "
        "order by sum(duration) desc" % (key, key)))
    total = sum(r[-4] for r in rows)
    print("total kernel time %.3f ms over %d dispatches" % (total / 1e6, sum(r[-5] for r in rows)))
    print("%-72s %7s %11s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%"))
    for r in rows[:top]:
        name = r[0]
        if by_grid:
            name = "%s [%dx%dx%d]" % (r[0][:48], r[1], r[2], r[3])
        n, tot, avg, mn, mx = r[-5:]
        print("%-72s %7d %11.1f %10.2f %10.2f %10.2f %6.2f" % (name[:72], n, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3,
                                                               100.0 * tot / total))


if __name__ == "__main__":
    main()
