#!/usr/bin/env python3
"""CU-time ledger of bench.py's timed region from a rocprofv3 --kernel-trace CSV (round 5, review item 1b).

    python scripts/ledger_r05.py <kernel_trace.csv> <forwards in the window> [<alone kernel_trace.csv>]

The window is the last N forwards of the trace (N = the timed steps: from the end of the deep-filter launch N+1 from the end to the
end of the last one).  Every dispatch is clipped to the window.  Scan kernels hold one CU per workgroup for their whole duration
(112-147 KiB of LDS or 16 waves x 128 registers: nothing co-resides), so their CU-time is exact: duration x workgroups.  The
time-parallel kernels (features, input products, projections, deep filter) launch more workgroups than the chip holds and take what
the scans leave; the trace cannot say how many CUs they held, so the ledger reports (a) the scans' CU-ms, (b) the time-weighted
distribution of CUs DEMANDED by scans, (c) the fraction of the window during which at least one time-parallel kernel was running
(= the free CUs had work), and (d) per kernel class: dispatches, mean duration in the window, and -- with a second trace of the same
forwards run one at a time -- the duration alone on the chip."""
import collections, csv, json, re, sys

N_CU = 256


def load(path):
    rows = list(csv.DictReader(open(path)))
    out = []
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0) * max(1, int(r.get("Grid_Size_Y", 1) or 1)) * max(1, int(r.get("Grid_Size_Z", 1) or 1))
        wx = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1) * max(1, int(r.get("Workgroup_Size_Y", 1) or 1)) * max(1, int(r.get("Workgroup_Size_Z", 1) or 1))
        out.append(dict(name=r["Kernel_Name"], s=s, e=e, wgs=max(1, gx // max(1, wx)), q=r.get("Queue_Id", "")))
    out.sort(key=lambda d: d["s"])
    return out


def cls(name):
    m = re.match(r"(?:void\s+)?([A-Za-z0-9_:]+)(<[^(]*>)?", name)
    base = m.group(1) if m else name
    if base.startswith("at::") or "elementwise" in name or "Fill" in name:
        return "aten:" + ("fill" if "Fill" in name else "other")
    t = (m.group(2) or "") if m else ""
    return base + (t if base.startswith("gsn_") else "")


def is_scan(c):
    return c.startswith("gsn_")


def main():
    rows = load(sys.argv[1])
    n_fwd = int(sys.argv[2])
    alone = load(sys.argv[3]) if len(sys.argv) > 3 else None
    df = sorted(r["e"] for r in rows if "deepfilter" in r["name"])
    assert len(df) > n_fwd, (len(df), n_fwd)
    lo, hi = df[-n_fwd - 1], df[-1]
    wall = (hi - lo) / 1e6  # ms
    acc = collections.defaultdict(lambda: dict(n=0, dur=0.0, cu_ms=0.0, wgs=0))
    events = []
    for r in rows:
        s, e = max(r["s"], lo), min(r["e"], hi)
        if e <= s:
            continue
        c = cls(r["name"])
        a = acc[c]
        a["n"] += 1
        a["dur"] += (e - s) / 1e6
        a["wgs"] = max(a["wgs"], r["wgs"])
        if is_scan(c):
            a["cu_ms"] += (e - s) / 1e6 * r["wgs"]
            events += [(s, r["wgs"], 0), (e, -r["wgs"], 0)]
        else:
            events += [(s, 0, 1), (e, 0, -1)]
    events.sort()
    # time-weighted: CUs demanded by scans; whether a time-parallel kernel is active
    hist = collections.defaultdict(float)
    tp_active = over = idle_free = 0.0
    cur, ntp, last = 0, 0, lo
    scan_cu_ms_capped = 0.0
    for t, dw, dn in events:
        dt = (t - last) / 1e6
        if dt > 0:
            hist[min(cur, 320) // 32 * 32] += dt
            scan_cu_ms_capped += dt * min(cur, N_CU)
            if ntp > 0:
                tp_active += dt
            elif cur < N_CU:
                idle_free += dt * (N_CU - cur)
            if cur > N_CU:
                over += dt
        cur += dw
        ntp += dn
        last = t
    total = N_CU * wall
    scans = sum(a["cu_ms"] for c, a in acc.items() if is_scan(c))
    alone_mean = {}
    if alone:
        d = collections.defaultdict(list)
        for r in alone:
            d[cls(r["name"])].append((r["e"] - r["s"]) / 1e6)
        alone_mean = {c: sum(v[len(v) // 2:]) / len(v[len(v) // 2:]) for c, v in d.items()}  # (second half: past the warm-up)
    out = dict(window_ms=round(wall, 3), forwards=n_fwd, ms_per_forward=round(wall / n_fwd, 4), cu_ms_available_per_forward=round(total / n_fwd, 1),
               scan_cu_ms_per_forward=round(scans / n_fwd, 1), scan_cu_ms_per_forward_capped_at_256=round(scan_cu_ms_capped / n_fwd, 1),
               scan_share_of_chip=round(scan_cu_ms_capped / total, 4),
               time_with_scan_demand_above_256=round(over / wall, 4),
               time_with_a_time_parallel_kernel_running=round(tp_active / wall, 4),
               cu_ms_certainly_idle_per_forward=round(idle_free / n_fwd, 1),
               scan_cu_demand_histogram={f"{k}-{k + 31}": round(v / wall, 4) for k, v in sorted(hist.items())},
               kernels=[])
    for c, a in sorted(acc.items(), key=lambda kv: -kv[1]["dur"]):
        k = dict(kernel=c, dispatches_per_forward=round(a["n"] / n_fwd, 2), mean_ms_in_region=round(a["dur"] / a["n"], 4), workgroups=a["wgs"],
                 sum_ms_per_forward=round(a["dur"] / n_fwd, 4))
        if is_scan(c):
            k["cu_ms_per_forward"] = round(a["cu_ms"] / n_fwd, 1)
        if c in alone_mean:
            k["mean_ms_alone"] = round(alone_mean[c], 4)
            k["slowdown_in_region"] = round(a["dur"] / a["n"] / alone_mean[c], 3)
            if is_scan(c):
                k["cu_ms_per_forward_at_alone_rate"] = round(a["cu_ms"] / n_fwd / k["slowdown_in_region"], 1)
        out["kernels"].append(k)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
