#!/usr/bin/env python3
"""DESIGN.md section 5's numbers, generated from a bench.py line so that the prose cannot drift from the record:

    python tools/design_numbers.py profiles/r05_bench_line.json            # print the block
    python tools/design_numbers.py profiles/r05_bench_line.json --write    # replace the block between the markers in DESIGN.md and README.md

The input is the ONE JSON line `python bench.py --gpus 1 --steps 20 --warmup 5` printed (the driver's BENCH_rNN.json holds the same line
under "parsed" / in "tail": pass that file and the line is dug out).  Every row names the key it was read from."""
import argparse
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- numbers: begin", "<!-- numbers: end -->"


def load_line(path):
    txt = open(path).read()
    try:
        d = json.loads(txt)
        if "trailer" in d:
            return d
        if isinstance(d.get("tail"), str):  # the driver's record: the line is inside "tail"
            txt = d["tail"]
    except ValueError:
        pass
    for ln in reversed(txt.splitlines()):
        ln = ln.strip()
        if ln.startswith("{") and '"trailer"' in ln:
            return json.loads(ln)
    raise SystemExit("no bench line in %s" % path)


def f(x, nd=0):
    if x is None:
        return "n/a"
    return ("{:,.%df}" % nd).format(x).replace(",", " ")


def block(d, src):
    h2h = d.get("host_to_host") or {}
    pin, pag = h2h.get("pinned") or {}, h2h.get("pageable") or {}
    lat = d.get("latency_ms_batch1") or {}
    lath = h2h.get("latency_ms_batch1_host") or {}
    rf, rv = d.get("roofline") or {}, d.get("roofline_valu") or {}
    cfg = d.get("configs") or {}
    hs = d.get("host_share") or {}
    ceil = h2h.get("pcie_ceiling_GBps") or {}
    prof = rf.get("profile") or {}
    rows = []
    add = lambda what, val, key: rows.append("| %s | %s | `%s` |" % (what, val, key))
    add("stereo pairs/s, inputs and outputs in HBM (the contract's `value`; KITTI 1242x375, D=128, 256 pairs per step)", "**%s**" % f(d.get("value")), "value")
    add("ms per 256-pair step / timed seconds / steps", "%s / %s / %s" % (f(d.get("ms_per_step"), 3), f(d.get("timed_seconds"), 2), d.get("steps")), "ms_per_step, timed_seconds, steps")
    add("host CPU cores busy during the timed region", f(d.get("host_cpu_cores_busy"), 2), "host_cpu_cores_busy")
    add("SURVEY 8(d)'s pair - page-locked host memory in, f32 D1 back in host memory (PCIe inclusive), median of three", "**%s** (runs %s; spread %s)" % (
        f(pin.get("pairs_per_s_d1")), ", ".join(f(x) for x in pin.get("pairs_per_s_d1_runs", [])), f(pin.get("pairs_per_s_d1_spread"), 3)), "host_to_host.pinned.pairs_per_s_d1")
    add("... of the link's own rate for that download (%s GB/s per direction with both busy = %s pairs/s)" % (f(ceil.get("each_direction_when_both_run"), 1), f(pin.get("link_ceiling_pairs_per_s_d1"))),
        f(pin.get("d1_over_link_ceiling"), 3), "host_to_host.pinned.d1_over_link_ceiling")
    add("... with the driver's 8-bit disparity image as the output", "**%s** (runs %s)" % (f(pin.get("pairs_per_s_dmap_u8")), ", ".join(f(x) for x in pin.get("pairs_per_s_dmap_u8_runs", []))),
        "host_to_host.pinned.pairs_per_s_dmap_u8")
    add("... D1 and D2", f(pin.get("pairs_per_s_d1_d2")), "host_to_host.pinned.pairs_per_s_d1_d2")
    add("... pageable caller memory: D1 / 8-bit", "%s / %s" % (f(pag.get("pairs_per_s_d1")), f(pag.get("pairs_per_s_dmap_u8"))), "host_to_host.pageable")
    add("who moves the chunks over PCIe", str(h2h.get("copies")), "host_to_host.copies")
    add("ms/frame, one pair per call, device memory: median / p99", "**%s** / %s" % (f(lat.get("median"), 3), f(lat.get("p99"), 3)), "latency_ms_batch1")
    paced = lat.get("at_30_frames_per_s") or {}
    if paced:
        add("... one frame every 33 ms (a camera's pace; the helper threads sleep in between): median / p99", "%s / %s" % (f(paced.get("median"), 3), f(paced.get("p99"), 3)), "latency_ms_batch1.at_30_frames_per_s")
    add("ms/frame through `sv_elas_process` (host pointers): page-locked / pageable, median", "%s / %s" % (f((lath.get("pinned") or {}).get("median"), 3), f((lath.get("pageable") or {}).get("median"), 3)),
        "host_to_host.latency_ms_batch1_host")
    add("the seven committed real frames cycled through the batch", f((d.get("value_real_pair") or {}).get("value")), "value_real_pair.value")
    add("one rank under an 8-rank host budget (2 CPUs): pairs/s / of `value` / cores busy", "%s / %s / %s" % (f(hs.get("value")), f(hs.get("ratio_to_value"), 3), f(hs.get("host_cpu_cores_busy"), 2)), "host_share")
    for name, what in (("kitti_d256", "KITTI, D=256 (LDS-pressure config)"), ("4k_d192", "4K 3840x2160, D=192")):
        c = cfg.get(name) or {}
        add(what, f(c.get("pairs_per_s")), "configs.%s.pairs_per_s" % name)
    add("`roofline` kernel; achieved GB/s; fraction of 8 TB/s (8(d) bytes / HIP-event duration of its launches in the timed region, %s us per %s-pair launch)" % (
        f(rf.get("avg_launch_us"), 1), f(rf.get("pairs_per_launch"))), "%s; %s; **%s**" % (rf.get("kernel"), f(rf.get("achieved"), 1), f(rf.get("frac"), 4)), "roofline")
    ser = rf.get("serial") or {}
    add("... without kernel overlap (one slot, one stream: %s us per %s-pair launch)" % (f(ser.get("avg_launch_us"), 1), f(ser.get("pairs_per_launch"))), "%s GB/s = %s" % (f(ser.get("achieved"), 1), f(ser.get("frac"), 4)),
        "roofline.serial")
    for kind in ("pipelined", "serial"):
        p = prof.get(kind) or {}
        if isinstance(p, dict) and p.get("avg_launch_us"):
            alg = rf.get("algorithmic_bytes_per_pair", 0) * (p.get("pairs_per_launch") or 0)
            fr = p.get("frac") or alg / (p["avg_launch_us"] * 1e-6) / 1e9 / rf.get("peak", 8000.0)
            add("... from the committed rocprofv3 summary `%s` (average %s us%s per %s-pair launch)" % (p.get("file"), f(p.get("avg_launch_us"), 1),
                                                                                                   (", median %s" % f(p["median_launch_us"], 1)) if p.get("median_launch_us") else "", p.get("pairs_per_launch")),
                f(fr, 4), "roofline.profile.%s" % kind)
    if rf.get("traffic") and rf.get("algorithmic_bytes_per_launch"):
        add("counter traffic of that kernel per launch / its 8(d) bytes", "%s MB / %s MB = %s x" % (f(rf["traffic"] / 1e6, 1), f(rf["algorithmic_bytes_per_launch"] / 1e6, 1),
                                                                                          f(rf["traffic"] / rf["algorithmic_bytes_per_launch"], 2)), "roofline.traffic")
    wp = rf.get("whole_pipeline") or {}
    add("whole pipeline: 88 N bytes x pairs/s over 8 TB/s", f(wp.get("frac"), 3), "roofline.whole_pipeline.frac")
    for kname in ("dense_match", "support_match"):
        kk = (rv.get("kernels") or {}).get(kname) or {}
        add("`%s`: wave64 VALU instructions per pair / issue floor at 4 cycles / measured us per pair / SAD byte rate of the 157 T/s peak" % kname,
            "%s M / %s us / %s us / %s" % (f((kk.get("wave_insts_per_pair") or 0) / 1e6, 2), f(kk.get("issue_floor_us_per_pair_at_4_cycles"), 2), f(kk.get("serial_us_per_pair"), 2), f(kk.get("sad_frac_of_peak"), 3)),
            "roofline_valu.kernels.%s" % kname)
    cb, ca = d.get("cpu_baseline") or {}, d.get("cpu_baseline_all_cores") or {}
    add("the reference's own `Elas::process` (`oracle/_ref`) on the GPU host: 1 thread / %s processes" % ca.get("cores"), "%s / %s pairs/s" % (f(cb.get("value"), 1), f(ca.get("value"), 1)), "cpu_baseline, cpu_baseline_all_cores")
    add("parity gate before / after the timed region", "%s / %s" % (d.get("parity_gate"), (d.get("parity_after") or {}).get("status") if isinstance(d.get("parity_after"), dict) else d.get("parity_after")),
        "parity_gate, parity_after")
    head = "%s generated by tools/design_numbers.py from %s; do not edit by hand -->\n" % (BEGIN, src)
    return head + "| quantity | value | key on the bench line |\n|---|---|---|\n" + "\n".join(rows) + "\n" + END


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("line")
    ap.add_argument("--write", action="store_true")
    a = ap.parse_args()
    d = load_line(a.line)
    b = block(d, os.path.relpath(os.path.abspath(a.line), ROOT))
    if not a.write:
        print(b)
        return
    for name in ("DESIGN.md", "README.md"):
        path = os.path.join(ROOT, name)
        txt = open(path).read()
        if BEGIN not in txt:
            print("%s: no marker block" % name)
            continue
        new = re.sub(re.escape(BEGIN) + r".*?" + re.escape(END), lambda m: b, txt, flags=re.S)
        open(path, "w").write(new)
        print("%s: block replaced" % name)


if __name__ == "__main__":
    main()
