#!/usr/bin/env python3
"""Where a chunk of the host-memory path spends its time: reads the file the engine writes with SV_CHUNK_TRACE=<file> (wall-clock time
of every pipeline stage of every chunk) and prints, per stage transition, the mean / median / maximum, plus how busy the two DMA
engines were (a copy runs from max(enqueued, previous copy done) to done).

    SV_CHUNK_TRACE=/tmp/t.txt python tools/h2h_time.py --reps 1 && python tools/chunk_trace.py /tmp/t.txt
"""
import sys
from collections import defaultdict

import numpy as np

ORDER = ["slot", "upload_queued", "upload_done", "phase1_queued", "phase1_done", "host_done", "phase2_queued", "phase2_done", "download_queued", "download_done", "free"]


def main(path, skip=16):
    per_slot = defaultdict(list)
    for line in open(path):
        if line.startswith("#"):
            per_slot = defaultdict(list)  # (the last handle of the file)
            continue
        slot, stage, ns = line.split()
        per_slot[int(slot)].append((int(ns), stage))
    chunks = []
    for slot, recs in per_slot.items():
        recs.sort()
        cur = None
        for ns, stage in recs:
            if stage == "slot":
                cur = {}
                chunks.append(cur)
            if cur is not None:
                cur[stage] = ns
    chunks = [c for c in chunks if "free" in c]
    chunks.sort(key=lambda c: c["slot"])
    chunks = chunks[skip:]
    if not chunks:
        print("no complete chunks")
        return
    span = (max(c["free"] for c in chunks) - min(c["slot"] for c in chunks)) * 1e-6
    print("%d chunks over %.1f ms: %.3f ms per chunk" % (len(chunks), span, span / len(chunks)))
    for a, b in zip(ORDER[:-1], ORDER[1:]):
        d = np.array([(c[b] - c[a]) * 1e-6 for c in chunks if a in c and b in c])
        if d.size:
            print("  %-16s -> %-16s mean %7.3f  median %7.3f  max %7.3f ms" % (a, b, d.mean(), np.median(d), d.max()))
    tot = np.array([(c["free"] - c["slot"]) * 1e-6 for c in chunks])
    print("  slot -> free: mean %.3f ms (slots in flight on average: %.1f)" % (tot.mean(), tot.sum() / span))
    for name, q, done in (("upload", "upload_queued", "upload_done"), ("download", "download_queued", "download_done")):
        cs = sorted((c for c in chunks if q in c and done in c), key=lambda c: c[done])
        busy, prev = 0.0, None
        for c in cs:
            start = c[q] if prev is None else max(c[q], prev)
            busy += max(0, c[done] - start)
            prev = c[done]
        if cs:
            print("  %s engine busy at most %.0f %% of the span (as seen by the host: a copy runs from max(queued, previous done) to done)" % (name, 100.0 * busy * 1e-6 / span))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 16)
