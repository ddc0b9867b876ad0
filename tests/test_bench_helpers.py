"""bench.py's host-side helpers that need no GPU: the one-line stdout contract and the committed gather ceiling."""
import json
import os
import subprocess
import sys

import helpers as H


def test_stdout_carries_one_json_line_whatever_libraries_print():
    """native libraries write to fd 1 too ("[Gloo] Rank 0 is connected ..." went out ahead of the line in round 5).
    claim_stdout() points fd 1 at stderr for the life of the process; emit() writes the line to the descriptor that WAS stdout."""
    code = ("import os, sys; sys.path.insert(0, %r); import bench; bench.claim_stdout(); "
            "os.write(1, b'[Gloo] noise from a native library\\n'); print('python-level noise'); "
            "bench.emit({'metric': 'x', 'value': 1.5})") % H.ROOT
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    lines = p.stdout.split("\n")
    assert len(lines) == 2 and lines[1] == "" and json.loads(lines[0]) == {"metric": "x", "value": 1.5}
    assert "noise from a native library" in p.stderr and "python-level noise" in p.stderr


def test_gather_ceiling_is_read_for_the_targets_footprint():
    """roofline.frac_of_gather_ceiling: the ceiling comes from the committed microbenchmark table, picked by the footprint of the
    target's half-overlapping 2-bit blocks (tlen / 2 bytes): 25 MB -> the 32 MiB row, 100 MB -> the 128 MiB row"""
    sys.path.insert(0, H.ROOT)
    import bench
    c50, c200 = bench.gather_ceiling(50_000_000 / 2.0), bench.gather_ceiling(200_000_000 / 2.0)
    assert c50 is not None and c200 is not None
    assert c50[1] == 32 and c200[1] == 128
    assert 3000.0 < c200[0] < c50[0] < 8000.0               # random lines: below the streaming peak, lower from HBM than from the memory-side cache
    tab = json.load(open(os.path.join(H.ROOT, "profiles", "gather_ceiling.json")))
    assert [r["mib"] for r in tab["regions"]] == sorted(r["mib"] for r in tab["regions"])
