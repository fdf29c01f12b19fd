#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>.json (per-dispatch counter lists written by tools/pmc_pass.sh) -> the summary kept
under profiles/: per kernel and counter {launches, mean, last, max}.
    python tools/pmc_summarise.py r02_final "<the command that produced it>"  > profiles/r02_final_pmc.json"""
import hashlib
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, command = sys.argv[1], sys.argv[2]
raw = json.load(open(os.path.join(root, "gpurun_out", "pmc_%s.json" % tag)))
out = {
    # bench.py reports these counters only while the library it loads is the one they were taken with
    "library_sha256": hashlib.sha256(open(os.path.join(root, "desi-mcmc_amd", "libceleste_hip.so"), "rb").read()).hexdigest(),
    "command": command + "  (one rocprofv3 --kernel-trace --pmc pass per counter group; tools/profile_r06.sh)",
    "units": "FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them; on gfx950 FETCH_SIZE counts half of the bytes of a "
             "streaming read (MI355X_MICROARCH.md): HBM bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.  SQ_* cycle counters "
             "in quad-cycles.  \"last\" = the last launch (a timed-region step); \"mean\" includes warm-up and untimed launches "
             "(the first renders of a run have no measured tile order yet).",
    "kernels": {k: {c: {"launches": len(v), "mean": sum(v) / len(v), "last": v[-1], "max": max(v)} for c, v in cs.items() if v}
                for k, cs in raw.items()},
}
json.dump(out, sys.stdout, indent=1)
