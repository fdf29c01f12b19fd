#!/usr/bin/env python3
"""Build-container check (reads /root/reference; not a test, not shipped): the share of a product
file's code lines that also occur, whitespace-normalised, in the reference tree's Python/Cython/C.
    python tools/line_overlap.py [files...]       default: every .py under desi-mcmc_amd/
Lines shorter than 12 characters, imports, decorators and bare returns are not counted."""
import glob
import os
import re
import sys

REF = "/root/reference"


def norm_lines(path):
    out = []
    for ln in open(path, errors="replace"):
        ln = re.sub(r"#.*$", "", ln)
        ln = re.sub(r"\s+", "", ln)
        if len(ln) < 12 or ln.startswith(("import", "from", "@", "return", '"""', "'''")):
            continue
        out.append(ln)
    return out


def main():
    ref = set()
    for ext in ("py", "pyx", "c", "h"):
        for p in glob.glob(os.path.join(REF, "**", "*." + ext), recursive=True):
            ref.update(norm_lines(p))
    files = sys.argv[1:] or sorted(glob.glob("desi-mcmc_amd/**/*.py", recursive=True))
    for f in files:
        ls = norm_lines(f)
        hit = [l for l in ls if l in ref]
        print("%-55s %4d / %4d = %.2f" % (f, len(hit), len(ls), len(hit) / max(len(ls), 1)))
        if len(sys.argv) > 1 and os.environ.get("SHOW"):
            for l in hit:
                print("    ", l[:110])


if __name__ == "__main__":
    main()
