#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS / occupancy table of libceleste_hip.so, from the compiler's own remarks.

    python tools/resource_usage.py            # markdown table of every kernel (what DESIGN.md section 5 embeds)
    python tools/resource_usage.py --json     # the same as JSON
    python tools/resource_usage.py --check    # compare the hot kernels with tools/resource_budget.json, exit 1 on a regression

Runs `make -C desi-mcmc_amd/csrc resource-usage` (hipcc -Rpass-analysis=kernel-resource-usage, ~25 s, no GPU needed) and
parses the remarks.  tests/test_abi_and_host.py::test_hot_kernels_keep_their_resource_budget calls check(): a hot kernel
that gains scratch, spills or loses an occupancy step fails the CPU suite.
"""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "desi-mcmc_amd", "csrc")
BUDGET = os.path.join(ROOT, "tools", "resource_budget.json")
FIELDS = {"TotalSGPRs": "sgpr", "VGPRs": "vgpr", "AGPRs": "agpr", "ScratchSize [bytes/lane]": "scratch", "Occupancy [waves/SIMD]": "occupancy",
          "SGPRs Spill": "sgpr_spill", "VGPRs Spill": "vgpr_spill", "LDS Size [bytes/block]": "lds"}


def demangle(names):
    for tool in ("c++filt", "/opt/rocm/lib/llvm/bin/llvm-cxxfilt"):
        try:
            out = subprocess.run([tool], input="\n".join(names) + "\n", capture_output=True, text=True, check=True).stdout.split("\n")
            return out[:len(names)]
        except Exception:
            continue
    return names


def short(name):
    """k_patch_ll_hw<3, double>(BandDev const*, ...) -> k_patch_ll_hw<3, double>"""
    name = re.sub(r"^void ", "", name)
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def collect():
    """-> {kernel: {vgpr, sgpr, scratch, occupancy, lds, ...}} for every __global__ function of the library"""
    p = subprocess.run(["make", "-C", CSRC, "resource-usage"], capture_output=True, text=True)
    text = p.stderr + p.stdout
    if p.returncode != 0:
        raise RuntimeError("make resource-usage failed:\n" + text[-2000:])
    kernels, cur, order = {}, None, []
    for line in text.split("\n"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = m.group(1)
            kernels[cur] = {}
            order.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+) \[-Rpass-analysis", line)
        if m and cur and m.group(1).strip() in FIELDS:
            v = m.group(2)
            kernels[cur][FIELDS[m.group(1).strip()]] = int(v) if v.lstrip("-").isdigit() else v
    # (template arguments that are types only differ in the signature: keep the mangled tail apart by the full demangled name)
    names = [short(n) for n in demangle(order)]
    return {n: kernels[o] for n, o in zip(names, order)}


def table(k):
    rows = ["| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | spills (V/S) | LDS B/block | waves/SIMD |", "|---|---|---|---|---|---|---|---|"]
    for name in sorted(k):
        r = k[name]
        rows.append("| `%s` | %s | %s | %s | %s | %s / %s | %s | %s |" % (name, r.get("vgpr"), r.get("agpr"), r.get("sgpr"), r.get("scratch"),
                                                                        r.get("vgpr_spill"), r.get("sgpr_spill"), r.get("lds"), r.get("occupancy")))
    return "\n".join(rows)


def check(k=None, budget_path=BUDGET):
    """-> list of regressions (empty: fine).  The budget names, per hot kernel, the most scratch and VGPR spills it may have and
    the fewest waves per SIMD; a kernel of the budget that no longer exists is a regression too (renamed: update the budget)."""
    k = collect() if k is None else k
    budget = json.load(open(budget_path))["kernels"]
    bad = []
    for name, b in budget.items():
        r = k.get(name)
        if r is None:
            bad.append("%s: not in the library any more (renamed? update tools/resource_budget.json)" % name)
            continue
        if r["scratch"] > b["max_scratch"]:
            bad.append("%s: %d B/lane of scratch (budget %d)" % (name, r["scratch"], b["max_scratch"]))
        if r["vgpr_spill"] > b.get("max_vgpr_spill", 0):
            bad.append("%s: %d VGPR spills (budget %d)" % (name, r["vgpr_spill"], b.get("max_vgpr_spill", 0)))
        if r["occupancy"] < b["min_occupancy"]:
            bad.append("%s: %d waves per SIMD by registers (budget %d)" % (name, r["occupancy"], b["min_occupancy"]))
        if "max_lds" in b and r["lds"] > b["max_lds"]:
            bad.append("%s: %d B of static LDS per block (budget %d: one more and a CU holds a wave less)" % (name, r["lds"], b["max_lds"]))
    return bad


if __name__ == "__main__":
    k = collect()
    if "--json" in sys.argv:
        print(json.dumps(k, indent=1, sort_keys=True))
    elif "--check" in sys.argv:
        bad = check(k)
        print("\n".join(bad) if bad else "resource budget: ok (%d hot kernels)" % len(json.load(open(BUDGET))["kernels"]))
        sys.exit(1 if bad else 0)
    else:
        print(table(k))
