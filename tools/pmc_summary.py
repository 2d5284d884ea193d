#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (tools/collect_pmc.sh) into the JSON bench.py reads.

  python3 tools/pmc_summary.py gpurun_out/r02/pmc_c3 --tag c3 -o profiles/r02_pmc_c3.json

Per kernel (short name = text before the argument list): number of dispatches seen and the MEAN per dispatch of
every counter; FETCH_SIZE / WRITE_SIZE are reported in bytes (rocprofv3 gives KB), and ``hbm_bytes`` applies the
gfx950 correction of MI355X_MICROARCH.md "HBM": 2 x FETCH_SIZE + WRITE_SIZE.  The summary is tagged with the
hash of the kernel sources (``src_hash``: gpflowpilco_amd/csrc/*.hip, *.h, include/*.h) so that bench.py can tell
whether the counters describe the code it is running.
"""
import argparse
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def src_hash(root=ROOT):
  h = hashlib.sha256()
  files = sorted(glob.glob(os.path.join(root, "gpflowpilco_amd", "csrc", "*.hip"))
                 + glob.glob(os.path.join(root, "gpflowpilco_amd", "csrc", "*.h"))
                 + glob.glob(os.path.join(root, "include", "*.h")))
  for f in files:
    h.update(os.path.basename(f).encode())
    with open(f, "rb") as fh:
      h.update(fh.read())
  return h.hexdigest()[:16]


def short_name(full: str) -> str:
  s = re.sub(r"^void\s+", "", full)
  depth = 0
  for i, ch in enumerate(s):            # cut at the '(' of the argument list (not inside template brackets)
    if ch == "<":
      depth += 1
    elif ch == ">":
      depth -= 1
    elif ch == "(" and depth == 0:
      return s[:i].strip()
  return s.strip()


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("dir")
  ap.add_argument("--tag", required=True)
  ap.add_argument("-o", "--out", required=True)
  ap.add_argument("--bench-args", default="")
  ap.add_argument("--include", default=r"^(k_|mm_)", help="regex on the short kernel name")
  args = ap.parse_args()
  sums = defaultdict(lambda: defaultdict(float))
  cnts = defaultdict(lambda: defaultdict(int))
  meta = {}
  files = sorted(glob.glob(os.path.join(args.dir, "pass*", "**", "*counter_collection.csv"), recursive=True))
  if not files:
    sys.exit(f"no counter_collection.csv under {args.dir}")
  inc = re.compile(args.include)
  for f in files:
    with open(f, newline="") as fh:
      for row in csv.DictReader(fh):
        k = short_name(row["Kernel_Name"])
        if not inc.search(k):
          continue
        c = row["Counter_Name"]
        sums[k][c] += float(row["Counter_Value"])
        cnts[k][c] += 1
        meta.setdefault(k, {"vgpr": int(row["VGPR_Count"]), "agpr": int(row["Accum_VGPR_Count"]),
                            "sgpr": int(row["SGPR_Count"]), "lds_bytes": int(row["LDS_Block_Size"]),
                            "scratch_bytes": int(row["Scratch_Size"]), "workgroup": int(row["Workgroup_Size"])})
        # dispatch duration under the counter pass (ns); informational only (counter passes run slower)
        meta[k].setdefault("_dur", []).append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
  kernels = {}
  for k in sorted(sums):
    ent = dict(meta[k])
    durs = ent.pop("_dur")
    ent["dur_us_under_pmc"] = round(sum(durs) / len(durs) / 1e3, 2)
    ent["dispatches"] = max(cnts[k].values())
    cm = {c: sums[k][c] / cnts[k][c] for c in sums[k]}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
      if c in cm:
        cm[c] = cm[c] * 1024.0          # KB -> bytes
    if "FETCH_SIZE" in cm and "WRITE_SIZE" in cm:
      cm["hbm_bytes"] = 2.0 * cm["FETCH_SIZE"] + cm["WRITE_SIZE"]
    ent["counters"] = {c: round(v, 1) for c, v in sorted(cm.items())}
    kernels[k] = ent
  out = {"tag": args.tag, "src_hash": src_hash(), "bench_args": args.bench_args,
         "collected_with": "tools/collect_pmc.sh (rocprofv3 --pmc, one pass per counter group)",
         "units": {"FETCH_SIZE": "bytes as reported (x2 = bytes read on gfx950)", "WRITE_SIZE": "bytes",
                   "hbm_bytes": "2*FETCH_SIZE + WRITE_SIZE", "SQ_*": "summed over the chip, mean per dispatch"},
         "kernels": kernels}
  os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
  with open(args.out, "w") as fh:
    json.dump(out, fh, indent=1, sort_keys=True)
  print(f"{args.out}: {len(kernels)} kernels, src_hash {out['src_hash']}")


if __name__ == "__main__":
  main()
