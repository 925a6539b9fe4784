#!/usr/bin/env python3
"""Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 per access pattern.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir>/F -- tools/ubench_hbm.bin calib
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d <dir>/W -- tools/ubench_hbm.bin calib
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d <dir>/E -- tools/ubench_enc640.bin 4096 calib
    tools/calib_pmc.py <dir>  > profiles/r03_pmc_calibration.json

Every calibration kernel moves a known number of useful bytes (tools/ubench_hbm.hip, `calib`); the factor is
useful bytes / (counter x 1024)."""
import collections, csv, glob, json, sys
GIB = 1 << 30
KNOWN = {  # kernel name prefix -> (counter, useful bytes, pattern)
    "void k_read<1, 4>": ("FETCH_SIZE", GIB, "16 B/lane contiguous, nontemporal loads"),
    "void k_read<0, 4>": ("FETCH_SIZE", GIB, "16 B/lane contiguous, plain loads"),
    "k_read_pieces64": ("FETCH_SIZE", GIB, "64-byte pieces (4 lanes x 16 B), one per 256 B"),
    "k_read_scatter16": ("FETCH_SIZE", GIB // 4, "scattered 16-byte windows, one per 256 B"),
    "void k_fill<1, 4>": ("WRITE_SIZE", GIB, "16 B/lane contiguous, nontemporal stores"),
    "void k_fill<0, 4>": ("WRITE_SIZE", GIB, "16 B/lane contiguous, plain stores"),
    "k_write_pieces64(": ("WRITE_SIZE", GIB, "64-byte aligned pieces, one per 256 B"),
    "k_write_pieces64_unaligned": ("WRITE_SIZE", GIB, "64-byte pieces at byte offset 28, one per 256 B"),
    # the encoder's image reads as the encoder issues them (tools/ubench_enc.hip -DGEO640, `<fields> calib`, 4096 fields of
    # 640x480: 240 rows x 2560 B each).  VERDICT round 4, weak #4: isolated 64-byte pieces count in full, but k_active's
    # pieces are HALF LINES whose other half follows a tile later -- what the memory side is asked for is the line
    "void k_rows64<64, 64>": ("FETCH_SIZE", 4096 * 240 * 2560, "k_active, narrow image tile: 64-byte half lines, the other half a tile later"),
    "void k_rows64<64, 128>": ("FETCH_SIZE", 4096 * 240 * 2560, "k_active, wide image tile: whole 128-byte lines, nontemporal"),
}
vals = collections.defaultdict(dict)
for f in glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        vals[r["Kernel_Name"]][r["Counter_Name"]] = float(r["Counter_Value"])
out = {}
for name, d in vals.items():
    for pre, (ctr, useful, what) in KNOWN.items():
        if (name.startswith(pre) or (not pre.startswith("void") and pre in name)) and ctr in d:
            kb = d[ctr]
            out[pre.strip("(")] = {"pattern": what, "counter": ctr, "counter_KB": kb, "useful_bytes": useful,
                                   "bytes_per_counter_KB": useful / kb if kb else None,
                                   "factor_vs_1024": useful / (kb * 1024) if kb else None}
json.dump({"method": "tools/ubench_hbm.bin calib under rocprofv3 --pmc (separate passes)", "patterns": out}, sys.stdout, indent=1)
print()
