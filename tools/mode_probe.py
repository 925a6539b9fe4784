#!/usr/bin/env python3
"""k_decode_wide at 1080p x 2048 has (at least) two speeds -- about 2.35 and 2.6-2.85 ms -- from process to process on one box
(profiles/r06_1080p_placement.txt, r06_block_order_s2.txt).  Per-launch durations inside each process and the board's state
(rocm-smi) beside them, for a few workgroup orders, round robin.

    python tools/mode_probe.py [--procs 6] > profiles/r06_decode_wide_modes.txt
"""
import argparse
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, default=6)
    args = ap.parse_args()
    orders = [int(x) for x in os.environ.get("MODE_ORDERS", "1,8,64,256,-1").split(",")]
    print("k_decode_wide<SysNTSC, 16>, 1920x1080 x 2048, per process: mean of 10 launches, then the 10 launches one by one (synchronised), rocm-smi after")
    for r in range(args.procs):
        for k in orders:
            env = dict(os.environ, CRTHIP_WIDE_ORDER=str(k), SWEEP_SERIES="1")
            try:
                out = subprocess.run([sys.executable, os.path.join(HERE, "placement_sweep.py"), "--child", "0"], env=env, stdout=subprocess.PIPE,
                                     stderr=subprocess.DEVNULL, timeout=300).stdout.decode()
                j = json.loads(out.strip().splitlines()[-1])
                print("round %d order %4d | decode %.3f active %.3f fieldpass %.3f | %s | %s | %s" % (
                    r, k, j["decode_ms"], j["active_ms"], j["fieldpass_ms"], " ".join("%.2f" % v for v in j["series"]["decode"]),
                    json.dumps(j.get("ptrs")), json.dumps(j.get("under_load"))))
            except Exception as ex:                               # noqa: BLE001
                print("round %d order %d failed: %s" % (r, k, ex))
            sys.stdout.flush()


if __name__ == "__main__":
    main()
