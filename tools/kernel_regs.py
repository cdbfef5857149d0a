#!/usr/bin/env python3
"""tools/kernel_regs.py <file.s> ... -- register / spill / LDS census of every kernel in `hipcc -S --cuda-device-only` output, plus how many scalar loads,
lane spills (v_readlane / v_writelane) and scratch accesses its text holds (a loop that re-reads kernel arguments shows up here: round 5, k_ddc_mfma)."""
import re, subprocess, sys
for path in sys.argv[1:]:
    t = open(path).read()
    meta = {m.group(1): m.group(2) for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.wavefront_size", t, re.S)}
    for name, body in meta.items():
        g = lambda k: (re.search(k + r":\s+(\d+)", body) or [0, "0"])[1]
        try:
            dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        except Exception:
            dn = name
        dn = dn.replace("(anonymous namespace)::", "").replace("void ", "")
        dn = re.sub(r"\(.*", "", dn)[:60]
        m = re.search(r"^" + re.escape(name) + r":.*?s_endpgm", t, re.S | re.M)
        txt = m.group(0) if m else ""
        print("%-60s vgpr %3s sgpr %3s s-spill %3s v-spill %3s lds %6s | s_load %3d lane-spill %3d scratch %3d mfma %4d" % (
            dn, g(".vgpr_count"), g(".sgpr_count"), g(".sgpr_spill_count"), g(".vgpr_spill_count"), g(".group_segment_fixed_size"),
            len(re.findall(r"\ts_load_", txt)), len(re.findall(r"v_readlane|v_writelane", txt)), len(re.findall(r"scratch_", txt)), len(re.findall(r"v_mfma", txt))))
