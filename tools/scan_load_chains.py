#!/usr/bin/env python3
"""tools/scan_load_chains.py <file.s ...> -- finds the disease round 6 met twice (the FFT filter's spectrum and twiddle loads): a kernel whose registers are all taken gets its
global loads as load / s_waitcnt vmcnt(0) / use chains -- one memory round trip after the other.  Prints, per kernel of the given ISA listings (hipcc -S --cuda-device-only),
the longest run of load-wait pairs and how many runs of four or more there are.  Round 6's scan of every .hip of the library: none in a hot production kernel
(k_ddc_fwd512's 45-88 pairs were un-chained and re-measured: no difference, its 25 us are one generation of workgroups; the others are serial-by-nature or fallback kernels).
   for f in csdr_amd/csrc/*.hip; do hipcc -O3 -std=c++17 -ffp-contract=off --offload-arch=gfx950 -S --cuda-device-only $f -o /tmp/$(basename $f).s; done; python tools/scan_load_chains.py /tmp/*.s"""
import re, subprocess, sys
for path in sys.argv[1:]:
    s = open(path).read().split("\n")
    funcs = [(i, l.split(":")[0]) for i, l in enumerate(s) if l.startswith("_Z") and ": " in l]
    for fi, (start, name) in enumerate(funcs):
        end = funcs[fi + 1][0] if fi + 1 < len(funcs) else len(s)
        ev = []
        for l in s[start:end]:
            l = l.strip()
            if not l or l.startswith((";", ".", "//")): continue
            op = l.split()[0]
            if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")) and "lds" not in l: ev.append("L")
            elif op == "s_waitcnt" and "vmcnt(0)" in l: ev.append("W")
            else: ev.append(".")
        runs = re.findall(r"(?:L\.?W\.?){4,}", re.sub(r"\.+", ".", "".join(ev)))
        if runs:
            dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()[:120]
            print("%s: %3d load-wait pairs in the longest run, %d runs: %s" % (path.split("/")[-1], max(r.count("L") for r in runs), len(runs), dn))
