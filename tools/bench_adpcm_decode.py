#!/usr/bin/env python3
"""tools/bench_adpcm_decode.py -- decode_ima_adpcm_u8_i16 at a few (streams, bytes per stream) shapes, with the kernel the library picks, with
CSDR_AMD_ADPCM_SERIAL=1 (one lane per stream) and with CSDR_AMD_ADPCM_SCAN=1 (two scans per stream): the measured points behind adpcm.hip's choice.
One process per setting (the switches are read once per process).  Prints one JSON line per shape."""
import json
import os
import subprocess
import sys
import time

SHAPES = [(65536, 256), (65536, 1024), (8192, 1024), (4096, 100), (2048, 4096), (1024, 65536), (64, 1 << 20)]

if len(sys.argv) > 1 and sys.argv[1] == "--leg":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import numpy as np
    import csdr_amd
    ctx = csdr_amd.Context(0); L = ctx.L
    out = {}
    for S, n in SHAPES:
        rng = np.random.default_rng(S + n)
        x = ctx.upload(rng.integers(0, 256, (S, n), dtype=np.uint8)); y = ctx.alloc(4 * S * n + 64); st = ctx.upload(np.zeros(2 * S, np.int32))
        for _ in range(3): L.csdr_amd_decode_ima_adpcm_u8_i16(ctx.h, x.ptr, y.ptr, S, n, n, 2 * n, st.ptr)
        ctx.sync(); t0 = time.perf_counter()
        reps = 20
        for _ in range(reps): L.csdr_amd_decode_ima_adpcm_u8_i16(ctx.h, x.ptr, y.ptr, S, n, n, 2 * n, st.ptr)
        ctx.sync(); out["%dx%d" % (S, n)] = round((time.perf_counter() - t0) / reps * 1e3, 4)
    print(json.dumps(out))
    sys.exit(0)

res = {}
for name, env in (("auto", {}), ("serial", {"CSDR_AMD_ADPCM_SERIAL": "1"}), ("scan", {"CSDR_AMD_ADPCM_SCAN": "1"})):
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg"], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
    res[name] = json.loads(p.stdout.decode().strip().splitlines()[-1])
for S, n in SHAPES:
    k = "%dx%d" % (S, n)
    print(json.dumps({"op": "decode_ima_adpcm_u8_i16", "streams": S, "bytes_per_stream": n, "ms_auto": res["auto"][k], "ms_serial": res["serial"][k], "ms_scan": res["scan"][k]}))
