"""Shared pieces of the bench scripts (bench.py = the driver's contract line for BASELINE config 2; bench_fir.py, bench_fftfilt.py,
bench_fastddc.py, bench_nfm.py = the same contract for configs 1, 3, 4, 5): the CPU-baseline leg (the unmodified reference timed in
process on this box's host cores, oracle/cpu_bench.c), the effective core count, and the look-up of committed rocprofv3 PMC summaries."""
import glob
import json
import math
import os
import subprocess

ROOT = os.path.dirname(os.path.abspath(__file__))
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)
FP32_PEAK_TFLOPS = 157.3       # MI355X_MICROARCH.md: FP32 vector = FP32-input MFMA peak


def effective_cores():
    """Host threads this process can actually run concurrently: min(CPU affinity, cgroup CPU quota)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            q, per = open(path).read().split()
            if q != "max":
                quota = float(q) / float(per)
        except Exception:  # noqa: BLE001
            pass
    if quota is None:
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except Exception:  # noqa: BLE001
            pass
    eff = aff if quota is None else max(1, min(aff, int(math.floor(quota + 1e-9))))
    return eff, aff, quota


def cpu_bench_exe(fast_fft=False):
    ref = os.path.join(ROOT, "oracle", "_ref", "cpu_bench_ref")
    mkl = os.path.join(ROOT, "oracle", "_ref", "cpu_bench_ref_mkl")
    port = os.path.join(ROOT, "oracle", "cpu_bench_port")
    if fast_fft and os.path.exists(mkl) and os.path.exists("/opt/conda/lib/libmkl_rt.so.1"):
        return mkl, "MKL FFTW3 interface (/opt/conda/lib/libmkl_rt.so.1)"
    if os.path.exists(ref):
        return ref, "oracle/fftw_shim.c (double-precision radix 2)"
    if os.path.exists(port):
        return port, None
    return None, None


def run_cpu_bench(threads, amount, mode="wfm", params=(), fast_fft=False, timeout=600):
    exe, fftprov = cpu_bench_exe(fast_fft)
    if exe is None:
        return None
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1")
    cmd = [exe, str(threads), str(amount)] + ([mode] + [str(p) for p in params] if mode != "wfm" or params else [])
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, env=env)
    out = json.loads(r.stdout.strip().splitlines()[-1])
    out["fft_provider"] = fftprov
    return out


def cpu_baseline(mode, params=(), unit="complex MS/s", single_amount=None, target_wall_s=15.0, probe_amount=None, fast_fft=False, describe=""):
    """Reference CPU path on the host cores, on a bounded sample (about target_wall_s of wall): one thread alone, then `effective cores`
    threads (one independent stream / channel group per thread).  `cores` = the threads actually used."""
    eff, aff, quota = effective_cores()
    try:
        probe_amount = probe_amount if probe_amount is not None else single_amount
        one = run_cpu_bench(1, single_amount, mode, params, fast_fft)
        if one is None:
            return None
        if "error" in one:
            return one
        # grow the all-cores sample until it takes about target_wall_s of wall (thread start-up and planning dominate tiny samples)
        amount = probe_amount
        allc = run_cpu_bench(eff, amount, mode, params, fast_fft)
        for _ in range(3):
            if allc["wall_s"] >= 0.4 * target_wall_s:
                break
            amount = amount * min(max(target_wall_s / max(allc["wall_s"], 1e-3), 1.5), 50.0)
            amount = int(amount) if amount >= 4 else round(amount, 2)
            allc = run_cpu_bench(eff, amount, mode, params, fast_fft)
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}
    return {"value": round(allc["msps"], 3), "unit": unit, "cores": eff, "kind": one["kind"],
            "sample": "%s: %d threads, %g %s (%.3e samples, %.1f s wall); 1 thread alone: %.2f MS/s (%.1f s wall)"
                      % (describe or mode, eff, allc["amount"], allc["amount_unit"], allc["samples"], allc["wall_s"], one["msps"], one["wall_s"]),
            "single_core_value": round(one["msps"], 3), "cpu_affinity": aff, "cgroup_cpu_quota_cores": quota,
            "fft_provider": allc.get("fft_provider")}


def pmc_traffic(kernel_name, match):
    """HBM-side bytes per launch of a kernel from the committed rocprofv3 PMC summaries (profiles/*_pmc_traffic.json, written by
    tools/pmc_summary.py from separate --pmc FETCH_SIZE / WRITE_SIZE passes of the same command).  A bench cannot read PMCs itself;
    the value is reported only when a summary exists for the same kernel and a workload dict containing every key/value of `match`."""
    import re
    best = None; best_round = -1
    squash = lambda t: str(t).split(" (")[0].replace(" ", "")
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:  # noqa: BLE001
            continue
        w = d.get("workload", {})
        m = re.match(r"r(\d+)", os.path.basename(f))
        rnd = int(m.group(1)) if m else 0
        # the SAME kernel instance (k_fftfilt_lds<4096> is not k_fftfilt_lds<8192>; a summary over several kernels -- "a + b" -- never stands for one of them), the
        # same workload, and the NEWEST round's file: a line must not pick up an older round's figure when a newer pass exists
        kn = squash(kernel_name)
        parts = [squash(p.strip()) for p in str(d.get("kernel", "")).split(" + ") if p.strip()]
        bases = {p.split("<")[0] for p in parts}
        if not parts or (len(parts) > 1 and len(bases) < len(parts)):      # a sum over template instances of ONE kernel (a sweep): never one launch's traffic
            continue
        same = all(p == kn or ("<" not in kn and (p.split("<")[0] == kn or (len(parts) > 1 and p.startswith(kn)))) for p in parts)
        if same and all(w.get(key) == v for key, v in match.items()) and rnd >= best_round:
            best_round = rnd
            best = (d["traffic_bytes_per_launch"], "profiles/" + os.path.basename(f) + " (rocprofv3 PMC passes, FETCH_SIZE x2 + WRITE_SIZE, bytes per launch)")
    return best


def spinup(step, sync, ms, chunk=8, cap=2048):
    """Untimed steps for about `ms` milliseconds before a bench's warm-up: an idle GPU takes ~25 ms of back-to-back launches to reach its running clocks
    (profiles/r3_notes.md: the first launches of a 0.9-ms kernel measure up to 25 % slow), and every metric here is a stream's steady-state rate.
    Returns the number of steps run (reported in the JSON line as "spinup_steps_before_warmup")."""
    import time
    n = 0
    if ms <= 0:
        return 0
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms and n < cap:
        for _ in range(chunk):
            step()
        n += chunk
        sync()
    return n


SPINUP_HELP = ("untimed steps for about this many ms BEFORE the W warm-up steps (GPU clock ramp after idle, see bench_common.spinup); 0 = off")
