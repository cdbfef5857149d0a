"""Evidence hygiene: the measured lines committed under profiles/ must agree with the PMC summaries they cite.
For the NEWEST round that has measured lines (profiles/rN_*_n1*.json):
  * a line that reports roofline.traffic cites a summary OF THE SAME ROUND (rN_*_pmc_traffic.json), the file exists, and its bytes per launch are the line's;
  * a traffic / algorithmic ratio above 1.2 must not be contradicted by that summary (and a summary over several template instances of one kernel never
    stands for one launch);
  * the kernel the summary is about is the kernel the line is about.
Older rounds' files are history and are not touched (round 4's n1 lines were generated BEFORE that round's PMC pass and cite round 3's summaries: the reason
this test exists; tools/profile_bench.sh now generates the line last)."""
import glob
import json
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def _round_of(name):
    m = re.match(r"r(\d+)", os.path.basename(name))
    return int(m.group(1)) if m else -1


def _lines():
    files = [f for f in glob.glob(os.path.join(PROF, "r*_n1*.json"))]
    if not files:
        return -1, []
    newest = max(_round_of(f) for f in files)
    return newest, sorted(f for f in files if _round_of(f) == newest)


def _squash(t):
    return str(t).split(" (")[0].replace(" ", "")


def test_newest_round_lines_agree_with_their_pmc_summaries():
    newest, files = _lines()
    assert files, "no measured lines under profiles/"
    checked = 0
    for f in files:
        d = json.load(open(f))
        r = d.get("roofline") or {}
        if r.get("traffic") is None:
            continue
        src = str(r.get("traffic_source") or "")
        m = re.search(r"profiles/(r\d+[A-Za-z0-9_]*_pmc_traffic\.json)", src)
        assert m, "%s: traffic without a traffic_source file" % os.path.basename(f)
        cited = m.group(1)
        if newest >= 5:      # enforced from round 5 on
            assert _round_of(cited) == newest, "%s cites %s: a PMC summary of ANOTHER round (regenerate the line after the PMC pass: tools/profile_bench.sh)" % (os.path.basename(f), cited)
        path = os.path.join(PROF, cited)
        assert os.path.exists(path), "%s cites %s, which is not committed" % (os.path.basename(f), cited)
        s = json.load(open(path))
        if newest >= 5:
            assert abs(s["traffic_bytes_per_launch"] - r["traffic"]) <= 1e-6 * max(r["traffic"], 1.0), (os.path.basename(f), cited)
            parts = [p.strip() for p in str(s.get("kernel", "")).split(" + ") if p.strip()]
            bases = {_squash(p).split("<")[0] for p in parts}
            assert len(parts) == 1 or len(bases) == len(parts), "%s: a sum over template instances (%s) is not one launch's traffic" % (cited, s.get("kernel"))
            kn = _squash(r.get("kernel", ""))
            assert all(_squash(p) == kn or _squash(p).split("<")[0] == kn.split("<")[0] or _squash(p).startswith(kn) for p in parts), (os.path.basename(f), r.get("kernel"), s.get("kernel"))
            algo = r.get("algorithmic_bytes_per_launch")
            if algo:
                ratio = r["traffic"] / algo
                assert ratio <= 1.2 or abs(s["traffic_over_algorithmic"] - ratio) < 0.02 * ratio, "%s: traffic %.2f x algorithmic, the cited summary says %.2f x" % (os.path.basename(f), ratio, s["traffic_over_algorithmic"])
        checked += 1
    assert checked or newest < 5


def test_bench_lookup_prefers_the_newest_round_and_one_instance():
    """bench_common.pmc_traffic: newest round wins; a summary that sums template instances of one kernel is skipped."""
    import sys
    sys.path.insert(0, ROOT)
    import bench_common as bc
    got = bc.pmc_traffic("k_fftfilt_wave", {"streams_per_gpu": 64, "blocks_per_step": 16, "taps": 1023})      # (round 6: the 1023-tap instance is k_fftfilt_wave)
    if got is None:
        return
    rounds = [_round_of(f) for f in glob.glob(os.path.join(PROF, "r*_fftfilt*_pmc_traffic.json"))]
    m = re.search(r"profiles/(r\d+)", got[1])
    assert m and int(m.group(1)[1:]) == max(rounds)
    algo = 16.0 * 64 * 16 * (65537 - 1023)
    assert got[0] / algo < 1.2, "the 1023-tap instance moves ~1.02 x its algorithmic bytes; %.2f x means a sum over the sweep's instances was picked" % (got[0] / algo)
