"""The evidence chain between profiles/ and bench.py's `offline` block (CPU-only checks).

* no tracked profile of the CURRENT round holds a failed run (a traceback, a segmentation fault) instead of a table;
* bench.load_offline() takes every number from the newest tracked file it names as `source` — re-derived here from the same
  file by an independent reading;
* tools/collect_profiles.sh refuses to overwrite a tracked file with a failed run."""
import glob
import importlib.util
import os
import re
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAILED = re.compile(r"Traceback \(most recent call last\)|Segmentation fault|core dumped")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_for_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _current_round():
    rounds = sorted({os.path.basename(p)[:3] for p in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_*"))})
    return rounds[-1]


def test_no_tracked_profile_of_the_current_round_is_a_failed_run():
    r = _current_round()
    bad = [p for p in glob.glob(os.path.join(ROOT, "profiles", r + "_*")) if FAILED.search(open(p, errors="replace").read())]
    assert not bad, bad


def test_offline_numbers_are_lines_of_the_files_they_cite():
    off = _bench().load_offline()
    bq = off["ball_query_traffic_bytes"]
    assert bq.get("error") is None, bq
    path = os.path.join(ROOT, bq["source"].split(" ")[0])
    assert os.path.basename(path).startswith(_current_round()), "bench.py cites %s, not the current round's file" % path
    text = open(path).read()
    # independent reading: median of every FETCH_SIZE / WRITE_SIZE line of the two kernels
    kib = 0.0
    for kern in ("ball_query_cells_kernel", "grid_build"):
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            vals = [sorted(float(v) for v in m.group(1).split())
                    for m in re.finditer(r"^[^\n]*%s[^\n]*?\s%s\s+([-+0-9.e ]+)$" % (kern, counter), text, re.M)]
            assert len(vals) == 1, (kern, counter, vals)
            kib += vals[0][len(vals[0]) // 2]
    assert bq["bytes"] == int(kib * 1024)
    us = off["ball_query_kernels_us"]
    for name, value in us.items():
        if name.startswith("void"):
            assert re.search(r"%s.*calls \d+ (median|avg) %s us" % (re.escape(name), re.escape("%.2f" % value)), text), (name, value)
    st = off["step_traffic_mib"]
    assert st.get("error") is None, st
    spath = os.path.join(ROOT, st["source"].split(" ")[0])
    assert os.path.basename(spath).startswith(_current_round())
    last = [ln for ln in open(spath) if ln.startswith("ALL KERNELS")][-1].split()
    assert [st["fetch_reported"], st["write"]] == [float(last[-2]), float(last[-1])]


def test_collect_profiles_keeps_a_good_file_when_the_new_run_failed(tmp_path):
    work = tmp_path / "repo"
    (work / "profiles").mkdir(parents=True)
    (work / "tools").mkdir()
    (work / "src").mkdir()
    script = open(os.path.join(ROOT, "tools", "collect_profiles.sh")).read()
    (work / "tools" / "collect_profiles.sh").write_text(script)
    (work / "profiles" / "r99_step_hbm_traffic.txt").write_text("ALL KERNELS   1.0 2.0\n")
    (work / "profiles" / "r99_ops.txt").write_text("old table\n")
    (work / "src" / "step_hbm_traffic.txt").write_text(
        "tools/pmc_step.sh: line 6:  2987 Segmentation fault ...\nTraceback (most recent call last):\nIndexError\n")
    (work / "src" / "ops.txt").write_text("new table\n")
    (work / "src" / "bench_line.json").write_text("{}\n")
    (work / "src" / "bench_kernel_stats.csv").write_text("a,b\n")
    res = subprocess.run(["bash", "tools/collect_profiles.sh", "src", "r99"], cwd=work, capture_output=True, text=True)
    assert res.returncode == 1 and "NOT copied" in res.stderr
    assert (work / "profiles" / "r99_step_hbm_traffic.txt").read_text() == "ALL KERNELS   1.0 2.0\n"
    assert (work / "profiles" / "r99_ops.txt").read_text() == "new table\n"
