"""The profiling helpers under scripts/ (they post-process rocprofv3 output into the tables kept in profiles/)."""
import os
import sqlite3
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args):
    r = subprocess.run([sys.executable, os.path.join(REPO, "scripts", script)] + [str(a) for a in args],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
    assert r.returncode == 0, r.stdout
    return r.stdout


def _kernel_db(path):
    db = sqlite3.connect(path)
    db.execute("create table kernels (name text, start integer, end integer, queue_id integer)")
    t = 0
    for step in range(3):
        for name, dur, q in (("conv1_fwd", 8000, 2), ("conv2_fwd", 12700, 2), ("adam", 20500, 2)):
            db.execute("insert into kernels values (?,?,?,?)", (name, t, t + dur, q))
            t += dur
        t += 6000  # inter-launch gap
    db.commit()
    db.close()


def test_prof_summary_and_timeline(tmp_path):
    p = str(tmp_path / "r.db")
    _kernel_db(p)
    out = _run("prof_summary.py", p, 10)
    lines = [l for l in out.splitlines() if l.startswith(("adam", "conv"))]
    assert lines[0].startswith("adam") and " 3 " in lines[0] and "20.5" in lines[0]  # sorted by total time; avg us
    tail = _run("prof_summary.py", p, 10, 0.34)  # last third of the window = the last step only
    assert any(l.startswith("adam") and " 1 " in l for l in tail.splitlines())
    tl = _run("prof_timeline.py", p, 4).splitlines()
    assert "gap_us" in tl[0] and len(tl) == 5
    assert tl[1].split()[3] == "0.00" and any(l.split()[3] == "6.00" for l in tl[2:])  # the 6 us inter-launch gap shows up


def test_pmc_summary():
    out = _run("pmc_summary.py", os.path.join(REPO, "tests", "fixtures", "pmc_counter_collection.csv"), 1)
    assert "conv2_fwd_lds_kernel" in out and "15.0 us under the profiler" in out
    # 2.24e6 MFMA-busy cycles over 15 us x 2.4 GHz x 1024 SIMDs = 6.1 % of the chip's SIMD-cycles
    assert "MFMA pipe busy on 6.1%" in out
    assert "LDS bank-conflict cycles / LDS active = 45.0%" in out
    assert "waiting on any instruction / wave cycles = 10.0%" in out


def test_demo1_profile_prints_the_host_phase_split(tmp_path):
    """``--profile`` wires ``utils.trace.StepTimer`` into the loop: the log ends with the host-side split."""
    import subprocess, sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(repo, "demo1", "train.py"), "--device", "cpu", "--synthetic", "true",
                        "--training_steps", "3", "--eval_step_interval", "2", "--batch_size", "8", "--eval_train_set", "false",
                        "--profile"], cwd=str(tmp_path), env=dict(os.environ, PYTHONPATH=repo, OMP_NUM_THREADS="2"),
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("host phases:")]
    assert line and "train steps (launch)" in line[0] and "evaluation" in line[0] and "images/sec" in line[0], r.stdout[-1500:]
