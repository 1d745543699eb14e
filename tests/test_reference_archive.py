"""CPU: the archive that carries the reference's two packages to the GPU box (oracle/fetch_ref.py -> oracle/_ref/reference_src.zip, git-ignored,
shipped like the built .so) is what bench.py's ``cpu_baseline`` leg runs there -- SURVEY.md 8(d): the reference's own train_one_epoch on the host
cores beside the native step.  Here: the recipe packs exactly the files of /root/reference/src/open_clip{,_train}; a process that has NO
/root/reference imports the reference from the archive and runs its ``train_one_epoch`` (oracle/ref_cpu_baseline.py); the product never
imports anything under oracle/."""
import json
import os
import subprocess
import sys
import zipfile

import pytest

from oracle.ref_shim import reference_available

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not reference_available(), reason="reference tree not present")


def test_archive_holds_the_two_packages_and_nothing_else():
    from oracle import fetch_ref
    path = fetch_ref.fetch()
    assert path and os.path.exists(path)
    with zipfile.ZipFile(path) as z:
        names = z.namelist()
        assert len(z.comment) == 64  # the sha256 over names + contents the next fetch() compares with
    want = sorted(rel for _, rel in fetch_ref._files())
    assert sorted(names) == want and all(n.split("/")[0] in ("open_clip", "open_clip_train") for n in names)
    assert "open_clip/loss.py" in names and "open_clip_train/train.py" in names and not [n for n in names if "__pycache__" in n]
    # out of the history, into the snapshot
    assert "oracle/_ref/" in open(os.path.join(ROOT, ".gitignore")).read().split()
    assert "oracle/_ref" not in open(os.path.join(ROOT, ".gpurunignore")).read()
    tracked = subprocess.run(["git", "-C", ROOT, "ls-files", "oracle/_ref"], capture_output=True, text=True).stdout.strip()
    assert tracked == ""


def test_reference_train_one_epoch_runs_from_the_archive(tmp_path):
    """as on the GPU box: /root/reference hidden (ref_shim.REFERENCE_SRC pointed at nothing), two steps of the reference's loop from the archive"""
    out = tmp_path / "ref.json"
    code = ("import sys; sys.path.insert(0, %r); sys.argv = ['ref_cpu_baseline', '--steps', '3', '--threads', '8', '--no-port', '--out', %r];"
            "from oracle import ref_shim; ref_shim.REFERENCE_SRC = '/nonexistent';"
            "from oracle import ref_cpu_baseline as m; m.main()") % (ROOT, str(out))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, PYTHONDONTWRITEBYTECODE="1"))
    assert r.returncode == 0, r.stderr[-2000:]
    rec = json.load(open(out))
    assert rec["kind"] == "reference" and "oracle/_ref/reference_src.zip" in rec["where"] and rec["pairs_per_s"] > 0
    assert rec["pairs_per_s_reference_log_line"] and abs(rec["pairs_per_s_reference_log_line"] / rec["pairs_per_s"] - 1) < 0.5


def test_product_never_imports_the_oracle_or_the_reference():
    import re
    bad = []
    for d, _, names in os.walk(os.path.join(ROOT, "open_clip_amd")):
        for n in names:
            if n.endswith(".py"):
                src = open(os.path.join(d, n)).read()
                for m in re.finditer(r"^\s*(from|import)\s+(oracle|open_clip\b|open_clip_train)[^\n]*", src, flags=re.M):
                    # the ONE seam where the caller's own `open_clip` package is touched: open_clip_amd.create_task wraps the native model and loss in
                    # the reference's task classes (factory.py:975-1043) -- imported inside that function, at call time, nothing else of the package
                    if n == "factory.py" and m.group(0).strip() == "from open_clip.task import CLIPTask, SigLIPTask":
                        continue
                    bad.append((os.path.join(d, n), m.group(0).strip()))
    assert not bad, bad
