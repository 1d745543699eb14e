"""Makes the REFERENCE importable on the GPU box -- TEST / BENCH INFRASTRUCTURE ONLY, never imported by the product.

The reference is a Python project: nothing to compile into ``oracle/_ref``.  What SURVEY.md 8(d) asks to be timed beside the native step is the
reference's own ``open_clip_train.train.train_one_epoch`` on the host cores of the MI355X box, and /root/reference does not exist there.  This
recipe packs the two packages that loop needs -- ``/root/reference/src/open_clip`` and ``/root/reference/src/open_clip_train``, read where they
lie -- into ONE archive, ``oracle/_ref/reference_src.zip``.  ``oracle/_ref/`` is git-ignored (the archive never enters the history, like the built
``.so`` files) and not gpurun-ignored (it travels with the repo snapshot, like them).  ``oracle/ref_shim.py`` unpacks it into a temporary
directory when /root/reference is absent; ``__graft_entry__.build()`` runs this when /root/reference is present.

    python -m oracle.fetch_ref            # -> oracle/_ref/reference_src.zip (+ its manifest printed)
"""
import hashlib
import os
import sys
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/src"
OUT_DIR = os.path.join(HERE, "_ref")
ARCHIVE = os.path.join(OUT_DIR, "reference_src.zip")
PACKAGES = ("open_clip", "open_clip_train")


def _files():
    for pkg in PACKAGES:
        for d, dirs, names in os.walk(os.path.join(SRC, pkg)):
            dirs[:] = sorted(x for x in dirs if x != "__pycache__")
            for n in sorted(names):
                if not n.endswith((".pyc", ".pyo")):
                    full = os.path.join(d, n)
                    yield full, os.path.relpath(full, SRC)


def fetch(verbose=False):
    """-> path of the archive, or None when /root/reference is not here (the GPU box: the archive it already carries is used as is)"""
    if not os.path.isdir(os.path.join(SRC, "open_clip")):
        return ARCHIVE if os.path.exists(ARCHIVE) else None
    os.makedirs(OUT_DIR, exist_ok=True)
    files = list(_files())
    h = hashlib.sha256()
    for full, rel in files:
        h.update(rel.encode())
        h.update(open(full, "rb").read())
    digest = h.hexdigest()
    if os.path.exists(ARCHIVE):
        try:
            with zipfile.ZipFile(ARCHIVE) as z:
                if z.comment.decode() == digest:
                    return ARCHIVE  # up to date
        except zipfile.BadZipFile:
            pass
    tmp = ARCHIVE + ".tmp"
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        z.comment = digest.encode()
        for full, rel in files:
            z.write(full, rel)
    os.replace(tmp, ARCHIVE)
    if verbose:
        print(f"{ARCHIVE}: {len(files)} files of {', '.join(PACKAGES)} from {SRC}, sha256 {digest[:16]}, {os.path.getsize(ARCHIVE) / 1e6:.1f} MB")
    return ARCHIVE


if __name__ == "__main__":
    p = fetch(verbose=True)
    print(p if p else "no /root/reference here and no archive yet")
    sys.exit(0 if p else 1)
