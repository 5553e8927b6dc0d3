"""Every profile a document or a source comment cites exists under profiles/ (the evidence is tracked with the claim)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cited_profiles_exist():
    files = [os.path.join(ROOT, f) for f in ("DESIGN.md", "README.md", "INTEGRATION.md", "bench.py", "include/gcn_spmm.h")]
    files += glob.glob(os.path.join(ROOT, "gcn_amd", "**", "*.py"), recursive=True)
    files += glob.glob(os.path.join(ROOT, "gcn_amd", "csrc", "*"))
    files += [f for f in glob.glob(os.path.join(ROOT, "tools", "*")) if os.path.isfile(f)]
    missing = []
    for f in files:
        try:
            text = open(f, encoding="utf-8").read()
        except (UnicodeDecodeError, IsADirectoryError):
            continue
        for m in re.finditer(r"profiles/([A-Za-z0-9_.*\-]+)", text):
            name = m.group(1).rstrip(".,;:)`")
            if not name:
                continue
            pat = os.path.join(ROOT, "profiles", name)
            if not (glob.glob(pat) or glob.glob(pat + "*")):
                missing.append((os.path.relpath(f, ROOT), "profiles/" + name))
    assert not missing, missing
