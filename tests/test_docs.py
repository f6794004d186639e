"""-m "not gpu": the documents keep up with the code — every environment variable the library or the `plonkit` binary reads is
listed in INTEGRATION.md's table, every entry point include/plonkit_amd.h declares is in INTEGRATION.md's index."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(*parts):
    with open(os.path.join(ROOT, *parts), encoding="utf-8") as f:
        return f.read()


def test_every_environment_variable_is_documented():
    csrc = os.path.join(ROOT, "plonkit_amd", "csrc")
    read = set()
    for name in sorted(os.listdir(csrc)):
        if name.endswith((".hip", ".cpp", ".h")):
            read |= set(re.findall(r'getenv\("((?:PLK|PLONKIT)_[A-Z0-9_]+)"\)', _read("plonkit_amd", "csrc", name)))
    assert len(read) >= 20, "the scan found suspiciously few variables: %s" % sorted(read)
    doc = _read("INTEGRATION.md")
    missing = sorted(v for v in read if "`%s`" % v not in doc)
    assert not missing, "read by the code, absent from INTEGRATION.md's table: %s" % missing


def test_the_index_of_entry_points_is_current():
    """INTEGRATION.md's index (tools/abi_index.py) lists every function include/plonkit_amd.h declares, under the header section
    that names the reference interface it replaces"""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "abi_index.py"), "--check"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    header = _read("include", "plonkit_amd.h")
    declared = set(re.findall(r"^\s*(?:const char \*|int32_t|uint64_t|uint32_t|void)\s*\**\s*(plk_[a-z0-9_]+)\s*\(", header, re.M))
    assert len(declared) >= 80
    doc = _read("INTEGRATION.md")
    index = doc[doc.index("abi-index:begin"): doc.index("abi-index:end")]
    missing = sorted(n for n in declared if "`%s`" % n not in index)
    assert not missing, missing
