#!/usr/bin/env python3
"""Rewrites the "Index of entry points" section of INTEGRATION.md from include/plonkit_amd.h: every exported function under the
heading of the header section it is declared in (the headings name the reference interface the group replaces).
tests/test_docs.py fails when the index is out of date.   usage: python tools/abi_index.py [--check]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEGIN, END = "<!-- abi-index:begin (tools/abi_index.py) -->", "<!-- abi-index:end -->"


def index():
    h = open(os.path.join(ROOT, "include", "plonkit_amd.h"), encoding="utf-8").read()
    sec, groups, order = "process, context (`Worker::new()`, src/plonk.rs:41,47,183)", {}, []
    lines = h.split("\n")
    for k, line in enumerate(lines):
        m = re.match(r"\s*/\* ---- (.*?)\s*-*\s*(\*/)?\s*$", line)
        if m:
            sec = m.group(1).strip().rstrip("-").strip()
            j = k
            while "*/" not in lines[j] and j + 1 < len(lines) and len(sec) < 260:      # a heading that runs over several comment lines
                j += 1
                sec += " " + lines[j].strip().lstrip("*").replace("*/", "").strip()
            sec = re.sub(r"\s+", " ", sec).strip()
            if len(sec) > 260:
                sec = sec[:257].rsplit(" ", 1)[0] + " …"
        if not re.match(r"^\s*(const char \*|int32_t|uint64_t|uint32_t|void)\s*\**\s*plk_", line):
            continue
        for n in re.findall(r"\b(plk_[a-z0-9_]+)\s*\(", line)[:1]:
            if sec not in groups:
                groups[sec] = []; order.append(sec)
            if n not in groups[sec]:
                groups[sec].append(n)
    rows = ["| header section (what the group replaces in the reference) | entry points |", "|---|---|"]
    for s in order:
        rows.append("| %s | %s |" % (s.replace("|", "\\|"), ", ".join("`%s`" % n for n in groups[s])))
    return "\n".join(rows), sum(len(v) for v in groups.values())


def main():
    path = os.path.join(ROOT, "INTEGRATION.md")
    doc = open(path, encoding="utf-8").read()
    table, count = index()
    block = "%s\n%s\n\n%d entry points; the declarations, argument conventions and error behaviour are in `include/plonkit_amd.h`.\n%s" % (BEGIN, table, count, END)
    if BEGIN in doc:
        new = doc[: doc.index(BEGIN)] + block + doc[doc.index(END) + len(END):]
    else:
        new = doc.rstrip("\n") + "\n\n## Index of entry points\n\n" + block + "\n"
    if "--check" in sys.argv:
        sys.exit(0 if new == doc else "INTEGRATION.md: the index of entry points is out of date (python tools/abi_index.py)")
    open(path, "w", encoding="utf-8").write(new)
    print("%d entry points indexed" % count)


if __name__ == "__main__":
    main()
