#!/usr/bin/env python
"""Where a kernel's spills sit: the instruction stream of one kernel in a `hipcc -S` listing reduced to its MFMA runs, barriers, branches,
labels, scratch accesses and SGPR-spill lane moves.  Usage: python tools/isa_spills.py file.s <substring of the mangled kernel name>"""
import re
import sys


def main():
    text = open(sys.argv[1]).read()
    m = re.search(r"\n(_Z[^\n:]*%s[^\n:]*):[^\n]*\n" % re.escape(sys.argv[2]), text)
    if not m:
        sys.exit("no such kernel")
    body = text[m.end():]
    body = body[:body.index("s_endpgm")]
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".", "//"))]
    out, mf = [], 0
    for l in lines:
        if l.startswith("v_mfma"):
            mf += 1
            continue
        if l.startswith(("scratch_", "s_barrier", "s_cbranch", "s_branch", "v_readlane", "v_writelane", "buffer_store", "buffer_load")) or l.endswith(":"):
            if mf:
                out.append("   [%d mfma]" % mf)
                mf = 0
            out.append(l.split(";")[0].strip())
    res, prev, cnt = [], None, 0
    for o in out:
        key = o.split()[0] if not o.endswith(":") else o
        if key == prev and key.startswith(("scratch_", "v_readlane", "v_writelane", "buffer_")):
            cnt += 1
        else:
            if cnt:
                res.append("      (x%d)" % (cnt + 1))
            cnt = 0
            res.append(o)
            prev = key
    print("\n".join(res))


if __name__ == "__main__":
    main()
