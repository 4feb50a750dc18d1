"""ISA summary of one kernel in a hipcc -S listing: instruction mix, scratch (spill) accesses and barriers with their position
relative to the MFMAs.  Usage: python tools/isa_stats.py file.s <substring of the mangled kernel name> [--dump]"""
import re
import sys


def main():
    path, key = sys.argv[1], sys.argv[2]
    text = open(path).read()
    m = re.search(r"\n(_Z[^\n:]*%s[^\n:]*):[^\n]*\n" % re.escape(key), text)
    if not m:
        sys.exit(f"no kernel matching {key}")
    body = text[m.end():]
    body = body[:body.index("s_endpgm")]
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", ".", "//"))]
    ins = [l for l in lines if not l.endswith(":")]
    mf = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
    print(m.group(1))
    print(f"instructions {len(ins)}, mfma {len(mf)} (first {mf[0]}, last {mf[-1]})")
    for pat in ("scratch_", "s_barrier", "ds_read", "ds_write", "global_load_lds", "global_load_dword", "buffer_store", "global_store", "s_waitcnt", "v_cvt", "s_nop"):
        idx = [i for i, l in enumerate(ins) if l.startswith(pat)]
        inside = [i for i in idx if mf[0] <= i <= mf[-1]]
        print(f"  {pat:20s} {len(idx):5d}   between first and last mfma: {len(inside)}")
    valu = [i for i, l in enumerate(ins) if l.startswith("v_") and not l.startswith("v_mfma")]
    print(f"  VALU (non-mfma)      {len(valu):5d}   between first and last mfma: {len([i for i in valu if mf[0] <= i <= mf[-1]])}")
    salu = [i for i, l in enumerate(ins) if l.startswith("s_") and not l.startswith(("s_waitcnt", "s_barrier", "s_nop"))]
    print(f"  SALU                 {len(salu):5d}   between first and last mfma: {len([i for i in salu if mf[0] <= i <= mf[-1]])}")
    if "--dump" in sys.argv:
        for i, l in enumerate(ins):
            print(i, l)


if __name__ == "__main__":
    main()
