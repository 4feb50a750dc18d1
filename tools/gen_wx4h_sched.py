#!/usr/bin/env python
"""Static issue schedule of one stage of conv_wx4h_kernel (csrc/conv_f16_wx4h.hip) -> csrc/conv_f16_wx4h_sched.inc.

conv_wx4h is the 4-wave / 8 x 32-pixel form of conv_wx4 (tools/gen_wx4_sched.py explains the slot model): TWO workgroups share a CU, so one
workgroup's prologue / epilogue runs beside the other's K loop.  What changes for the schedule:

  * the weight stage no longer fits LDS twice next to V (2 x 80 KB per CU), so it lives in a RING of four (dy)-groups of 4*NREP KB:
    a stage is three groups dy = 0,1,2 of 3*NREP MFMAs, each closed by a workgroup barrier; during group n the wave issues its NREP
    pieces of group n+3 (= the same dy of the NEXT stage) into the slot group n-1 has just left, and at the end of group n it waits for
    its pieces of group n+1 (issued two groups earlier) with s_waitcnt vmcnt(K): K = the vector-memory operations it has issued since --
    a number this script knows, because it places them.  The A fragments of a group can therefore only be read behind the barrier that
    opens it; the B fragments (V planes, written at least a stage earlier) any time.
  * 256 threads stage 8 + 2 rows: the main item is unchanged, the two halo rows are one (row, x-tile, channel) value PAIR per thread
    and stage (positions jw and jw+3 from the same six scalars): hP / hHi / hSub / hLo / hSt.

Usage: python tools/gen_wx4h_sched.py > virnet_amd/csrc/conv_f16_wx4h_sched.inc      (knobs: WX4_CAP, WX4_LDS_LAT ...)"""
import os
import sys

CAP = int(os.environ.get("WX4_CAP", "5"))
LDS_LAT = int(os.environ.get("WX4_LDS_LAT", "3"))
HEAD_CAP = int(os.environ.get("WX4H_HEAD_CAP", "14"))      # the slot behind a barrier: the first MFMA waits for its A fragments anyway
S1_START = int(os.environ.get("WX4_S1_START", "8"))
S1_START_PRE = int(os.environ.get("WX4_S1_START_PRE", "3"))


class Op:
    def __init__(self, name, code, cost, deps=(), earliest=0, kind="valu"):
        self.name, self.code, self.cost, self.deps, self.earliest, self.kind = name, code, cost, list(deps), earliest, kind
        self.slot = None
        self.deadline = None


def I(n):
    return "WX_I(%d)" % n


def put_ops(x, j, ops):
    rd = ["pA%d" % x]
    ops.append(Op("pA%d" % x, "pA(%s, %s);" % (I(x), I(j)), 2))
    if j in (1, 2, 3, 4):
        ops.append(Op("pB%d" % x, "pB(%s, %s);" % (I(x), I(j)), 2))
        rd.append("pB%d" % x)
        ops.append(Op("pV%d" % x, "pV(%s, %s);" % (I(x), I(j)), 2, [("pA%d" % x, 1), ("pB%d" % x, 1)]))
    else:
        ops.append(Op("pV%d" % x, "pV(%s, %s);" % (I(x), I(j)), 2, [("pA%d" % x, 1)]))
        rd.append("pV%d" % x)
    ops.append(Op("pHi%d" % x, "pHi(%s);" % I(x), 2, [("pV%d" % x, 1)]))
    ops.append(Op("pSub%d" % x, "pSub(%s);" % I(x), 4, [("pHi%d" % x, 1)]))
    ops.append(Op("pLo%d" % x, "pLo(%s);" % I(x), 2, [("pSub%d" % x, 1)]))
    ops.append(Op("pSt%d" % x, "pSt(%s, %s);" % (I(x), I(j)), 2, [("pLo%d" % x, 1)], kind="ldsw"))
    return rd


def halo_ops(jw, ops):
    ops.append(Op("hP", "hP(%s);" % I(jw), 4))
    ops.append(Op("hHi", "hHi();", 2, [("hP", 1)]))
    ops.append(Op("hSub", "hSub();", 2, [("hHi", 1)]))
    ops.append(Op("hLo", "hLo();", 1, [("hSub", 1)]))
    ops.append(Op("hSt", "hSt(%s);" % I(jw), 4, [("hLo", 1)], kind="ldsw"))
    return ["hP"]


def frag_reads(nrep):
    """(name, code, first use slot, earliest slot)"""
    gsz = 3 * nrep
    reads = []
    for dy in range(3):
        reads.append(("rdB%d" % dy, "rdB(%s);" % I(dy), gsz * dy, 0))
    for g in range(3 * nrep):
        reads.append(("rdA%d" % g, "rdA(%s);" % I(g), 3 * g, (g // nrep) * gsz))
    out = []
    for name, code, use, first in sorted(reads, key=lambda r: r[2]):
        o = Op(name, code, 2, kind="ldsr", earliest=first)
        o.deadline = max(first, use - LDS_LAT)
        out.append(o)
    return out


def dma_ops(nrep):
    gsz = 3 * nrep
    out = []
    for dy in range(3):
        for i in range(nrep):
            o = Op("dma%d_%d" % (dy, i), "dma(%s, %s);" % (I(dy), I(i)), 4, kind="dma")
            o.fixed = gsz * dy + 1 + i
            out.append(o)
    return out


def build(nrep, ji, pre):
    nm = 9 * nrep
    ops = frag_reads(nrep)
    dma = dma_ops(nrep)
    stg = []
    if ji == 0:
        rd = put_ops(0, 2, stg) + put_ops(1, 5, stg) + halo_ops(2, stg)
        lds = []
        for b in range(6):
            lds.append(Op("ldp%d" % b, "ldp(%s);" % I(b), 3, [(r, 1) for r in rd if r[0] == "p"], kind="vmem"))
        for b in range(6):
            lds.append(Op("ldh%d" % b, "ldh(%s);" % I(b), 3, [(r, 1) for r in rd if r[0] == "h"] + [("ldp5", 0)], kind="vmem"))
        first = [o for o in stg if o.name in rd]
        rest = [o for o in stg if o.name not in rd]
        ops += dma + first + lds + rest
    elif ji == 1:
        if pre >= 1:
            t0 = S1_START_PRE
            sdeps = []
            if pre == 2:      # SFT scale / shift of the chunk: read from the LDS table the prologue filled (no registers held across stages)
                stg.append(Op("rdsft", "rdsft();", 3, earliest=max(0, t0 - 1), kind="ldsr2"))
                sdeps = [("rdsft", 1)]
            for b in range(6):
                stg.append(Op("pr%d" % b, "pr(%s);" % I(b), 6 if pre == 1 else 12, sdeps, earliest=t0))
            stg.append(Op("prH", "prH();", 6, sdeps, earliest=t0))
            pdeps = [("pr%d" % b, 1) for b in range(6)]
            hdeps = [("prH", 1)]
        else:
            pdeps, hdeps = [], []
        p0 = len(stg)
        put_ops(0, 0, stg)
        put_ops(1, 3, stg)
        halo_ops(0, stg)
        for o in stg[p0:]:
            if o.name in ("pA0", "pA1", "pB1", "pV0"):
                o.deps += pdeps
                o.earliest = S1_START
            if o.name == "hP":
                o.deps += hdeps
                o.earliest = S1_START
        ops += dma + stg
    else:
        put_ops(0, 1, stg)
        put_ops(1, 4, stg)
        halo_ops(1, stg)
        ops += dma + stg
    return ops, nm


def build_final(nrep, ji):
    nm = 9 * nrep
    ops = frag_reads(nrep)
    if ji < 2:
        ops += dma_ops(nrep)
    if ji == 0:
        stg = []
        put_ops(0, 2, stg)
        put_ops(1, 5, stg)
        halo_ops(2, stg)
        ops += stg
    if ji == 2:
        for i in range(8):
            ops.append(Op("epf%d" % i, "epf(%s);" % I(i), 3, earliest=1 + i, kind="vmem"))
    return ops, nm


def schedule(ops, nm, nrep):
    gsz = 3 * nrep
    load = [0] * (nm + 1)
    by = {o.name: o for o in ops}

    def cap(s):
        return HEAD_CAP if (s < nm and s % gsz == 0) else CAP

    # 0. the weight pieces: fixed slots right behind the barrier that frees their ring slot
    for o in ops:
        if o.kind == "dma":
            o.slot = o.fixed
            load[o.slot] += o.cost
    # 1. fragment reads at their deadlines (moved earlier if the slot is full, never in front of the barrier that publishes them)
    for o in ops:
        if o.kind == "ldsr":
            s = o.deadline
            while s > o.earliest and load[s] + o.cost > cap(s):
                s -= 1
            o.slot = s
            load[s] += o.cost
    # 2. everything else in list order: earliest slot that satisfies its dependences and has room; a vector-memory load never shares
    #    the front of a group with the weight pieces (they must be OLDER than every load of their group: the vmcnt arithmetic counts on it)
    for o in ops:
        if o.slot is not None:
            continue
        s = o.earliest
        for d, lat in o.deps:
            assert by[d].slot is not None, (o.name, d)
            s = max(s, by[d].slot + lat)
        while s < nm:
            if o.kind == "vmem" and (s % gsz) <= nrep:
                s += 1
                continue
            if load[s] + min(o.cost, CAP) > cap(s):
                s += 1
                continue
            break
        s = min(s, nm)
        o.slot = s
        load[s] += o.cost
    return load


def vmem_per_group(ops, nm, nrep):
    gsz = 3 * nrep
    x = [0, 0, 0]
    for o in ops:
        if o.kind == "vmem":
            n = 4 if o.name == "ldsft" else 1
            x[min(2, o.slot // gsz)] += n
    return x


def emit(nrep, ji, pre, out, x_stage0):
    if pre is None:
        ops, nm = build_final(nrep, ji)
    else:
        ops, nm = build(nrep, ji, pre)
    load = schedule(ops, nm, nrep)
    gsz = 3 * nrep
    x = vmem_per_group(ops, nm, nrep)
    np_ = nrep
    # K of the wait that closes group dy: vector-memory operations issued behind the pieces of the group that is about to be read
    if pre is not None:
        if ji == 0:
            ks = [2 * np_ + x[0], 2 * np_ + x[0] + x[1], 2 * np_ + x[0] + x[1] + x[2]]
        elif ji == 1:
            ks = [2 * np_ + x_stage0[1] + x_stage0[2], 2 * np_ + x_stage0[2], 2 * np_]
        else:
            ks = [2 * np_, 2 * np_, 2 * np_]
    else:
        if ji < 2:
            ks = [2 * np_, 2 * np_, 2 * np_]
        else:
            ks = [np_ + x[0], x[0] + x[1], -1]          # (-1: the kernel's own end-of-loop wait)
    # loads of the epilogue's operand tile counted in K: the kernel subtracts them when the instantiation has no such tile
    es = [x[0], x[0] + x[1], 0] if (pre is None and ji == 2) else [0, 0, 0]
    order = {"ldsr": 0, "ldsr2": 0, "dma": 1, "vmem": 2, "valu": 3, "ldsw": 4}
    out.append("#define WX4H_STAGE_%d_%d_%d \\" % (nrep, ji, pre) if pre is not None else "#define WX4H_FINAL_%d_%d \\" % (nrep, ji))
    for s in range(nm + 1):
        here = sorted([o for o in ops if o.slot == s], key=lambda o: order[o.kind])
        line = "  SB(); " + " ".join(o.code for o in here) + " SB();"
        if s < nm:
            line += " mfma(%s, %s);" % (I(s // 3), I(s % 3))
            if s % 3 == 2:
                line += " WX_TS(%d);" % (s // 3)
            if (s + 1) % gsz == 0 and s + 1 < nm:
                line += " gbar(%s, %s, %s);" % (I((s + 1) // gsz - 1), I(ks[(s + 1) // gsz - 1]), I(es[(s + 1) // gsz - 1]))
        out.append(line + " \\")
    out.append("  gend(%s); \\" % I(ks[2]))
    out.append("  /* issue units per slot: %s ; vmem per group %s ; K %s */" % (" ".join(str(v) for v in load), x, ks))
    out.append("")
    return x


def main():
    out = ["// GENERATED by tools/gen_wx4h_sched.py (CAP=%d, LDS_LAT=%d, HEAD_CAP=%d) -- do not edit; see that script for the model." % (CAP, LDS_LAT, HEAD_CAP),
           "// WX4H_STAGE_<NREP>_<ji>_<PRE>: the body of one stage of conv_wx4h_kernel as fenced issue slots, one per MFMA, with the group",
           "// barriers gbar(dy, K) / gend(K) (K = vmcnt of the wait for the next group's weight pieces); WX4H_FINAL_<NREP>_<ji>: last chunk.", ""]
    for nrep in (1, 2, 3, 5):
        x0 = {}
        for ji in range(3):
            for pre in (0, 1, 2):
                x = emit(nrep, ji, pre, out, x0.get(pre))
                if ji == 0:
                    x0[pre] = x
            emit(nrep, ji, None, out, None)
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
