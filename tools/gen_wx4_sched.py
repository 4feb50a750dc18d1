#!/usr/bin/env python
"""Static issue schedule of one stage of conv_wx4_kernel (csrc/conv_f16_wx4.hip) -> csrc/conv_f16_wx4_sched.inc.

A stage of the kernel is 9*NREP MFMAs of one wave.  A wave issues in order and an MFMA keeps the matrix pipe busy for 32 cycles, so
everything else the wave has to do in that stage -- fragment reads, the Winograd input transform + fp16 split of the next chunk's
pixels, pixel loads, weight DMA -- is cut into micro-operations (the lambdas of the kernel) and placed into the SLOTS between the
MFMAs: a dependent micro-operation lands at least one slot (= one MFMA, >= 32 cycles) behind its producer, so no VALU latency is
exposed, fragment reads land LDS_LAT slots ahead of the MFMA that consumes them, and no slot carries more than CAP issue units.
hipcc's own scheduler, given the same code as one block, sinks every read to its use and clumps the VALU work (tried: plain code,
sched_group_barrier pipelines); here every slot is fenced with sched_barrier and the placement is decided by this script.

Micro-operations (names = lambdas in the kernel; X = put context 0/1, J = Winograd position, b = pixel, g = MFMA group (dy, slab)):
  rdA(g) rdB(dy)            fragment reads (LDS)
  pA/pB/pV(X,J)             the rows of B^T on the thread's 6 pixels x 4 channels      pHi/pSub/pLo(X) fp16 split   pSt(X,J) LDS store
  hSa/hSb/hV(JW) hHi hSub hLo hSt(JW)   the same for the thread's one halo value
  pr(b) prHa prHb           pre-activation (stage 1)            ldp(b) ldh(b) ldsft   pixel loads of the next chunk (stage 0)
  dma(i)                    one 1-KB piece of the next stage's weights
Stage 0 writes positions {2,5} (+ halo pair 2) and requests the next chunk's pixels as soon as the old ones have been read (the DMA
pieces are issued BEFORE them: the end-of-stage wait vmcnt(pixel loads) counts on it); stage 1 pre-activates and writes {0,3}, touching
the new pixels from slot S1_START on (one stage after their request: HBM latency under load is about a stage); stage 2 writes {1,4}.
The kernel's end-of-stage waits are the s_waitcnt BUILTIN, so hipcc knows that no weight piece is pending when a stage starts and its own
waits for the pixel registers are exact (with asm waits every one of them became vmcnt(0), and the first touches had to sit in front of
the stage's first piece).

Usage: python tools/gen_wx4_sched.py > virnet_amd/csrc/conv_f16_wx4_sched.inc      (knobs: CAP, LDS_LAT below)"""
import sys

CAP = int(__import__('os').environ.get('WX4_CAP', '5'))        # issue units per slot (an MFMA hides about five single-issue instructions)
LDS_LAT = int(__import__('os').environ.get('WX4_LDS_LAT', '3'))    # slots between a fragment read and the MFMA that consumes it
HEAD_CAP = 14  # slot 0 sits behind the barrier, in front of the first MFMA which waits for its fragments anyway
S1_START = int(__import__("os").environ.get("WX4_S1_START", "8"))   # stage 1: first slot that touches the pixels requested in stage 0
S1_START_PRE = int(__import__("os").environ.get("WX4_S1_START_PRE", "3"))   # ... when they are pre-activated first (more work to place)
# MFMA order inside a row tap: "slab" = per slab the three products back to back (lo*hi, hi*lo, hi*hi: one accumulator, A changes once,
# B every time); "part" = per product all slabs (B fragment constant over 2*NREP MFMAs, then NREP with the lo plane)
ORDER = __import__("os").environ.get("WX4_ORDER", "slab")
PART_SEQ = (0, 2, 1)


def slot_mfma(s, nrep):
    """slot -> (group g = dy*nrep + slab, part)"""
    if ORDER == "slab":
        return s // 3, s % 3
    dy, t = divmod(s, 3 * nrep)
    return dy * nrep + t % nrep, PART_SEQ[t // nrep]


def first_use(g, nrep):
    if ORDER == "slab":
        return 3 * g
    dy, nr = divmod(g, nrep)
    return dy * 3 * nrep + nr


class Op:
    def __init__(self, name, code, cost, deps=(), earliest=0, kind="valu"):
        self.name, self.code, self.cost, self.deps, self.earliest, self.kind = name, code, cost, list(deps), earliest, kind
        self.slot = None


def I(n):
    return "WX_I(%d)" % n


def put_ops(x, j, ops):
    """micro-ops of one position for put context x; returns the names of the ops that read d0 (for the load WAR edges)"""
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
    ops.append(Op("hSa", "hSa(%s);" % I(jw), 3))
    ops.append(Op("hSb", "hSb(%s);" % I(jw), 3))
    ops.append(Op("hV", "hV();", 1, [("hSa", 1), ("hSb", 1)]))
    ops.append(Op("hHi", "hHi();", 1, [("hV", 1)]))
    ops.append(Op("hSub", "hSub();", 2, [("hHi", 1)]))
    ops.append(Op("hLo", "hLo();", 1, [("hSub", 1)]))
    ops.append(Op("hSt", "hSt(%s);" % I(jw), 2, [("hLo", 1)], kind="ldsw"))
    return ["hSa", "hSb"]


def build(nrep, ji, pre):
    ng, nm = 3 * nrep, 9 * nrep
    ndi = (12 * nrep + 7) // 8
    ops = []
    # fragment reads: deadline = the first MFMA that uses them
    reads = []
    for dy in range(3):
        reads.append(("rdB%d" % dy, "rdB(%s);" % I(dy), 3 * nrep * dy))
    for g in range(ng):
        reads.append(("rdA%d" % g, "rdA(%s);" % I(g), first_use(g, nrep)))
    for name, code, use in sorted(reads, key=lambda r: r[2]):
        o = Op(name, code, 2, kind="ldsr")
        o.deadline = max(0, use - LDS_LAT)
        ops.append(o)
    dma = [Op("dma%d" % i, "dma(%s);" % I(i), 4, kind="dma") for i in range(ndi)]
    stg = []
    if ji == 0:
        rd = put_ops(0, 2, stg) + put_ops(1, 5, stg) + halo_ops(2, stg)
        lds = []
        for b in range(6):
            lds.append(Op("ldp%d" % b, "ldp(%s);" % I(b), 3, [(r, 1) for r in rd if r[0] == "p"] + [(d.name, 0) for d in dma], kind="vmem"))
        for b in range(6):
            lds.append(Op("ldh%d" % b, "ldh(%s);" % I(b), 3, [(r, 1) for r in rd if r[0] == "h"] + [("ldp5", 0)] + [(d.name, 0) for d in dma], kind="vmem"))
        # the DMA pieces go first (they must be older than the pixel loads: the end-of-stage wait counts on it), one per slot from slot
        # 1 on; then whatever still reads the OLD pixels, then the loads -- the sooner they leave, the more of their latency stage 0
        # hides -- and the rest of the position work behind them
        for i, d in enumerate(dma):
            d.earliest = 1 + i
        first = [o for o in stg if o.name in rd]
        rest = [o for o in stg if o.name not in rd]
        ops += dma + first + lds + rest
    elif ji == 1:
        # the pixels requested in stage 0 are touched from slot S1_START on (the compiler's waits for them are exact now that the
        # end-of-stage waits are visible to it: gen header), pre-activation first
        if pre >= 1:
            t0 = S1_START_PRE
            sdeps = []
            if pre == 2:      # SFT scale / shift of the chunk: read from the LDS table the prologue filled (no registers held across stages)
                stg.append(Op("rdsft", "rdsft();", 3, earliest=max(0, t0 - 1), kind="ldsr2"))
                sdeps = [("rdsft", 1)]
            for b in range(6):
                stg.append(Op("pr%d" % b, "pr(%s);" % I(b), 6 if pre == 1 else 12, sdeps, earliest=t0))
            stg.append(Op("prHa", "prHa();", 6, sdeps, earliest=t0))
            stg.append(Op("prHb", "prHb();", 6, sdeps, earliest=t0))
            pdeps = [("pr%d" % b, 1) for b in range(6)]
            hdeps = [("prHa", 1), ("prHb", 1)]
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
            if o.name in ("hSa", "hSb"):
                o.deps += hdeps
                o.earliest = S1_START
        for i, d in enumerate(dma):
            d.earliest = 1 + i
        ops += dma + stg
    else:
        put_ops(0, 1, stg)
        put_ops(1, 4, stg)
        halo_ops(1, stg)
        for i, d in enumerate(dma):
            d.earliest = 1 + 2 * i
        ops += dma + stg
    return ops, nm


def build_first0(nrep, pre):
    """stage 0 of an item's FIRST chunk in the persistent kernel: positions {2,5} are in V already (stored by the previous epilogue, or by
    the workgroup's prologue), so the stage only multiplies, moves weights and requests the next chunk's pixels"""
    ng, nm = 3 * nrep, 9 * nrep
    ndi = (12 * nrep + 7) // 8
    ops = []
    reads = []
    for dy in range(3):
        reads.append(("rdB%d" % dy, "rdB(%s);" % I(dy), 3 * nrep * dy))
    for g in range(ng):
        reads.append(("rdA%d" % g, "rdA(%s);" % I(g), first_use(g, nrep)))
    for name, code, use in sorted(reads, key=lambda r: r[2]):
        o = Op(name, code, 2, kind="ldsr")
        o.deadline = max(0, use - LDS_LAT)
        ops.append(o)
    dma = [Op("dma%d" % i, "dma(%s);" % I(i), 4, kind="dma") for i in range(ndi)]
    for i, d in enumerate(dma):
        d.earliest = 1 + i
    lds = []
    for b in range(6):
        lds.append(Op("ldp%d" % b, "ldp(%s);" % I(b), 3, [(d.name, 0) for d in dma], kind="vmem"))
    for b in range(6):
        lds.append(Op("ldh%d" % b, "ldh(%s);" % I(b), 3, [("ldp5", 0)] + [(d.name, 0) for d in dma], kind="vmem"))
    return ops + dma + lds, nm


def build_final(nrep, ji):
    """stages of the LAST chunk: there is no next chunk to stage, so stage 0 only finishes positions {2,5} (+ halo pair 2), stages 1 and 2
    only read and multiply, and stage 2 -- whose weights' successor does not exist either -- requests the epilogue's first operand tile
    (residual or mask, micro-operation epf(i) = one 16-byte load per thread) into the registers the staging no longer needs"""
    ng, nm = 3 * nrep, 9 * nrep
    ndi = (12 * nrep + 7) // 8
    ops = []
    reads = []
    for dy in range(3):
        reads.append(("rdB%d" % dy, "rdB(%s);" % I(dy), 3 * nrep * dy))
    for g in range(ng):
        reads.append(("rdA%d" % g, "rdA(%s);" % I(g), first_use(g, nrep)))
    for name, code, use in sorted(reads, key=lambda r: r[2]):
        o = Op(name, code, 2, kind="ldsr")
        o.deadline = max(0, use - LDS_LAT)
        ops.append(o)
    if ji < 2:
        dma = [Op("dma%d" % i, "dma(%s);" % I(i), 4, kind="dma") for i in range(ndi)]
        for i, d in enumerate(dma):
            d.earliest = 1 + i
        ops += dma
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


def schedule(ops, nm):
    load = [0] * (nm + 1)
    by = {o.name: o for o in ops}

    def cap(s):
        return HEAD_CAP if s == 0 else CAP

    # 1. fragment reads at their deadlines (moved earlier if the slot is full)
    for o in ops:
        if o.kind == "ldsr":
            s = o.deadline
            while s > 0 and load[s] + o.cost > cap(s):
                s -= 1
            o.slot = s
            load[s] += o.cost
    # 2. everything else in list order: earliest slot that satisfies its dependences and has room
    for o in ops:
        if o.slot is not None:
            continue
        s = o.earliest
        for d, lat in o.deps:
            assert by[d].slot is not None, (o.name, d)
            s = max(s, by[d].slot + lat)
        while s < nm and load[s] + min(o.cost, CAP) > cap(s):      # (an operation larger than a slot takes an empty one)
            s += 1
        s = min(s, nm)
        o.slot = s
        load[s] += o.cost
    return load


def emit(nrep, ji, pre, out, kind=None):
    if kind == "first0":
        ops, nm = build_first0(nrep, pre)
    elif pre is None:
        ops, nm = build_final(nrep, ji)
    else:
        ops, nm = build(nrep, ji, pre)
    load = schedule(ops, nm)
    order = {"ldsr": 0, "ldsr2": 0, "dma": 1, "vmem": 2, "valu": 3, "ldsw": 4}
    if kind == "first0":
        out.append("#define WX4_STAGE0F_%d_%d \\" % (nrep, pre))
    else:
        out.append("#define WX4_STAGE_%d_%d_%d \\" % (nrep, ji, pre) if pre is not None else "#define WX4_FINAL_%d_%d \\" % (nrep, ji))
    for s in range(nm + 1):
        here = sorted([o for o in ops if o.slot == s], key=lambda o: order[o.kind])
        line = "  SB(); " + " ".join(o.code for o in here) + " SB();"
        if s < nm:
            line += " mfma(%s, %s);" % tuple(I(v) for v in slot_mfma(s, nrep))
            if s % 3 == 2:
                line += " WX_TS(%d);" % (s // 3)          # (timing builds: s_memtime stamp behind every MFMA group)
        out.append(line + " \\")
    out.append("  /* issue units per slot: %s */" % " ".join(str(x) for x in load))
    out.append("")


def main():
    out = ["// GENERATED by tools/gen_wx4_sched.py (CAP=%d, LDS_LAT=%d, HEAD_CAP=%d) -- do not edit; see that script for the model." % (CAP, LDS_LAT, HEAD_CAP),
           "// WX4_STAGE_<NREP>_<ji>_<PRE>: the body of one stage of conv_wx4_kernel as fenced issue slots, one per MFMA;",
           "// WX4_FINAL_<NREP>_<ji>: the same for the stages of the last chunk (nothing to stage for a next one).", ""]
    for nrep in (1, 2, 3):
        for ji in range(3):
            for pre in (0, 1, 2):
                emit(nrep, ji, pre, out)
            emit(nrep, ji, None, out)
    # the persistent kernel's extra stage form (conv_f16_wx4p.hip): three-slab workgroups, PRE 0 / 1
    out.append("// WX4_STAGE0F_<NREP>_<PRE>: first stage of an item in conv_wx4p_kernel (positions {2,5} of its first chunk are in V already)")
    out.append("")
    for pre in (0, 1):
        emit(3, 0, pre, out, kind="first0")
    sys.stdout.write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
