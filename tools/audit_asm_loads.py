#!/usr/bin/env python3
"""Audit hipcc output for the in-flight window of inline-asm LDS reads and inline-asm buffer loads.

An `asm volatile("ds_read_b128 %0, ...")` destination counts, for the compiler, as written at
the end of the asm statement -- but the data lands ~100+ cycles later.  Any compiler-generated
instruction that reads or writes those registers before the asm `s_waitcnt lgkmcnt(N)` that
retires them sees/destroys garbage (cdna_hip_programming.md 5.7).  This script walks each
kernel in a .s file along every control-flow path (branches and loops followed), tracks the
destination ranges of asm ds_reads (retired by asm `s_waitcnt lgkmcnt`) and of asm buffer / global
loads into registers (retired by asm `s_waitcnt vmcnt`) and reports every non-asm instruction that
touches a range while it is in flight.

Second check, same walk: an inline-asm VALU instruction that reads a VGPR an MFMA wrote a few
instructions earlier.  The hazard recogniser inserts the wait states an MFMA result needs before
a VALU read only for instructions it can see; with accumulators in AGPRs a visible
v_accvgpr_read always sits in between, with VGPR accumulators (two waves per SIMD, <= 256
registers) an asm ReLU read the result early and returned garbage.

Third check, same walk: a counted `s_waitcnt vmcnt(N)` in front of a slice barrier (the training kernels
keep their row stores in flight across it, the three-slot inference ring the next slice's pieces) must have at
least N vector-memory operations between it and the last LDS-DMA piece of the slice it opens, on every path.

Usage: audit_asm_loads.py file.s [kernel-name-substring]
"""
import re, sys

def regs(tok):
    out = set()
    for m in re.finditer(r'\b([va])\[(\d+):(\d+)\]', tok):
        out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    for m in re.finditer(r'\b([va])(\d+)\b', tok):
        out.add((m.group(1), int(m.group(2))))
    return out

def parse(lines):
    """-> list of (line_no, code, in_asm) for real instructions, plus label -> instruction index."""
    insts, labels, in_asm = [], {}, False
    for no, ln in lines:
        s = ln.strip()
        if '#ASMSTART' in s:
            in_asm = True; continue
        if '#ASMEND' in s:
            in_asm = False; continue
        m = re.match(r'^(\.LBB\w+):', s)
        if m:
            labels[m.group(1)] = len(insts); continue
        if not s or s.startswith(';') or s.startswith('.') or s.endswith(':'):
            continue
        insts.append((no, s.split(';')[0].strip(), in_asm))
    return insts, labels


def audit_mfma_to_asm_valu(insts, name):
    """Second check, straight-line: an inline-asm VALU instruction reading a VGPR an MFMA wrote <= 20 instructions earlier."""
    bad, mfma_dst = 0, []
    for idx, (no, code, in_asm) in enumerate(insts):
        mfma_dst = [(r, i) for r, i in mfma_dst if idx - i <= 20]   # 16-pass MFMA -> VALU read: 18 wait states
        if in_asm and code.startswith('v_') and not code.startswith('v_mfma'):
            srcs = regs(' '.join(code.split()[2:]))
            for d, i in mfma_dst:
                hit = srcs & d
                if hit:
                    bad += 1
                    if bad <= 12:
                        print(f"  {name}: line {no}: asm `{code}` reads {sorted(hit)[:4]} written by an MFMA {idx - i} instructions earlier")
        if code.startswith('v_mfma'):
            d = {x for x in regs(code.split()[1].rstrip(',')) if x[0] == 'v'}
            if d:
                mfma_dst.append((d, idx))
        elif code.startswith(('v_', 'ds_read', 'global_load', 'buffer_load', 'scratch_load')) and len(code.split()) > 1:
            over = regs(code.split()[1].rstrip(','))          # a later writer owns the register again
            mfma_dst = [(r - over, i) for r, i in mfma_dst]
    return bad


def audit_inflight(insts, labels, name):
    """First check, along every control-flow path (loops included): state = the ordered destinations of the asm
    ds_reads (retired by lgkmcnt) and of the asm buffer / global loads (retired by vmcnt) still in flight.  Each
    (instruction, state) pair is explored once; the states a loop produces repeat after an iteration, so the walk
    terminates (capped all the same)."""
    bad, reported = 0, set()
    seen = set()
    work = [(0, ((), ()))]
    steps = 0
    CAP = (32, 64)   # more asm loads than this in flight on a path = never retired (vmcnt is a 6-bit counter)
    KIND = ("ds_read", "load")
    while work and steps < 2_000_000:
        idx, state = work.pop()
        while idx < len(insts):
            key = (idx, state)
            if key in seen:
                break
            seen.add(key)
            steps += 1
            no, code, in_asm = insts[idx]
            lds, vm = state
            if in_asm:
                q = 0 if code.startswith('ds_read') else (1 if code.startswith(('buffer_load', 'global_load')) and ' lds' not in code else -1)
                if q >= 0:
                    entry = (frozenset(regs(code.split()[1].rstrip(','))), no)
                    lds, vm = (lds + (entry,), vm) if q == 0 else (lds, vm + (entry,))
                    if len((lds, vm)[q]) > CAP[q]:
                        if ('unbounded', no) not in reported:
                            reported.add(('unbounded', no))
                            bad += 1
                            print(f"  {name}: line {no}: more than {CAP[q]} asm {KIND[q]}s in flight on some path (never retired)")
                        break
                elif code.startswith('s_waitcnt'):
                    m = re.search(r'lgkmcnt\((\d+)\)', code)
                    if m:
                        keep = int(m.group(1))
                        lds = lds[len(lds) - keep:] if keep else ()
                    m = re.search(r'vmcnt\((\d+)\)', code)
                    if m:
                        keep = int(m.group(1))
                        vm = vm[max(0, len(vm) - keep):] if keep else ()
                state = (lds, vm)
                idx += 1
                continue
            if code.startswith('s_waitcnt'):
                if 'lgkmcnt(0)' in code:
                    lds = ()   # the compiler's own full LDS wait retires everything
                if 'vmcnt(0)' in code:
                    vm = ()
                state = (lds, vm)
                idx += 1
                continue
            if code.startswith('s_endpgm'):
                break
            if code.startswith('s_branch') or code.startswith('s_cbranch'):
                tgt = labels.get(code.split()[1])
                if tgt is not None:
                    work.append((tgt, state))
                if code.startswith('s_branch'):
                    break
                idx += 1
                continue
            if not code.startswith('s_barrier'):
                touched = regs(code)
                for q, queue in enumerate((lds, vm)):
                    for rs, at in queue:
                        hit = touched & rs
                        if hit and (no, at) not in reported:
                            reported.add((no, at))
                            bad += 1
                            if bad <= 12:
                                print(f"  {name}: line {no}: `{code}` touches {sorted(hit)[:4]} (asm {KIND[q]} at line {at} still in flight)")
            idx += 1
    return bad


THREE_SLOT_RING = re.compile(r"mlp_bf16x6_kernelILi\dELb[01]E")   # the six-piece forwards (inference and training): pieces are fetched TWO slices ahead


def wait_states(code):
    """Wait states an instruction puts between a vector write before it and an MFMA after it, as measured on MI355X
    (tools/valu_mfma_hazard_ubench.hip, profiles/r03_valu_mfma_hazard.log): s_nop N gives N + 1; one vector or scalar ALU
    instruction gives one (NOT enough on its own); an MFMA or an LDS / vector-memory instruction gives at least two."""
    op = code.split()[0]
    if op == 's_nop':
        return int(code.split()[1], 0) + 1
    if op.startswith(('v_mfma', 'ds_', 'buffer_', 'global_', 'flat_', 'scratch_')):
        return 2
    return 1


def audit_asm_valu_to_mfma(insts, name):
    """Fourth check, straight-line: an MFMA reading (as A / B operand) a VGPR that an inline-asm vector instruction wrote with
    fewer than TWO wait states in between.  Measured: with none or one (`s_nop 0` -- which is what hipcc puts behind an asm
    statement -- or one ALU instruction) the MFMA reads the OLD register value.  The hazard recogniser inserts the wait states
    for instructions it can see, not for inline asm: a build whose asm v_cvt_pk_bf16_f32 was followed by `s_nop 0` and the
    MFMA reading it returned garbage (round 3)."""
    bad, recent = 0, []     # (registers, wait states since, line)
    for idx, (no, code, in_asm) in enumerate(insts):
        if code.startswith('v_mfma'):
            ops = code.split(None, 1)[1].split(',')
            srcs = regs(','.join(ops[1:3]))          # srcA, srcB
            for r, ws, n0 in recent:
                if ws < 2 and r & srcs:
                    bad += 1
                    print(f"  {name}: line {no}: `{code}` reads a register an inline-asm vector instruction wrote {ws} wait state(s) earlier (line {n0})")
                    break
        recent = [(r, ws + wait_states(code), n0) for r, ws, n0 in recent if ws + wait_states(code) < 2]
        if in_asm and code.startswith('v_') and not code.startswith('v_mfma'):
            recent.append((regs(code.split(None, 1)[1].split(',')[0]), 0, no))
    return bad


def audit_counted_slice_waits(insts, labels, name, slots=2):
    """Third check: the slice barriers of the MLP kernels wait for this wave's LDS-DMA pieces (`buffer_load ... lds`) of
    the slice being opened with `s_waitcnt vmcnt(N)` + `s_barrier`.  vmcnt retires in issue order, so the N youngest
    vector-memory operations may only stay in flight if every one of them was issued AFTER the last piece of the slice being
    opened -- else the wait can pass with such a piece still in flight.  In a two-slot ring that piece is the last one
    issued before the wait (the younger operations are the training kernels' row stores); in a three-slot ring the slice
    being opened was fetched a slice earlier, so it is the last piece issued before the PREVIOUS slice barrier (and the
    twelve pieces issued since are the next slice's, legitimately in flight).  State = operations since either piece
    (saturating); every (instruction, state) triple is explored once, branches and loops followed."""
    bad, reported, seen = 0, set(), set()
    work = [(0, 64, 64)]          # nothing is known at kernel entry: no piece has been issued yet
    steps = 0
    while work and steps < 4_000_000:
        idx, since, since_prev = work.pop()
        while idx < len(insts):
            if (idx, since, since_prev) in seen:
                break
            seen.add((idx, since, since_prev))
            steps += 1
            no, code, in_asm = insts[idx]
            op = code.split()[0] if code else ''
            if op.startswith(('buffer_load', 'global_load', 'scratch_load', 'flat_load')) and code.rstrip().endswith(' lds'):
                since, since_prev = 0, min(since_prev + 1, 64)   # an LDS-DMA piece
            elif op.startswith(('buffer_', 'global_', 'scratch_', 'flat_')) and not op.startswith('buffer_wbinvl') and not op.startswith('buffer_inv'):
                since, since_prev = min(since + 1, 64), min(since_prev + 1, 64)
            elif op == 's_waitcnt':
                m = re.search(r'vmcnt\((\d+)\)', code)
                nxt = next((c for _, c, a in insts[idx + 1: idx + 3] if c), '')
                if m and in_asm and nxt.startswith('s_barrier'):
                    n, have = int(m.group(1)), (since_prev if slots == 3 else since)
                    if n > 0 and have < n and (no,) not in reported:
                        reported.add((no,))
                        bad += 1
                        print(f"  {name}: line {no}: `{code}` before a slice barrier, but only {have} vector-memory operations "
                              f"were issued since the last LDS-DMA piece of the slice it opens ({slots}-slot ring) on some path: a piece may still be in flight")
                    since_prev = since        # the last piece so far precedes this barrier
            if code.startswith('s_endpgm'):
                break
            if code.startswith('s_branch') or code.startswith('s_cbranch'):
                tgt = labels.get(code.split()[1])
                if tgt is not None:
                    work.append((tgt, since, since_prev))
                if code.startswith('s_branch'):
                    break
            idx += 1
    return bad


def audit(lines, name, full_name=None):
    insts, labels = parse(lines)
    bad = audit_inflight(insts, labels, name) + audit_mfma_to_asm_valu(insts, name) + audit_asm_valu_to_mfma(insts, name)
    # the two-slot weight ring of the MLP / delta-chain kernels: every barrier needs ALL of the wave's pieces.  (The dW GEMM's
    # three-buffer tiles leave the pieces of the NEXT chunk in flight on purpose: there the younger operations are pieces.)
    if "mlp_" in name or "delta_chain" in name or "render_fused" in name:
        bad += audit_counted_slice_waits(insts, labels, name, 3 if THREE_SLOT_RING.search(full_name or name) else 2)
    return bad


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ''
    cur, name, total = [], None, 0
    for no, ln in enumerate(open(path), 1):
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            name, cur = m.group(1), []
        if name:
            cur.append((no, ln))
            if 's_endpgm' in ln:
                if want in name and 'kernel' in name:
                    b = audit(cur, name[:40], name)
                    print(f"{name[:60]}: {b} suspicious touches")
                    total += b
                name = None
    sys.exit(1 if total else 0)

if __name__ == '__main__':
    main()
