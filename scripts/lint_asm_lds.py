#!/usr/bin/env python3
"""Static check of the gfx950 assembly of kernels that read LDS through inline-asm `ds_read_b128` (csrc/hvx_flat_tile.hip).

The compiler believes an inline-asm output is valid as soon as the asm statement has executed, so nothing stops it from moving,
copying (v_mov / v_accvgpr_write) or consuming such a register before the LDS has answered; the kernels state their own
`s_waitcnt lgkmcnt(N)` instead.  This lint reports any instruction that touches a register a `ds_read` has requested but that no
`s_waitcnt lgkmcnt` has covered yet.  (Found on hardware first: the fp8 instantiation of flat_tile4_kernel, at 256 VGPRs + 256
AGPRs, copied fragment registers with v_mov right behind their ds_read and returned wrong candidates.)

Method: the kernel's text is cut into basic blocks (labels, s_branch / s_cbranch_* / s_endpgm), a forward data-flow pass carries
the set of registers with an outstanding request across the edges (union over predecessors, iterated to a fixed point), and
inside a block the requests are kept in issue order: LDS operations return in order, so `lgkmcnt(N)` completes everything but
the N youngest -- if the block itself has issued at least N since its entry, everything inherited is complete too (otherwise the
inherited set is kept: conservative).

Second rule (round 4, csrc/hvx_flat_smallb.hip): the compiler's hazard recogniser pads the wait states between a matrix
instruction's write-back and a VALU / LDS / memory instruction that reads the result -- but not for an instruction inside an
inline-asm statement, which it cannot see into.  An asm `ds_write_b32` of an accumulator register issued right behind the last
v_mfma of a block stored the value of one MFMA step earlier (found on hardware by tests/native/smallq_probe.hip).  The lint counts
issue slots (s_nop N = N + 1) since the last v_mfma / v_smfmac that wrote a register and reports an inline-asm instruction
reading it sooner than MFMA_WAIT slots later (19: the 16-pass figure, an upper bound for every shape used here); the count is
carried across blocks as the minimum over predecessors.

usage: lint_asm_lds.py file.s kernel_substring [kernel_substring ...]      (exit status 1 when a hazard is found)"""
import re
import sys

MFMA_WAIT = 19
REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
LABEL = re.compile(r"^([.\w$]+):")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.update((m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1))
        else:
            out.add((m.group(4), int(m.group(5))))
    return frozenset(out)


def parse_blocks(lines):
    """-> (blocks: list of dicts {label, insts: [(lineno, text)], succ: [label, or None for fall-through]}, label -> index)"""
    blocks, cur = [], {"label": None, "insts": [], "succ": []}
    in_asm = False
    for no, raw in lines:
        if "#ASMSTART" in raw:
            in_asm = True
        elif "#ASMEND" in raw:
            in_asm = False
        line = raw.split(";")[0].rstrip()
        m = LABEL.match(line)
        if m:
            if cur["insts"] or cur["label"] is not None:
                if not cur["succ"]:
                    cur["succ"] = [None]  # falls through
                blocks.append(cur)
            cur = {"label": m.group(1), "insts": [], "succ": []}
            continue
        line = line.strip()
        if not line or line.startswith("."):
            continue
        op = line.split()[0]
        cur["insts"].append((no, line))
        if in_asm:
            cur.setdefault("asm", set()).add(no)
        if op == "s_branch":
            cur["succ"] = [line.split()[1]]
            blocks.append(cur)
            cur = {"label": None, "insts": [], "succ": []}
        elif op.startswith("s_cbranch"):
            cur["succ"] = [line.split()[1], None]
            blocks.append(cur)
            cur = {"label": None, "insts": [], "succ": []}
        elif op == "s_endpgm":
            cur["succ"] = ["<end>"]
            blocks.append(cur)
            cur = {"label": None, "insts": [], "succ": []}
    if cur["insts"]:
        blocks.append(cur)
    index = {b["label"]: i for i, b in enumerate(blocks) if b["label"]}
    return blocks, index


def run_block(block, inherited, report):
    """walk one block; `inherited` = registers requested before the block and not known complete; returns the out set"""
    inherited = set(inherited)
    local = []  # [(dest regs, lineno, text)] requests of this block in issue order (other LDS / scalar-memory ops: empty set)
    for no, line in block["insts"]:
        op = line.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", line)
            if m:
                keep = int(m.group(1))
                if len(local) >= keep:
                    inherited.clear()
                    local = local[len(local) - keep:] if keep else []
            continue
        rest = line[len(op):]
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            parts = [p.strip() for p in rest.split(",")]
            dest = regs_of(parts[0]) if op.startswith("ds_read") else frozenset()
        else:
            dest = None
        touched = regs_of(rest)
        if report is not None:
            hit = None
            for regs, pno, ptxt in local:
                if regs & touched:
                    hit = (pno, ptxt)
                    break
            if hit is None and inherited & touched:
                hit = (0, "a request of a predecessor block")
            if hit is not None:
                report.append((no, line, hit[0], hit[1]))
        if dest is not None:
            local.append((dest, no, line))
    out = set(inherited)
    for regs, _, _ in local:
        out |= regs
    return out


def run_block_mfma(block, state, report):
    """state: register -> issue slots since a matrix instruction wrote it (absent = long ago); returns the out state"""
    state = dict(state)
    asm = block.get("asm", ())
    for no, line in block["insts"]:
        op = line.split()[0]
        rest = line[len(op):]
        if report is not None and no in asm and not op.startswith("s_"):
            parts = [p.strip() for p in rest.split(",")]
            # operands an instruction READS: all of them for stores / ds_write, all but the first otherwise
            reads = regs_of(rest if op.startswith("ds_write") or "store" in op else ",".join(parts[1:]))
            for r in reads:
                if r in state and state[r] < MFMA_WAIT:
                    report.append((no, line, state[r]))
                    break
        slots = 1
        if op == "s_nop":
            slots = int(rest.strip() or 0) + 1
        for r in list(state):
            state[r] += slots
            if state[r] >= 64:
                del state[r]
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            for r in regs_of(rest.split(",")[0]):
                state[r] = 0
    return state


def lint_kernel_mfma(lines):
    blocks, index = parse_blocks(lines)
    n = len(blocks)
    succ = []
    for i, b in enumerate(blocks):
        s = []
        for t in b["succ"]:
            if t is None:
                if i + 1 < n:
                    s.append(i + 1)
            elif t in index:
                s.append(index[t])
        succ.append(s)
    ins = [dict() for _ in range(n)]
    work = list(range(n))
    rounds = 0
    while work and rounds < 100000:
        rounds += 1
        i = work.pop()
        out = run_block_mfma(blocks[i], ins[i], None)
        for j in succ[i]:
            changed = False
            for r, v in out.items():
                if r not in ins[j] or v < ins[j][r]:
                    ins[j][r] = v
                    changed = True
            if changed:
                work.append(j)
    problems = []
    for i, b in enumerate(blocks):
        run_block_mfma(b, ins[i], problems)
    return problems


def lint_kernel(lines):
    blocks, index = parse_blocks(lines)
    n = len(blocks)
    succ = []
    for i, b in enumerate(blocks):
        s = []
        for t in b["succ"]:
            if t is None:
                if i + 1 < n:
                    s.append(i + 1)
            elif t in index:
                s.append(index[t])
        succ.append(s)
    ins = [set() for _ in range(n)]
    work = list(range(n))
    while work:
        i = work.pop()
        out = run_block(blocks[i], ins[i], None)
        for j in succ[i]:
            if not out <= ins[j]:
                ins[j] |= out
                work.append(j)
    problems = []
    for i, b in enumerate(blocks):
        run_block(b, ins[i], problems)
    return problems


def main():
    path, wanted = sys.argv[1], sys.argv[2:]
    text = open(path).read().split("\n")
    bad = 0
    for want in wanted:
        starts = [i for i, l in enumerate(text) if re.match(r"^[A-Za-z_][\w$.]*:", l) and want in l.split(":")[0]]
        if not starts:
            print(f"{want}: kernel not found")
            bad = 1
            continue
        for st in starts:
            end = next(i for i in range(st, len(text)) if "s_endpgm" in text[i])
            body = [(i + 1, text[i]) for i in range(st + 1, end + 1)]
            probs = lint_kernel(body)
            late = lint_kernel_mfma(body)
            print(f"{text[st].split(':')[0]}: {len(probs)} hazard(s), {len(late)} inline-asm read(s) inside a matrix instruction's write-back window")
            for no, line, slots in late[:8]:
                print(f"   line {no}: `{line}` reads a register a v_mfma wrote {slots} issue slot(s) earlier (< {MFMA_WAIT})")
            bad |= bool(late)
            for no, line, pno, ptxt in probs[:int(__import__("os").environ.get("LINT_SHOW", "8"))]:
                print(f"   line {no}: `{line}` touches a register requested at line {pno}: `{ptxt}` with no covering s_waitcnt lgkmcnt")
            bad |= bool(probs)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
