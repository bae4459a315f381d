#!/usr/bin/env python3
"""Build-time proof obligation for tri.hip::trsm_fused_kernel<..., XASM = true>.

The kernel issues its X-operand loads from inline asm (`global_load_dword[x2] vD, vOff, s[base]` behind an `s_nop 4`) and covers them with
its OWN counted `s_waitcnt vmcnt(N)`.  hipcc believes the destination registers are defined at the asm statement, so nothing stops its
register allocator from copying or spilling one of them before the load has landed -- a silent wrong result.  This script reads
the gfx950 assembly hipcc produced for THIS build (`hipcc --cuda-device-only -S tri.hip`) and replays every XASM = true instantiation
against a model of the vector-memory counter:

  * every vector-memory instruction (global_/buffer_/scratch_/flat_ load or store, LDS-DMA included) enters a FIFO; `s_waitcnt vmcnt(N)`
    retires the oldest entries until N are left (gfx9: loads and stores return in issue order on this counter);
  * an asm-issued load is recognised by the `s_nop 4` in front of it and its scalar-base address form;
  * while such a load is in the FIFO, NO other instruction may name one of its destination registers -- as a source (copy, spill store,
    arithmetic) or as a destination (the register was handed to something else);
  * the replay is linear in program order; at a label the FIFO is kept (a loop body is checked with the state its first entry leaves --
    the kernel's waits are the same on every trip, so the first trip is the representative one) and at `s_endpgm` it is cleared.

Second obligation, for EVERY instantiation (XASM or not): the panels of U and the inverses of the diagonal blocks reach LDS through LDS-DMA
(`global_load_lds_*`) and are read by all wavefronts after an `s_barrier`; the wait in front of that barrier is a COUNTED one wherever
younger register loads may keep flying.  The count is only right if those loads sit behind the DMA pieces in the instruction stream -- and
hipcc may move them (it sank them to the end of the block in one build: 8 of 10 pieces were allowed to fly, a wrong tile in some wavefronts).
So: at every `s_barrier`, NO LDS-DMA request may be left in the FIFO.

Exit status: bit 0 set = an asm-load violation (the Makefile rebuilds tri.o with -DRLHIP_TF_XASM_DEFAULT=0: the plain-C++-load twin becomes the
default); bit 1 set = an LDS-DMA piece can be outstanding at a barrier (the Makefile rebuilds with -DRLHIP_TF_DRAIN=1 -- the diagonal block's
rendezvous drain the counter -- and checks THAT build; if it still fails the build stops).
usage: check_trsm_asm.py tri.gfx950.s
"""
import re
import sys

VMEM = re.compile(r"^\s*(global_|buffer_|scratch_|flat_)(load|store|atomic)")
REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")
WAIT = re.compile(r"vmcnt\((\d+)\)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def check_function(name, lines):
    fifo = []          # entries: (is_asm_load, dest_regs, line_no, text)
    guarded = {}       # register -> (line_no, text) of the asm load in flight that owns it
    problems = []
    dma_problems = []  # (barrier line, DMA line, text)
    prev_nop4 = False
    n_asm = 0
    for ln, raw in lines:
        text = raw.split(";")[0].strip()
        if not text or text.startswith(".") or text.endswith(":"):
            continue
        op = text.split()[0]
        if op == "s_nop":
            prev_nop4 = text.split()[1:] == ["4"]
            continue
        if op == "s_endpgm":
            fifo.clear(); guarded.clear(); prev_nop4 = False
            continue
        if op == "s_barrier":
            for (_, _, dln, dtext) in fifo:
                if dtext.startswith("global_load_lds"):
                    dma_problems.append((ln, dln, dtext))
            prev_nop4 = False
            continue
        if op == "s_waitcnt":
            m = WAIT.search(text)
            if m:
                keep = int(m.group(1))
                while len(fifo) > keep:
                    is_asm, dst, _, _ = fifo.pop(0)
                    if is_asm:
                        for r in dst:
                            guarded.pop(r, None)
            prev_nop4 = False
            continue
        operands = text[len(op):]
        is_vmem = bool(VMEM.match(text))
        is_asm_load = is_vmem and prev_nop4 and op in ("global_load_dwordx2", "global_load_dword") and re.search(r",\s*s\[\d+:\d+\]", operands)
        used = regs_of(operands)
        if is_asm_load:
            dst = regs_of(operands.split(",")[0])
            clash = (used - dst) & set(guarded)                 # its own address register must not be a register still in flight
        else:
            dst = set()
            clash = used & set(guarded)
        for r in sorted(clash):
            problems.append((ln, raw.strip(), r, guarded[r]))
        if is_vmem:
            fifo.append((bool(is_asm_load), dst, ln, text))
            if is_asm_load:
                n_asm += 1
                for r in dst:
                    if r in guarded:
                        problems.append((ln, raw.strip(), r, guarded[r]))
                    guarded[r] = (ln, text)
        prev_nop4 = False
    return n_asm, problems, dma_problems


def main(path):
    funcs, cur, name = {}, None, None
    with open(path) as f:
        for ln, raw in enumerate(f, 1):
            m = re.match(r"^(_Z\w*trsm_fused_kernel\w*):", raw)
            if m:
                name, cur = m.group(1), []
                funcs[name] = cur
                continue
            if cur is not None:
                if raw.startswith(".Lfunc_end"):
                    cur = None
                    continue
                cur.append((ln, raw))
    xasm = {n: l for n, l in funcs.items() if re.search(r"Li[012]ELb1EEE", n)}       # <T, NW, HPR, OOP, XASM = true>
    if not xasm:
        print("check_trsm_asm: no trsm_fused_kernel<..., XASM = true> instantiation found in", path)
        return 1
    bad = 0
    for n, lines in sorted(funcs.items()):
        is_xasm = n in xasm
        n_asm, problems, dma = check_function(n, lines)
        if is_xasm and n_asm == 0:
            print(f"check_trsm_asm: {n}: no asm-issued load recognised (pattern changed?)")
            bad |= 1
        if is_xasm:
            for ln, text, r, owner in problems[:12]:
                print(f"check_trsm_asm: {n}: line {ln}: `{text}` names v{r} while the asm load of line {owner[0]} (`{owner[1]}`) is still in flight")
            if problems:
                bad |= 1
        for bln, dln, dtext in dma[:12]:
            print(f"check_trsm_asm: {n}: s_barrier at line {bln} while the LDS-DMA request of line {dln} (`{dtext}`) may be outstanding")
        if dma:
            bad |= 2
        print(f"check_trsm_asm: {n[:60]}...: {n_asm} asm-issued loads, {len(problems) if is_xasm else 0} violations, {len(dma)} LDS-DMA requests open at a barrier")
    return bad


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
