#!/usr/bin/env python3
"""Build-time proof obligation for every kernel that stages operands by LDS-DMA (`global_load_lds_*`: global memory -> LDS with no register in
between, completion visible only through the vector-memory counter).

Such a kernel publishes an LDS stage to the other wavefronts with `s_waitcnt vmcnt(N)` + `s_barrier`.  The count N is only right if the
requests sit in the instruction stream where the source put them -- hipcc moved loads across such a wait once (round 5, tri.hip: 8 of 10 pieces
of a diagonal inverse were still flying at the rendezvous; wrong tiles in SOME wavefronts of SOME runs, invisible to 630 tests).  This script
reads the gfx950 assembly of THIS build (`hipcc --cuda-device-only -S file.hip`) and replays every kernel that contains an LDS-DMA request
against a model of the vector-memory counter (gfx9: loads and stores retire in issue order on vmcnt):

  * every vector-memory instruction enters a FIFO; `s_waitcnt vmcnt(N)` retires the oldest entries until N are left;
  * at every `s_barrier`, count the LDS-DMA requests still in the FIFO.  A kernel whose design drains before each rendezvous must show 0; a
    kernel that keeps the NEXT stage's requests flying across the rendezvous (the stream-K GEMM: one stage = 6 pieces per wavefront) may show
    at most that many -- and because the FIFO is ordered, "at most one stage's worth" means every piece of the stage being published has landed;
  * the replay is linear in program order: at a label the FIFO is kept (a loop body is replayed with the state its first entry leaves; the
    kernels' waits are the same on every trip), at `s_endpgm` it is cleared.

usage: check_lds_dma_asm.py file.gfx950.s 'kernel-name-regex=allowed' ...      (kernels with LDS-DMA that match no rule: allowed = 0)
Exit status 0 = every kernel within its allowance, 1 = a violation (the Makefile stops the build), 2 = no LDS-DMA kernel found where one was
expected (a rule matched nothing: the pattern changed and the proof would be vacuous).
"""
import re
import sys

VMEM = re.compile(r"^\s*(global_|buffer_|scratch_|flat_)(load|store|atomic)")
WAIT = re.compile(r"vmcnt\((\d+)\)")


def is_dma(text):
    return text.startswith("global_load_lds") or (text.startswith("buffer_load") and re.search(r"\blds\b", text) is not None)


def replay(lines):
    fifo = []            # (is_dma, line_no, text)
    worst = 0
    worst_at = None
    n_dma = n_bar = 0
    for ln, raw in lines:
        text = raw.split(";")[0].strip()
        if not text or text.startswith(".") or text.endswith(":"):
            continue
        op = text.split()[0]
        if op == "s_endpgm":
            fifo.clear()
            continue
        if op == "s_barrier":
            n_bar += 1
            open_dma = [e for e in fifo if e[0]]
            if len(open_dma) > worst:
                worst, worst_at = len(open_dma), (ln, open_dma[0][1], open_dma[0][2])
            continue
        if op == "s_waitcnt":
            m = WAIT.search(text)
            if m:
                keep = int(m.group(1))
                del fifo[:max(0, len(fifo) - keep)]
            continue
        if VMEM.match(text):
            dma = is_dma(text)
            n_dma += dma
            fifo.append((dma, ln, text))
    return n_dma, n_bar, worst, worst_at


def main(argv):
    path = argv[1]
    rules = []
    for spec in argv[2:]:
        rx, allow = spec.rsplit("=", 1)
        rules.append([re.compile(rx), int(allow), 0])
    funcs, cur = {}, None
    with open(path) as f:
        for ln, raw in enumerate(f, 1):
            m = re.match(r"^(_Z\w+):", raw)
            if m and cur is None:
                cur = []
                funcs[m.group(1)] = cur
                continue
            if cur is not None:
                if raw.startswith(".Lfunc_end"):
                    cur = None
                    continue
                cur.append((ln, raw))
    bad = 0
    seen = 0
    for name, lines in sorted(funcs.items()):
        n_dma, n_bar, worst, at = replay(lines)
        if n_dma == 0:
            continue
        seen += 1
        allow = 0
        for r in rules:
            if r[0].search(name):
                allow = r[1]
                r[2] += 1
                break
        ok = worst <= allow
        print(f"check_lds_dma: {name[:96]}: {n_dma} LDS-DMA requests, {n_bar} barriers, at most {worst} open at a barrier (allowed {allow}) -> {'ok' if ok else 'VIOLATION'}")
        if not ok:
            print(f"check_lds_dma:   s_barrier at line {at[0]} while the request of line {at[1]} (`{at[2]}`) and {worst - 1} younger ones may be outstanding")
            bad = 1
    for r in rules:
        if r[2] == 0:
            print(f"check_lds_dma: rule `{r[0].pattern}` matched no kernel with LDS-DMA requests in {path}: the proof would be vacuous")
            bad = bad or 2
    if seen == 0 and not rules:
        print(f"check_lds_dma: no kernel with LDS-DMA requests in {path}")
    return bad


if __name__ == "__main__":
    sys.exit(main(sys.argv))
