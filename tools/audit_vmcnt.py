"""Audit of the hand-placed ``s_waitcnt vmcnt(N)`` in the convolution kernels.

A hand-placed ``vmcnt(N)`` means "everything older than the newest N vector-memory operations of this wave has landed" (loads
complete in order).  Two uses in csrc/conv_split_kernels.hip:
  (a) k_conv_split_ws: [LDS-DMA of weight tile j+1][AR activation loads][vmcnt(AR)][barrier] -- covers the DMA group just issued
      provided at least N vector-memory instructions sit between its last piece and the wait;
  (b) k_conv_split_halo, 3 stages: [LDS-DMA of tile s+2 (NB pieces)][vmcnt(NB) or vmcnt(NB+NPIECE)][barrier] -- covers tile s+1,
      issued before the PREVIOUS barrier, provided at least N vector-memory instructions were issued since that barrier.
Were the compiler to move one of the counted instructions out of its window, a DMA piece could still be in flight when the
barrier hands the LDS stage to the consumer waves.  This script disassembles the file for gfx950 and, for every inline-asm
``s_waitcnt vmcnt(N)`` with N > 0, computes over the kernel's control-flow graph the MINIMUM, over every backward path, of (a) the
vector-memory instructions since the newest LDS-DMA instruction and (b) those since the previous ``s_barrier``.  The wait is sound
when either minimum reaches N.

    python tools/audit_vmcnt.py            # prints one line per hand-placed wait; exit code 1 if any window is short
"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "nerf-det_amd", "csrc", "conv_split_kernels.hip")


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
                        "--offload-device-only", "-S", SRC, "-o", out], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    # ---- control-flow graph per kernel: blocks split at labels and after branches ----
    VMEM = re.compile(r"^(buffer_load|global_load|buffer_store|global_store|buffer_atomic|global_atomic)")
    kernels = [(i, re.match(r"^(_Z\w+):", l).group(1)) for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
    bad, rows = 0, []
    for ki, (k0, kname) in enumerate(kernels):
        k1 = kernels[ki + 1][0] if ki + 1 < len(kernels) else len(lines)
        # instruction stream of this kernel: (line_no, text, in_inline_asm)
        ins, in_asm = [], False
        for i in range(k0 + 1, k1):
            t = lines[i].strip()
            if "#ASMSTART" in t:
                in_asm = True
                continue
            if "#ASMEND" in t:
                in_asm = False
                continue
            if not t or t.startswith(";") or (t.startswith(".") and not re.match(r"^\.LBB\w+:", t)):
                continue
            ins.append((i + 1, t, in_asm))
            if t.startswith("s_endpgm"):
                break
        blocks, cur, label_of = [], [], {}
        for item in ins:
            lm = re.match(r"^(\.LBB\w+):", item[1])
            if lm:
                if cur:
                    blocks.append(cur)
                cur = []
                label_of[lm.group(1)] = len(blocks)
                continue
            cur.append(item)
            if re.match(r"^(s_branch|s_cbranch_\w+|s_endpgm)", item[1]):
                blocks.append(cur)
                cur = []
        if cur:
            blocks.append(cur)
        preds = {b: [] for b in range(len(blocks))}
        for b, blk in enumerate(blocks):
            last = blk[-1][1] if blk else ""
            m = re.match(r"^(s_branch|s_cbranch_\w+)\s+(\.LBB\w+)", last)
            if m and m.group(2) in label_of:
                preds[label_of[m.group(2)]].append(b)
            if not last.startswith(("s_branch", "s_endpgm")) and b + 1 < len(blocks):
                preds[b + 1].append(b)

        def back_min(b, idx, stop_at_dma):
            """Fewest vector-memory instructions on any backward path from (block b, instruction idx) to the first barrier
            (or, with stop_at_dma, the first LDS-DMA instruction); None if no path reaches one."""
            import heapq
            best, heap, seen = None, [(0, b, idx)], {}
            while heap:
                cost, bb, ii = heapq.heappop(heap)
                if seen.get((bb, ii), 1 << 30) <= cost:
                    continue
                seen[(bb, ii)] = cost
                hit = False
                for j in range(ii - 1, -1, -1):
                    t = blocks[bb][j][1]
                    if t.startswith("s_barrier") and not stop_at_dma:
                        hit = True
                        # a barrier reached with the wave's queue drained (a vmcnt(0) right before it, nothing issued in between):
                        # nothing older than this path's instructions is in flight -- the path cannot leave a stale stage
                        for jj in range(j - 1, -1, -1):
                            tt = blocks[bb][jj][1]
                            if VMEM.match(tt):
                                break
                            if re.search(r"s_waitcnt .*vmcnt\(0\)|s_waitcnt vmcnt\(0\)", tt):
                                cost = 1 << 20
                                break
                        break
                    if VMEM.match(t):
                        if " lds" in t and stop_at_dma:
                            hit = True
                            break
                        cost += 1
                    if t.startswith("s_barrier") and stop_at_dma:
                        cost = None    # a barrier before any DMA: this path has no DMA group to protect
                        break
                if cost is None:
                    continue
                if hit:
                    best = cost if best is None else min(best, cost)
                    continue
                for pb in preds[bb]:
                    heapq.heappush(heap, (cost, pb, len(blocks[pb])))
            return best

        for b, blk in enumerate(blocks):
            for idx, (ln, t, in_asm) in enumerate(blk):
                w = re.search(r"s_waitcnt vmcnt\((\d+)\)", t)
                if not (in_asm and w):
                    continue
                n = int(w.group(1))
                c_dma, c_bar = back_min(b, idx, True), back_min(b, idx, False)
                ok = n == 0 or (c_dma is not None and c_dma >= n) or (c_bar is not None and c_bar >= n)
                bad += 0 if ok else 1
                rows.append((kname, ln, n, c_dma, c_bar, "ok" if ok else "SHORT"))
    for i, r in enumerate(rows):
        rows[i] = r[:4] + ("drained" if r[4] is not None and r[4] >= (1 << 20) else r[4],) + r[5:]
    for k, ln, n, a, c, status in rows:
        print(f"{k[:58]:58s} line {ln:6d} vmcnt({n:2d})  min ops since the newest LDS-DMA: {str(a):>4s}   min ops since the previous barrier: {str(c):>4s}   {status}")
    print(f"{len(rows)} hand-placed waits, {bad} short windows")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
