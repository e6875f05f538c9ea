#!/usr/bin/env python3
"""Offline (CPU) estimate of k_scan_region's LDS bank-conflict cycles under different lane <-> window arrangements.

The region kernel's chunk stages run lane = window; a wave's 64 lanes are 64 consecutive windows of one rung along a window ROW
(f = i * nj + j, 64 consecutive f).  Its two byte reads per tree level cost 2 LDS cycles when the lanes sit at the same tree node
(addresses base + lane * step + one offset) and ~6.9 when every lane is at its own node (scripts/micro/lds_gather.hip: the bank
model -- two groups of 32 lanes, (a / 4) mod 32, one cycle per distinct dword on the busiest bank -- predicts the measured cycles
within 2 %).  Neighbouring windows see nearly the same pixels and mostly take the same branch, so HOW the lanes of a 32-lane group
are laid over the window grid decides how often they diverge.  This script walks the first trees of the facefinder cascade over a
SYN-FACES frame on the CPU (NumPy, the reference's arithmetic: pigo.go:123-135), compacts the survivors after each stage as the
kernel does, and prices every wave-wide byte read with that bank model for

    row     64 consecutive windows of a window row (the kernel today)
    16x4    a block of 16 x 4 windows per wave, each 32-lane group an 8 x 4 block
    8x8     a block of 8 x 8 windows per wave, each 32-lane group an 8 x 4 block

    python scripts/lds_conflict_sim.py [--rows 1080 --cols 1920] [--smax 51] [--pitch 380]
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np


def group_cycles(addr):
    """addr: int64 [nwaves, 64] byte addresses (-1 = inactive lane) -> mean LDS cycles per wave instruction (bank model)."""
    tot = 0.0
    for g in range(2):
        a = addr[:, 32 * g:32 * g + 32]
        dw = np.where(a >= 0, a >> 2, -1)
        cyc = np.ones(a.shape[0], dtype=np.int64)
        # per bank: number of distinct dwords
        bank = np.where(dw >= 0, dw & 31, -1)
        srt = np.sort(np.where(dw >= 0, (bank << 40) | dw, -1), axis=1)  # sort by (bank, dword)
        b = srt >> 40
        new = np.ones_like(srt, dtype=bool)
        new[:, 1:] = srt[:, 1:] != srt[:, :-1]
        new &= srt >= 0
        # count distinct dwords per bank: run-length over sorted banks
        for k in range(32):
            cnt = (new & (b == k)).sum(axis=1)
            cyc = np.maximum(cyc, cnt)
        tot += cyc.mean()
    return tot


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1080)
    ap.add_argument("--cols", type=int, default=1920)
    ap.add_argument("--smin", type=int, default=20)
    ap.add_argument("--smax", type=int, default=51)
    ap.add_argument("--pitch", type=int, default=380)
    ap.add_argument("--cell-w", type=int, default=320)
    ap.add_argument("--cell-h", type=int, default=270)
    ap.add_argument("--trees", type=int, default=4)
    ap.add_argument("--kind", default="faces")
    a = ap.parse_args()
    from oracle.np_restatement import NpPigo
    from pigo_amd import synth
    pg = NpPigo.unpack(synth.facefinder_bytes())
    img = synth.make_frames(a.kind, 1, a.rows, a.cols, seed=1234)[0].astype(np.int64)
    # the reference's ladder (pigo.go:226-250)
    scales, s = [], float(a.smin)
    while s <= a.smax:
        scales.append(int(s))
        s = s * 1.1
    arrangements = ("row", "16x4", "8x8")
    tot = {k: np.zeros((a.trees, 6)) for k in arrangements}   # LDS cycles of pixel reads, weighted by wave instructions
    nin = {k: np.zeros(a.trees) for k in arrangements}
    for s in scales:
        step = max(int(0.1 * s), 1)
        off = s // 2 + 1
        rr = np.arange(off, a.rows - off + 1, step)
        cc = np.arange(off, a.cols - off + 1, step)
        # one cell of the region grid (windows whose centre lies in it), somewhere in the middle of the frame with faces around
        for (r0, c0) in ((a.cell_h, a.cell_w), (2 * a.cell_h, 3 * a.cell_w)):
            ri = rr[(rr >= r0) & (rr < r0 + a.cell_h)]
            ci = cc[(cc >= c0) & (cc < c0 + a.cell_w)]
            ni, nj = len(ri), len(ci)
            if ni == 0 or nj == 0:
                continue
            for arr in arrangements:
                # enumeration: list of (i, j) per lane slot, -1 = no window
                if arr == "row":
                    f = np.arange(((ni * nj + 63) // 64) * 64)
                    I, J = f // nj, f % nj
                    valid = f < ni * nj
                else:
                    bw, bh = (16, 4) if arr == "16x4" else (8, 8)
                    nbx, nby = (nj + bw - 1) // bw, (ni + bh - 1) // bh
                    lane = np.arange(64)
                    if arr == "16x4":
                        lx = (lane & 7) + 8 * (lane >> 5)
                        ly = (lane >> 3) & 3
                    else:
                        lx = lane & 7
                        ly = ((lane >> 3) & 3) + 4 * (lane >> 5)
                    by, bx = np.meshgrid(np.arange(nby), np.arange(nbx), indexing="ij")
                    I = (by.ravel()[:, None] * bh + ly[None, :]).ravel()
                    J = (bx.ravel()[:, None] * bw + lx[None, :]).ravel()
                    valid = (I < ni) & (J < nj)
                I = np.where(valid, I, 0)
                J = np.where(valid, J, 0)
                R, Cc = ri[I], ci[J]
                alive = valid.copy()
                order = np.arange(len(R))
                out = np.zeros(len(R), dtype=np.float32)
                for t in range(a.trees):
                    # the stage's input: compacted survivors in enumeration order (tree 0: the enumeration itself, holes included)
                    if t == 0:
                        sel = order
                        act = alive
                    else:
                        sel = order[alive]
                        pad = (-len(sel)) % 64
                        act = np.concatenate([np.ones(len(sel), bool), np.zeros(pad, bool)])
                        sel = np.concatenate([sel, np.zeros(pad, dtype=sel.dtype)])
                    if act.sum() == 0:
                        break
                    r, c = R[sel], Cc[sel]
                    idx = np.ones(len(sel), dtype=np.int64)
                    tc = pg.codes[t].astype(np.int64)
                    nw = len(sel) // 64
                    for l in range(6):
                        c0_, c1_, c2_, c3_ = tc[4 * idx], tc[4 * idx + 1], tc[4 * idx + 2], tc[4 * idx + 3]
                        y1, x1 = (r * 256 + c0_ * s) >> 8, (c * 256 + c1_ * s) >> 8
                        y2, x2 = (r * 256 + c2_ * s) >> 8, (c * 256 + c3_ * s) >> 8
                        # LDS address inside the region (origin: the cell's corner minus a halo): row * pitch + column
                        a1 = np.where(act, (y1 - r0 + 64) * a.pitch + (x1 - c0 + 64), -1).reshape(nw, 64)
                        a2 = np.where(act, (y2 - r0 + 64) * a.pitch + (x2 - c0 + 64), -1).reshape(nw, 64)
                        tot[arr][t, l] += (group_cycles(a1) + group_cycles(a2)) * nw
                        idx = 2 * idx + (img[y1, x1] <= img[y2, x2])
                    nin[arr][t] += nw
                    o = out[sel] + pg.preds[t][idx - 64]
                    keep = act & (o > pg.thr[t])
                    out[sel[act]] = o[act]
                    newalive = np.zeros(len(R), bool)
                    newalive[sel[keep]] = True
                    alive = newalive
    print(f"{a.kind} {a.cols}x{a.rows}, rungs {scales[0]}..{scales[-1]}, two cells of {a.cell_w}x{a.cell_h}, pitch {a.pitch}: LDS cycles per wave-wide BYTE READ (2 = conflict-free, ~6.9 = every lane its own node)")
    for arr in arrangements:
        print(f"  arrangement {arr}:")
        for t in range(a.trees):
            if nin[arr][t] == 0:
                continue
            per = tot[arr][t] / nin[arr][t] / 2
            print("    tree %d (%7d wave batches): levels %s   mean %.2f" % (t, nin[arr][t], " ".join("%.2f" % v for v in per), per.mean()))
    base = sum(tot["row"][t].sum() for t in range(a.trees))
    for arr in arrangements[1:]:
        v = sum(tot[arr][t].sum() for t in range(a.trees))
        print("  pixel-read LDS cycles, %s / row: %.3f" % (arr, v / base))


if __name__ == "__main__":
    main()
