#!/usr/bin/env python3
"""Issue-cycle MODEL of the narrow-tile asm GEMM bodies, counted from the generated text (scripts/gen_w4n.py) -- NOT a measurement.

One wave per SIMD issues its instructions in order: a K-step cannot take less than the sum of its instructions' issue cycles, nor less than its MFMAs' time in the matrix
pipe (16 cycles each at the peak), nor less than its DMA's time on the L2 -> LDS path.  Issue costs assumed (wave64 on a 16-lane SIMD): vector ALU / DPP / accumulator
move 4 cycles, MFMA 4 to issue, LDS and buffer instructions 4, scalar 1, s_nop n: n + 1; BESIDE MFMAs (the narrow queue) a transcendental 4 + 2 and a packed
fp32 operation 4 + 22 (the microarchitecture guide's measured prices of fillers between MFMAs); ALONE (the wide bodies' epilogue) a transcendental 16 (quarter rate), a packed operation 4.  DMA: 48 KB per K-step at 36 B/clk/CU
(profiles/r02_dma_probe.txt, 128-byte rows, two stages in flight).  What the model cannot see: waits that actually stall (store acknowledges behind the in-order vmcnt,
a late DMA piece), LDS bank conflicts, the clock the chip chooses.  The wide bodies' column uses the same issue costs on scripts/gen_w4a.py's text and the K-step time
MEASURED for them (2 600 cycles) instead of a modelled one.

    python scripts/w4n_issue_model.py > profiles/r06_w4n_issue_model.txt
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_w4a as GA  # noqa: E402
import gen_w4n as G  # noqa: E402

DMA_CYCLES = 48 * 1024 / 36.0
WIDE_KSTEP_MEASURED = 2600
CE_8WAVE_TILE_MEASURED = int(799.7e-6 / 32 * 2.1e9)       # profiles/r05_single_stream_kernel_stats.csv, clock: profiles/r05_power_ab.txt


BESIDE_MFMA = True      # narrow bodies: the queue runs between MFMAs.  /opt/skills/guides/MI355X_MICROARCH.md ("price of one filler beside MFMAs, beyond its issue slot"): a transcendental
                        # ~2, a packed fp32 VALU operation +22 (1 v_pk_fma_f32 vs 2 v_fma_f32) ... +13 each (2 v_pk_add_f32: +26).  Alone (the wide bodies' epilogue): transcendentals at quarter rate.


def cost(ln):
    op = ln.split()[0]
    if ln.endswith(":"):
        return 0, "label"
    if op.startswith("v_mfma"):
        return 4, "mfma"
    if op in ("v_rcp_f32", "v_exp_f32"):
        return (6 if BESIDE_MFMA else 16), "trans"
    if op.startswith("v_pk_"):
        return (4 + 22 if BESIDE_MFMA else 4), "pk"
    if op.startswith("v_"):
        return 4, "valu"
    if op.startswith("ds_") or op.startswith("buffer_"):
        return 4, "mem"
    if op == "s_nop":
        return int(ln.split()[1], 0) + 1, "nop"
    return 1, "salu"


PIECE = 35              # issue cycles of one LDS-DMA piece (buffer_load_dwordx4 ... lds): the guide measures ~60 among bare MFMAs and 100-185 in a crowded phase; the wide bodies' own
                        # measured K-step leaves (2 600 - 2 048) / 16 = 35 per piece if everything beyond the MFMAs is charged to the pieces -- the smallest number consistent with both
GAP = 12                # cycles of other instructions an MFMA hides behind itself (16 in the pipe, 4 to issue): the guide's "<= 5 fillers of ~4 cycles per 32-cycle MFMA", halved


def tally(lines):
    tot, cls = 0, {}
    for ln in lines:
        c, k = cost(ln)
        tot += c
        cls[k] = cls.get(k, 0) + 1
    return tot, cls


def gap_time(lines):
    """A stricter model of one K-step: every MFMA takes 16 cycles and hides GAP cycles of what follows it; whatever a gap holds beyond that is exposed.  A DMA piece costs PIECE."""
    t, in_gap, started = 0.0, 0.0, False
    for ln in lines:
        c, k = cost(ln)
        if k == "mfma":
            if started:
                t += 16 + max(0.0, in_gap - GAP)
            else:
                t += in_gap                            # (what stands in front of the step's first MFMA)
            in_gap, started = 0.0, True
        else:
            in_gap += PIECE if (k == "mem" and ln.rstrip().endswith(" lds")) else c
    return t + 16 + max(0.0, in_gap - GAP)


def narrow(bkm, epi, opts=()):
    g = G.Gen(bkm, epi, opts)
    marks = []
    real_step = g.step

    def step(*a, **k):
        i0 = len(g.a.l)
        real_step(*a, **k)
        marks.append((i0, len(g.a.l)))
    g.step = step
    g.body()
    steps = [tally(g.a.l[i0:i1]) for i0, i1 in marks]       # text order: first triple, second, middle, last (12 K-steps)
    assert len(steps) == 12
    t = [max(64 * 16, DMA_CYCLES, s[0]) for s in steps]
    narrow.gap = sum(max(DMA_CYCLES, gap_time(g.a.l[i0:i1])) for i0, i1 in marks)
    # K = 768: first + second + one pass of the middle + last = the 12 K-steps of the text
    return steps, t


def wide_epilogue(bkm, epi):
    global BESIDE_MFMA
    BESIDE_MFMA = False
    try:
        return wide_epilogue_(bkm, epi)
    finally:
        BESIDE_MFMA = True


def wide_epilogue_(bkm, epi):
    g = GA.Gen(bkm, epi, 8)
    lines = g.body()
    i0 = max(i for i, l in enumerate(lines) if l == "s_nop 15")       # the epilogue starts behind the tile's last K-step
    i1 = max(i for i, l in enumerate(lines) if l.startswith("s_sub_u32 s%d, s%d, 1" % (GA.S_TILE, GA.S_TILE)))
    return tally(lines[i0:i1])


if __name__ == "__main__":
    print("# Issue-cycle MODEL of the narrow-tile asm GEMM (scripts/w4n_issue_model.py): counted from the generated text, NOT measured.  Cycles per wave.")
    print("# narrow K-step = max(1 024 MFMA-pipe cycles, %d DMA cycles (48 KB at 36 B/clk/CU), issue cycles of its text); wide K-step = %d (measured, DESIGN 7.2) + its epilogue's issue cycles; ceexp: the 8-wave kernel's measured %d cycles per tile" % (DMA_CYCLES, WIDE_KSTEP_MEASURED, CE_8WAVE_TILE_MEASURED))
    print("# body           issue cycles of the 12 K-steps of a K = 768 tile (first triple | second | middle | last)                       256 x 128 tile   per 256 x 256   wide body, per 256 x 256   narrow / wide     STRICTER model: tile, narrow / wide")
    print("# (stricter model: an MFMA hides only %d cycles of the instructions behind it, an LDS-DMA piece costs %d cycles to issue -- see gap_time; the truth is for the hardware to say)" % (GAP, PIECE))
    for form, opts in (("loop form (first / second / middle / last triple; only the first and the last drain the epilogue queue)", ()),
                       ("loop-free form, K = 768 (flat=12: every K-step but the first drains the queue) -- what K = 768 launches take", ("flat=12",)),
                       ("the same with the queue's arithmetic left PACKED (pk=1: v_pk_fma / mul / add_f32 as in the wide bodies' epilogue) -- why it is not", ("flat=12", "pk=1"))):
        print(f"# {form}")
        for bkm, epi in G.BODIES:
            steps, t = narrow(bkm, epi, opts)
            tile = sum(t)
            if epi == "ceexp":                                 # no wide asm body exists: the 8-wave kernel's launch, measured (799.7 us for 68 x 120 tiles of 256 x 256 = 32 rounds at 2.1 GHz)
                wide = CE_8WAVE_TILE_MEASURED
            else:
                we, _ = wide_epilogue(bkm, epi)
                wide = 12 * WIDE_KSTEP_MEASURED + we
            iss = " ".join(f"{s[0]:5d}" for s in steps)
            print(f"  {'KM' if bkm else 'KC'} {epi:8s}  {iss[:17]} | {iss[18:35]} | {iss[36:53]} | {iss[54:]}      {tile:8.0f}        {2 * tile:8.0f}        {wide:8d}                 {2 * tile / wide:.2f}"
                  f"          {narrow.gap:8.0f}   {2 * narrow.gap / wide:.2f}")
    print("# launch level at 17 408 tokens on 256 CUs (tiles -> rounds of the persistent grid; the wide bodies at their per-launch height):")
    for name, N, wide_tiles in (("out-proj (N = 768)", 768, 234), ("q|k|v (N = 2 304)", 2304, 702), ("FFN lin1 (N = 3 072)", 3072, 936)):
        nt = 68 * (N // 128)
        print(f"#   {name:22s} narrow {nt:5d} tiles = {-(-nt // 256)} rounds of 256 x 128     wide {wide_tiles:4d} tiles = {-(-wide_tiles // 256)} rounds of 224 x 256 (= {1.75 * -(-wide_tiles // 256):.2f} narrow-tile equivalents)")
