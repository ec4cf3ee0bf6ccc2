#!/usr/bin/env python3
"""Race / hazard checker for the NARROW-tile four-wave GEMM bodies (scripts/gen_w4n.py): scripts/w4a_hazard_check.py's symbolic executor on the
three-stage ring.  Rules R1-R10 are the wide bodies' (no register read or overwritten while a load still has to write it; a stage read only after its
pieces were waited for AND a barrier passed; a stage refilled only after every read of it returned AND a barrier passed; whole K-steps published; nothing
outstanding at the end; the fixed-latency hazards gfx940-class hardware does not interlock).  What differs:

  * three stages per operand: the LDS-DMA's m0 constant and the fragment reads' (address register, offset) decide the stage -- A: stages 0 / 1 through
    v116 / v117 with offsets < 64K, stage 2 through v118 / v119; B: one address pair, offset / 16K;
  * a K-step is 8 A pieces + 4 B pieces; ONE barrier per K-step both publishes K-step k + 1 and frees the stage K-step k has finished with;
  * R6 also covers v_accvgpr_mov_b32 (the finished tile's accumulators move to a[128:255] in front of the next tile's first MFMAs);
  * R11: an accumulator of the second set is read by the epilogue only between the move that filled it and the next move into it, and every one of the 128
    is read exactly as often per tile as the epilogue form needs (once) -- i.e. the queue really drains one whole tile per tile;
  * R12: the epilogue queue is empty when a tile's last K-step ends (no store of tile t may be issued after tile t + 1's accumulators moved in);
  * R13: the data registers of a wide store (buffer_store_dwordx3 / x4) are not written within two wait states behind it (the store reads them late: the guide's
    "an asm ..._store_dwordx4 ends with s_nop 1"; the wide generator pads, this one relies on the next writer being far away -- proven here);
  * R6 with the guide's 12 states between an MFMA and any other reader of its result (the wide checker's 11 is the ISA table's number for 8 passes).

    python scripts/w4n_hazard_check.py            (all bodies x K = 576 / 768 / 960 / 2304 x 1-3 tiles; exit code 1 on the first violation)
"""
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_w4n as G  # noqa: E402
import w4a_hazard_check as H  # noqa: E402
from w4a_hazard_check import Violation, vregs  # noqa: E402

NEED = {"A": 8, "B": 4}


class Sim(H.Sim):
    def __init__(self, lines, K, ntiles, name=""):
        super().__init__(lines, 8, K, ntiles, name)
        self.reg = {(o, s_): dict(state="EMPTY", dma=[], reads=[], drained=0) for o in "AB" for s_ in (0, 1, 2)}
        self.acc1_epoch = [0] * 128        # how many times a[128 + r] has been filled
        self.acc1_reads = [0] * 128        # reads since the last fill
        self.moves = 0

    # ---- LDS protocol with per-operand piece counts
    def lds_dma(self, i, operand, stage, op):
        rg = self.reg[(operand, stage)]
        if rg["state"] == "READY":
            if rg["drained"] != len(rg["reads"]) or any(not o["done"] for o in rg["reads"]):
                self.fail(i, f"R3: LDS-DMA into {operand} stage {stage} while fragment reads of it are outstanding or no barrier has been passed since they returned "
                             f"({rg['drained']} of {len(rg['reads'])} reads covered)")
            rg.update(state="FILLING", dma=[], reads=[], drained=0)
        elif rg["state"] == "EMPTY":
            rg.update(state="FILLING", dma=[])
        rg["dma"].append(op)
        if len(rg["dma"]) > NEED[operand]:
            self.fail(i, f"R4: more than {NEED[operand]} DMA pieces into {operand} stage {stage} before it was published")

    def barrier(self, i):
        for (operand, stage), rg in self.reg.items():
            if rg["state"] == "FILLING" and len(rg["dma"]) == NEED[operand] and all(o["done"] for o in rg["dma"]):
                rg.update(state="READY", reads=[], drained=0)
                if operand == "A":
                    self.stats["ksteps"] += 1
            elif rg["state"] == "READY" and all(o["done"] for o in rg["reads"]):
                rg["drained"] = len(rg["reads"])

    def step_(self, i):
        ln = self.lines[i]
        op, _, rest = ln.partition(" ")
        args = [a.strip() for a in rest.split(",")] if rest else []
        if op == "s_add_u32" and args and args[0] == "m0":
            self.n_instr += 1
            self.m0_w = self.ws + 1
            base = int(re.fullmatch(r"s(\d+)", args[1]).group(1))
            const = int(args[2], 0)
            self.m0 = ("A", const // 32768) if base == G.S_M0A else ("B", const // 16384)
            return i + 1
        if op.startswith("ds_read") and vregs(args[1].split()[0])[0] in G.V_AA + G.V_AA2 + G.V_AB:
            self.n_instr += 1
            dst = vregs(args[0])
            addr = vregs(args[1].split()[0])
            self.read_v(i, addr)
            self.write_v(i, dst)
            m = re.search(r"offset:(\d+)", rest)
            off = int(m.group(1)) if m else 0
            o = dict(line=i, what="a fragment read (ds_read)", dst=dst)
            if addr[0] in G.V_AA:
                if off >= 65536:
                    self.fail(i, "a 16-bit ds_read offset")
                self.lds_read(i, "A", off // 32768, o)
            elif addr[0] in G.V_AA2:
                if off >= 32768:
                    self.fail(i, "stage-2 A read past the stage")
                self.lds_read(i, "A", 2, o)
            else:
                self.lds_read(i, "B", off // 16384, o)
            self.issue(self.lg, o, self.LGKM_MAX)
            for r in dst:
                self.pend[r] = o
            return i + 1
        if op == "v_accvgpr_mov_b32":
            self.n_instr += 1
            d, s_ = int(args[0][1:]), int(args[1][1:])
            if not (128 <= d < 256 and s_ == d - 128):
                self.fail(i, "an accumulator move that is not a[r] -> a[128 + r]")
            self.gap(i, self.acc_w, [s_], 12, "R6")
            r = d - 128
            if self.acc1_epoch[r] > 1 and self.acc1_reads[r] != 1:      # (epoch 1 is the phantom tile in front of the first one: its reads happen too)
                self.fail(i, f"R11: a[{d}] refilled after {self.acc1_reads[r]} reads of the previous tile's value (1 expected)")
            self.acc1_epoch[r] += 1
            self.acc1_reads[r] = 0
            self.moves += 1
            self.valu_w[("a", d)] = self.ws + 1
            return i + 1
        if op == "global_load_dwordx2":                      # (CE_EXP: the rows' target ids) a VMEM load like any other: in-order vmcnt, destination pending until waited for
            self.n_instr += 1
            dst = vregs(args[0])
            self.read_v(i, vregs(args[1]))
            self.write_v(i, dst)
            o = dict(line=i, what="a global load", dst=dst)
            self.issue(self.vm, o, self.VM_MAX)
            for r in dst:
                self.pend[r] = o
            return i + 1
        if op in ("v_permlane32_swap_b32", "v_permlane16_swap_b32"):
            # both operands are read AND written; a VALU result may be swapped only two wait states after it was produced (the 8-wave kernel's s_nop 1: R7's rule)
            self.n_instr += 1
            regs = vregs(args[0]) + vregs(args[1])
            self.read_v(i, regs)
            self.write_v(i, regs)
            self.gap(i, self.valu_w, regs, 2, "R7")
            self.gap(i, self.trans_w, regs, 1, "R8")
            for r in regs:
                self.valu_w[r] = self.ws + 1
                self.trans_w.pop(r, None)
            return i + 1
        if op == "v_accvgpr_read_b32":
            a_ = int(args[1][1:])
            if a_ < 128:
                self.fail(i, "the epilogue reads the K loop's accumulator set")
            if self.acc1_epoch[a_ - 128] == 0:
                self.fail(i, f"R11: a[{a_}] read before any tile moved into it")
            self.acc1_reads[a_ - 128] += 1
        return super().step_(i)


    def run(self):
        i, n = 0, len(self.lines)
        while i < n:
            i = self.step(i)
            if self.n_instr > 20_000_000:
                raise Violation(f"{self.name}: does not terminate")
        if any(not o["done"] for o in self.vm + self.lg) or self.pend:
            raise Violation(f"{self.name} K={self.K} tiles={self.ntiles}: R5: operations outstanding at the end of the kernel")
        nk = self.K // 64
        want_mfma = nk * self.ntiles * 64
        want_k = nk * self.ntiles + 1                      # (+ the first of the three phantom K-steps the last tile's last triple fetches through null descriptors)
        if self.stats["mfma"] != want_mfma:
            raise Violation(f"{self.name} K={self.K} tiles={self.ntiles}: executed {self.stats['mfma']} MFMAs, expected {want_mfma}")
        if self.stats["ksteps"] != want_k:
            raise Violation(f"{self.name} K={self.K} tiles={self.ntiles}: published {self.stats['ksteps']} K-steps, expected {want_k}")
        if self.moves != 128 * (self.ntiles + 1) or any(r != 1 for r in self.acc1_reads):
            raise Violation(f"{self.name} K={self.K} tiles={self.ntiles}: R11/R12: {self.moves} accumulator moves, reads of the last tile {set(self.acc1_reads)}")
        return self.stats


def check_all(shapes=((576, 1), (576, 3), (768, 2), (960, 2), (2304, 1)), verbose=False, bodies=None, opts=()):
    n = 0
    for bkm, epi in (bodies or G.BODIES):
        lines, _ = G.generate(bkm, epi, opts)
        for K, ntiles in shapes:
            st = Sim(lines, K, ntiles, name=f"narrow {'KM' if bkm else 'KC'} {epi}").run()
            n += 1
            if verbose:
                print(f"narrow {'KM' if bkm else 'KC'} {epi:8s} K={K:4d} tiles={ntiles}: {st['mfma']} MFMAs, {st['ksteps']} K-steps published: ok", flush=True)
    return n


if __name__ == "__main__":
    try:
        n = check_all(verbose=True)
    except Violation as e:
        print("VIOLATION:", e)
        sys.exit(1)
    print(f"{n} (body, shape) runs: every counted wait and the one barrier per K-step hold")
